/* snappier_hip_debug.h -- TEST HOOKS exported by libsnappier_hip.so next to the product API of snappier_hip.h.
 *
 * Not part of the drop-in boundary (no DllImport binds these; csharp/ never calls them): they exist so that the parity
 * tests can drive a DEVICE FUNCTION of a kernel directly with the reference's own known-answer vectors, instead of
 * observing it only through compressed bytes.  All pointers are device memory; every call enqueues one tiny kernel on
 * `stream` (a hipStream_t) and returns the hipError_t of the launch as an int.
 */
#ifndef SNAPPIER_HIP_DEBUG_H
#define SNAPPIER_HIP_DEBUG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FindMatchLength (SnappyCompressor.cs:562-688; KATs SnappyCompressorTests.cs:10-96) as the WINDOW compressor computes it
 * (compress_win.hip, wave_match_extend: 64 lanes compare, one ballot): d_buf[0..n) is a fragment, the match candidate
 * starts at `cand`, the position at `p`, `known` bytes are already known to match; *d_out = total match length, bounded by
 * the fragment end n. */
int snp_debug_match_length(const uint8_t* d_buf, uint32_t n, uint32_t p, uint32_t cand, uint32_t known, uint32_t* d_out,
                           void* stream);

/* The same function as the LANE compressor computes it (compress_lanes.hip, lane_find_match_length: 8 bytes per step,
 * XOR + count-trailing-zeros, byte tail): *d_out = bytes of d_buf[s1..] and d_buf[s2..] that match, s2 + result <= n.
 * This is the form the headline kernel (>= 16 384 fragments) runs. */
int snp_debug_lane_match_length(const uint8_t* d_buf, uint32_t n, uint32_t s1, uint32_t s2, uint32_t* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SNAPPIER_HIP_DEBUG_H */
