// decompress.hip -- launchers of the block decoders, and the serial decoder: one wavefront walks the tags of one block one at a time.
//
// The default decoder is decode_chains.hip (k_decode_chains: sub-chain parse + tag-per-lane batches; whole blocks, lists of blocks, and
// the 64 KiB fragments of one large block).  The serial kernel here is the reference's loop as it stands (SnappyDecompressor.cs:184-347:
// tag by tag, ~65 scalar instructions each) -- the baseline of the parity tests and a debugging aid (SNP_OPT_DECODE_LAYOUT keeps it
// reachable); every front end hands its irregular remainder to the same code (decode_common.h, serial_tail).
// Front ends that were measured and lost live in lab/decompress_r04.hip (variant builds only).
#include "decode_common.h"

// decode_chains.hip
extern "C" hipError_t snp_launch_decode_chains(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out, const u64* out_off,
                                               const u32* out_cap, u32* out_len, i32* status, const u8* chunk_type, int fenced, int redo_only,
                                               unsigned lds_bytes, hipStream_t stream, const u32* frag_skip);
extern "C" hipError_t snp_launch_decode_chains_list(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out, const u64* out_off,
                                                    const u32* out_cap, u32* out_len, i32* status, const u8* chunk_type, int fenced,
                                                    unsigned lds_bytes, hipStream_t stream, const u32* list, u32* ctl, u32 waves, u32 sub_cap);

namespace {

template <bool FENCED, bool FRAG>
__global__ __launch_bounds__(SNP_WAVE) void k_decompress_serial(SNP_D_PARAMS)
{
    const u32 b = blockIdx.x;
    const u32 lane = lane_id();
    DecBlk B;
    if (!block_begin<FRAG>(SNP_D_ARGS, b, lane, B)) return;
    serial_tail<FENCED, FRAG>(B, b, lane, out_len, status);
}

}  // namespace

// The blocks of `list` (decode_chains.hip, k_decode_chains_list): waves = wavefronts to launch (one chip-full).
// mode bit 0: FENCED; bits 8..: dynamic LDS bytes / 256 requested per wavefront purely to cap how many blocks a CU decodes at once.
extern "C" hipError_t snp_launch_decompress_list(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                                 const u64* out_off, const u32* out_cap, u32* out_len, i32* status,
                                                 const u8* chunk_type, int mode, hipStream_t stream, const u32* list, u32* ctl,
                                                 u32 waves, u32 sub_cap)
{
    return snp_launch_decode_chains_list(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, status, chunk_type, mode & 1,
                                         static_cast<unsigned>(mode >> 8) * 256u, stream, list, ctl, waves, sub_cap);
}

// mode bit 0: FENCED (drain vmcnt before a wave reads output bytes it stored itself), bit 1: the serial kernel, bit 4: only the blocks
// decompress_small.hip left marked -1, bits 8..: LDS throttle as above.  frag_skip != nullptr: the blocks are the 64 KiB fragments of one
// large block (tag_index.hip).
extern "C" hipError_t snp_launch_decompress(const u8* in, const u64* in_off, const u32* in_len, u32 nblocks, u8* out,
                                            const u64* out_off, const u32* out_cap, u32* out_len, i32* status,
                                            const u8* chunk_type, int mode, hipStream_t stream, const u32* frag_skip)
{
    if (nblocks == 0) return hipSuccess;
    const unsigned lds_bytes = static_cast<unsigned>(mode >> 8) * 256u;
    const int redo_only = (mode >> 4) & 1;
    if (!(mode & 2))
        return snp_launch_decode_chains(in, in_off, in_len, nblocks, out, out_off, out_cap, out_len, status, chunk_type, mode & 1, redo_only,
                                        lds_bytes, stream, frag_skip);
#define SNP_LAUNCH_SERIAL(F, G)                                                                                              \
    hipLaunchKernelGGL((k_decompress_serial<F, G>), dim3(nblocks), dim3(SNP_WAVE), lds_bytes, stream, in, in_off, in_len,    \
                       nblocks, out, out_off, out_cap, out_len, status, (G) ? nullptr : chunk_type, frag_skip, (G) ? 0 : redo_only)
    if (frag_skip) {
        if (mode & 1) SNP_LAUNCH_SERIAL(true, true); else SNP_LAUNCH_SERIAL(false, true);
    } else {
        if (mode & 1) SNP_LAUNCH_SERIAL(true, false); else SNP_LAUNCH_SERIAL(false, false);
    }
#undef SNP_LAUNCH_SERIAL
    return hipGetLastError();
}
