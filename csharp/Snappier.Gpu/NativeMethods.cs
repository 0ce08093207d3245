// P/Invoke surface of libsnappier_hip.so -- one declaration per function of include/snappier_hip.h, same order.
// Every signature is cdecl, plain pointers and sizes (size_t = nuint, uint64_t = ulong, int32_t status = SnpStatus).
using System;
using System.Runtime.InteropServices;

namespace Snappier.Gpu;

/// <summary>snp_status (include/snappier_hip.h); values 1:1 with the reference's exceptions (ThrowHelper.cs:8-36).</summary>
public enum SnpStatus : int
{
    Ok = 0,
    OutputTooSmall = 1,     // "Output buffer is too small."  -> false from Try*, ArgumentException from Compress/Decompress
    BadOffset = 2,          // "Invalid copy offset"          -> InvalidDataException
    TooLong = 3,            // "Data too long"                -> InvalidDataException
    Incomplete = 4,         // "Incomplete Snappy block."     -> InvalidDataException
    BadLength = 5,          // "Invalid stream length"        -> InvalidDataException
    CrcMismatch = 6,        // "Chunk CRC mismatch."          -> InvalidDataException
    ChunkType = 7,          // "Unknown chunk type"           -> InvalidDataException
    Overlap = 8,            // "Input and output spans must not overlap." -> InvalidOperationException
    BadArg = 9,
    Device = 10,            // no HIP device / HIP error: the caller falls back to the managed codec
    TruncatedStream = 11,
}

/// <summary>snp_option (include/snappier_hip.h): per-context configuration; no option changes a result.</summary>
public enum SnpOption : int
{
    DecodeLayout = 1,           // 0 by the previous batch, 1 one block per wavefront, 2 a lane per small block, 3/4/5 a team of 4/8/16 lanes, 6 the serial kernel
    SmallBlockMax = 2,
    SmallBlockMinBatch = 3,
    CompressLayout = 4,         // 0 by batch size, 2 fragment per lane (HBM tables), 3 fragment per wavefront (LDS table), 4 the same with the table in a global-memory slot, 5 both table forms side by side
    CompressWindowMaxBatch = 5,
    TableProbeTries = 6,        // workspaces' worth of candidate pieces the DEVICE's hash-table workspace search may hold (default 2; 3..24 = the thorough search, 1 = a plain workspace of the context's own)
    TableProbeMaxBytes = 7,     // cap on the transient footprint of that search (default: half of free memory; an explicit cap is honoured up to 7/8)
    ParallelDecodeMin = 8,
    Fenced = 9,
    DecodeLeftovers = 10,
    CrcKernel = 11,          // CRC-32C kernel: 0 = three LDS tables (default, 5.8 TB/s), 1 = table-free (1.7 TB/s), 2 = four 8-bit tables (round 3)
    // launch shapes (round 6; snappier_hip.h explains each -- defaults are the measured best, no option changes a result)
    CompressWindowPositions = 12,
    CompressWindowGlobalMinBatch = 13,
    CompressLaneStores = 14,
    CompressLaneProbes = 15,
    CompressLanesPerWavefront = 16,
    CompressSlice = 17,
    CompressSmallInputLds = 18,
    CompressSmallInputLanes = 19,
    FrameScan = 20,
    DecodeLdsThrottle = 21,
    CompressWindowGlobalSlots = 22,
    CompressWindowDualMinBatch = 23,
}

/// <summary>Names this binding used before round 5 (same numbers).</summary>
public static class SnpOptionCompat
{
    [System.Obsolete("renamed: SnpOption.CrcKernel (same number; 0 / 1 mean the same)")]
    public const SnpOption CrcTableFree = SnpOption.CrcKernel;
}

public enum SnpHash : int
{
    Crc32C = 0,             // HashTable.cs:109-117 (x64 SSE4.2 / ARM64 CRC, .NET 8+): what the managed build emits on this host
    Mul = 1,                // HashTable.cs:121-122 (netstandard2.0 / intrinsics off)
}

internal static unsafe class NativeMethods
{
    private const string Lib = "snappier_hip";                                  // libsnappier_hip.so
    private const CallingConvention Cc = CallingConvention.Cdecl;

    public const int BlockSize = 65536, MaxBlockCompressed = 76491, VarintMax = 5, StreamHeaderLength = 10, ChunkHeaderLength = 8;

    // ---- context
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_ctx_create(int device, int hashVariant, IntPtr stream, out IntPtr ctx);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern void snp_ctx_destroy(IntPtr ctx);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_ctx_set_stream(IntPtr ctx, IntPtr stream);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern IntPtr snp_ctx_last_error(IntPtr ctx);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_ctx_synchronize(IntPtr ctx);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern ulong snp_ctx_counter(IntPtr ctx, int which);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_ctx_set_option(IntPtr ctx, int option, long value);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_ctx_get_option(IntPtr ctx, int option, out long value);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_ctx_reserve_compress(IntPtr ctx, uint nfragments);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern IntPtr snp_status_string(int status);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern IntPtr snp_version();

    // ---- host-only arithmetic
    [DllImport(Lib, CallingConvention = Cc)] internal static extern long snp_max_compressed_length(long n);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern long snp_max_fragment_compressed_length(long n);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_get_uncompressed_length(byte* input, nuint n, out uint length, out uint headerBytes);

    // ---- single buffer, host pointers (blocking)
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_try_compress(IntPtr ctx, byte* input, nuint n, byte* output, nuint cap, out nuint written);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_try_decompress(IntPtr ctx, byte* input, nuint n, byte* output, nuint cap, out nuint written);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_try_compress_segments(IntPtr ctx, byte** segments, nuint* segmentLengths, uint nseg, byte* output, nuint cap, out nuint written);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_try_decompress_segments(IntPtr ctx, byte** segments, nuint* segmentLengths, uint nseg, byte* output, nuint cap, out nuint written);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_crc32c(IntPtr ctx, byte* input, nuint n, int masked, out uint crc);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern long snp_frame_max_encoded_length(long n);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_frame_encode(IntPtr ctx, byte* input, nuint n, byte* output, nuint cap, out nuint written);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_frame_decoded_length(byte* input, nuint n, out ulong length);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_frame_decode(IntPtr ctx, byte* input, nuint n, byte* output, nuint cap, out nuint written);

    // ---- batch, device pointers (asynchronous on the context's stream)
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_compress_batch(IntPtr ctx, IntPtr dIn, IntPtr dInOff, IntPtr dInLen, uint nblocks, IntPtr dOut, IntPtr dOutOff, IntPtr dOutLen, IntPtr dStatus);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_decompress_batch(IntPtr ctx, IntPtr dIn, IntPtr dInOff, IntPtr dInLen, uint nblocks, IntPtr dOut, IntPtr dOutOff, IntPtr dOutCap, IntPtr dOutLen, IntPtr dStatus);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_crc32c_batch(IntPtr ctx, IntPtr dIn, IntPtr dInOff, IntPtr dInLen, uint nblocks, int masked, IntPtr dCrc);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_concat_batch(IntPtr ctx, IntPtr dIn, IntPtr dInOff, IntPtr dInLen, uint nblocks, IntPtr dOut, IntPtr dDstOff);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern ulong snp_frame_encode_workspace(ulong n);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_frame_encode_device(IntPtr ctx, IntPtr dIn, ulong n, IntPtr dOut, ulong cap, IntPtr dWritten, IntPtr dWork);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_frame_decode_chunks_device(IntPtr ctx, IntPtr dIn, IntPtr chunkType, IntPtr bodyOff, IntPtr bodyLen, IntPtr chunkCrc, uint nchunks, IntPtr dOut, IntPtr outOff, IntPtr outCap, IntPtr outLen, IntPtr status);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern ulong snp_frame_decode_workspace(uint maxChunks);
    [DllImport(Lib, CallingConvention = Cc)] internal static extern SnpStatus snp_frame_decode_device(IntPtr ctx, IntPtr dIn, ulong n, IntPtr dOut, ulong cap, uint maxChunks, IntPtr dWork, IntPtr dResult);

    internal static string StatusString(SnpStatus s) => Marshal.PtrToStringUTF8(snp_status_string((int)s)) ?? s.ToString();
}
