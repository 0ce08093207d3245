// Snappier.Gpu.Snappy -- the block API of Snappier.Snappy (Snappier/Snappy.cs:20-283) with the codec work done by
// libsnappier_hip.so.  Same method names, argument meaning, return values and exception types/messages; allocation
// (arrays, ArrayPool owners) stays managed, the native side never keeps a pointer past the call.
using System;
using System.Buffers;
using System.IO;

namespace Snappier.Gpu;

public static unsafe class Snappy
{
    // ---- routing ------------------------------------------------------------------------------------------------------------
    // One host-pointer call into the library costs a fixed ~0.3 ms (decompress) to ~1.5 ms (compress) of launches and PCIe round
    // trips before the first byte moves, then runs PCIe-bound (profiles/r04zz_host_api.jsonl: 1 MiB 0.66 / 1.6 GB/s compress /
    // decompress, 4 MiB 2.1 / 5.1, 16 MiB 7.5 / 11.9, 256 MiB 25.0 / 24.2, 1 GiB 25.6 / 25.2).  ONE managed Snappier thread does
    // ~0.5-1 GB/s compress and ~1-2 GB/s decompress whatever the size: a single call is faster on the GPU from 2-4 MiB on.  A host
    // that keeps ALL its cores busy with independent Snappier calls, though, moves 10-25 GB/s in aggregate (BENCH cpu_baseline legs,
    // INTEGRATION.md "Where the GPU loses"): against that the PCIe-bound call only wins from ~64 MiB (compress) / ~128 MiB
    // (decompress) per call, and device-resident callers should use the batch entry points instead.  Hence two presets; the default
    // is the conservative one.  Below the thresholds (and whenever no HIP device is usable) the call stays on the managed path.
    // The BYTES are the same on both paths only when the GPU context's hash equals the one managed Snappier picks at run time:
    // SnpHash.Crc32C = x64 with SSE4.2 / ARM64 with CRC on .NET 8+; elsewhere (older runtimes, ARM without CRC, a Mul context)
    // managed Snappier emits the Mul bytes -- still valid Snappy, but output then changes with input size across the threshold.
    // AssertHashMatchesManaged() checks this once and turns the managed routing off (thresholds 0) when the two differ.
    // Many small inputs belong in one batch call instead (SnappyStreamChunkCodec, or snp_compress_batch from a device-resident caller).

    /// <summary>Thresholds for "one call must be faster than one managed thread" (4 MiB / 4 MiB).</summary>
    public static void UsePerCallLatencyRouting() { MinGpuCompressBytes = 4 << 20; MinGpuDecompressBytes = 4 << 20; }

    /// <summary>Thresholds for "the call must beat what the host's cores do in aggregate" (64 MiB / 128 MiB; the default).</summary>
    public static void UseHostThroughputRouting() { MinGpuCompressBytes = 64 << 20; MinGpuDecompressBytes = 128 << 20; }

    /// <summary>Inputs shorter than this are compressed by managed Snappier (default 64 MiB; 0 = always the GPU).</summary>
    public static int MinGpuCompressBytes { get; set; } = 64 << 20;

    /// <summary>Blocks that declare fewer bytes than this are decompressed by managed Snappier (default 128 MiB; 0 = always the GPU).</summary>
    public static int MinGpuDecompressBytes { get; set; } = 128 << 20;

    /// <summary>Compresses a probe input on both paths; if the bytes differ (the managed runtime uses another TableEntry hash than the GPU
    /// context) every size goes to the GPU from now on, so that output never depends on input size.  Returns whether they matched.</summary>
    public static bool AssertHashMatchesManaged()
    {
        if (!GpuContext.IsAvailable) return true;
        byte[] probe = new byte[70000];
        for (int i = 0; i < probe.Length; ++i) probe[i] = (byte)((i * 31 + (i >> 7)) & 0x3f);
        // the probe goes to the GPU directly: the routing thresholds are shared state and are not touched (a concurrent caller, or an exception
        // here, must not find them at 0)
        byte[] gpuBuf = new byte[GetMaxCompressedLength(probe.Length)];
        int gpuLen;
        fixed (byte* pin = probe)
        fixed (byte* pout = gpuBuf)
        {
            ThrowIfFailed(NativeMethods.snp_try_compress(GpuContext.Current.Handle, pin, (nuint)probe.Length, pout, (nuint)gpuBuf.Length, out nuint w));
            gpuLen = checked((int)w);
        }
        byte[] managed = global::Snappier.Snappy.CompressToArray(probe);
        bool same = gpuBuf.AsSpan(0, gpuLen).SequenceEqual(managed);
        if (!same) { MinGpuCompressBytes = 0; MinGpuDecompressBytes = 0; }
        return same;
    }

    private static bool UseGpuForCompress(long inputLength) => inputLength >= MinGpuCompressBytes && GpuAvailable();

    // GpuContext.IsAvailable never throws (GpuContext.Create swallows a missing library) and remembers a failure; the guard here is for
    // hosts where even the type initialiser of NativeMethods cannot run.
    private static bool GpuAvailable()
    {
        try { return GpuContext.IsAvailable; }
        catch (DllNotFoundException) { return false; }
        catch (EntryPointNotFoundException) { return false; }
        catch (TypeInitializationException) { return false; }
    }

    private static bool UseGpuForDecompress(ReadOnlySpan<byte> input)
    {
        if (!GpuAvailable()) return false;                      // first: nothing below may touch the native library on a host without it
        if (MinGpuDecompressBytes > 0)
        {
            // the declared length decides (VarIntEncoding.Read.cs:38-79, restated in managed code): a malformed preamble goes to the
            // managed path, which throws the reference's own exception for it
            if (!TryReadDeclaredLength(input, out uint declared)) return false;
            if (declared < (uint)MinGpuDecompressBytes) return false;
        }
        return true;
    }

    private static bool TryReadDeclaredLength(ReadOnlySpan<byte> input, out uint value)
    {
        value = 0;
        int shift = 0;
        for (int i = 0; i < 5 && i < input.Length; ++i)
        {
            uint b = input[i];
            uint v = b & 0x7fu;
            if (shift == 28 && v > 0xfu) return false;          // the fifth byte may only carry four bits (Helpers.LeftShiftOverflows)
            value |= v << shift;
            if (b < 128) return true;
            shift += 7;
        }
        return false;
    }

    /// <summary>Snappy.GetMaxCompressedLength (Snappy.cs:20-24).</summary>
    public static int GetMaxCompressedLength(int inputLength)
    {
        // Helpers.MaxCompressedLength + VarIntEncoding.MaxLength (Snappy.cs:20-24, Helpers.cs:17-46), in managed arithmetic: the same value as
        // snp_max_compressed_length (tests/test_capi_cpu.py pins the native one to the formula), and no P/Invoke on a host without the library --
        // CompressToMemory calls this before any routing decision.
        if (inputLength < 0) throw new ArgumentOutOfRangeException(nameof(inputLength));
        long v = 32L + inputLength + inputLength / 6 + 1 + 5;
        if (v > int.MaxValue) throw new ArgumentOutOfRangeException(nameof(inputLength));
        return (int)v;
    }

    /// <summary>Snappy.Compress (Snappy.cs:37-45): throws ArgumentException when the output span is too small.</summary>
    public static int Compress(ReadOnlySpan<byte> input, Span<byte> output)
    {
        if (!TryCompress(input, output, out int written))
            throw new ArgumentException("Output buffer is too small.", nameof(output));
        return written;
    }

    /// <summary>Snappy.TryCompress (Snappy.cs:55-67).</summary>
    public static bool TryCompress(ReadOnlySpan<byte> input, Span<byte> output, out int bytesWritten)
    {
        bytesWritten = 0;
        if (!UseGpuForCompress(input.Length)) return global::Snappier.Snappy.TryCompress(input, output, out bytesWritten);
        if (output.IsEmpty) return false;                                        // Snappy.cs:57-62
        fixed (byte* pin = input)
        fixed (byte* pout = output)
        {
            SnpStatus st = NativeMethods.snp_try_compress(GpuContext.Current.Handle, pin, (nuint)input.Length, pout, (nuint)output.Length, out nuint written);
            if (st == SnpStatus.OutputTooSmall) return false;                    // SnappyCompressor.cs:63-68
            ThrowIfFailed(st);
            bytesWritten = checked((int)written);
            return true;
        }
    }

    /// <summary>Snappy.CompressToMemory (Snappy.cs:99-112): the result is rented from ArrayPool and owned by the caller.</summary>
    public static IMemoryOwner<byte> CompressToMemory(ReadOnlySpan<byte> input)
    {
        byte[] buffer = ArrayPool<byte>.Shared.Rent(GetMaxCompressedLength(input.Length));
        try
        {
            int length = Compress(input, buffer);
            return new PooledOwner(buffer, length);
        }
        catch
        {
            ArrayPool<byte>.Shared.Return(buffer);
            throw;
        }
    }

    /// <summary>Snappy.CompressToArray (Snappy.cs:123-134).</summary>
    public static byte[] CompressToArray(ReadOnlySpan<byte> input)
    {
        using IMemoryOwner<byte> owner = CompressToMemory(input);
        return owner.Memory.ToArray();
    }

    /// <summary>Snappy.Compress(ReadOnlySequence, IBufferWriter) (Snappy.cs:82-89).  The segments are pinned where they lie and
    /// uploaded one after the other by snp_try_compress_segments: the sequence is never flattened on the managed side.</summary>
    public static void Compress(ReadOnlySequence<byte> input, IBufferWriter<byte> output)
    {
        ArgumentNullException.ThrowIfNull(output);
        if (!UseGpuForCompress(input.Length)) { global::Snappier.Snappy.Compress(input, output); return; }
        int max = GetMaxCompressedLength(checked((int)input.Length));
        Span<byte> dest = output.GetSpan(max);
        SnpStatus st;
        nuint written;
        using (var segs = new PinnedSegments(input))
        fixed (byte* pout = dest)
            st = NativeMethods.snp_try_compress_segments(GpuContext.Current.Handle, segs.Pointers, segs.Lengths, (uint)segs.Count, pout, (nuint)dest.Length, out written);
        ThrowIfFailed(st);
        output.Advance(checked((int)written));
    }

    /// <summary>Snappy.GetUncompressedLength (Snappy.cs:142-143): InvalidDataException("Invalid stream length") on a bad preamble.</summary>
    public static int GetUncompressedLength(ReadOnlySpan<byte> input)
    {
        // managed (VarIntEncoding.Read.cs:16-79; the same rules as snp_get_uncompressed_length, which tests/test_capi_cpu.py holds to the
        // reference's KATs): nothing that only inspects a preamble may need the native library
        if (!TryReadDeclaredLength(input, out uint length) || length > int.MaxValue) throw new InvalidDataException("Invalid stream length");
        return (int)length;
    }

    /// <summary>Snappy.Decompress (Snappy.cs:153-162).</summary>
    public static int Decompress(ReadOnlySpan<byte> input, Span<byte> output)
    {
        if (!TryDecompress(input, output, out int written))
            throw new ArgumentException("Output buffer is too small.", nameof(output));
        return written;
    }

    /// <summary>Snappy.TryDecompress (Snappy.cs:172-186): false only when the output span is too small; corrupt data throws.</summary>
    public static bool TryDecompress(ReadOnlySpan<byte> input, Span<byte> output, out int bytesWritten)
    {
        bytesWritten = 0;
        if (!UseGpuForDecompress(input)) return global::Snappier.Snappy.TryDecompress(input, output, out bytesWritten);
        fixed (byte* pin = input)
        fixed (byte* pout = output)
        {
            byte dummy = 0;
            SnpStatus st = NativeMethods.snp_try_decompress(GpuContext.Current.Handle, pin, (nuint)input.Length,
                                                            output.IsEmpty ? &dummy : pout, (nuint)output.Length, out nuint written);
            if (st == SnpStatus.OutputTooSmall) return false;
            ThrowIfFailed(st);
            bytesWritten = checked((int)written);
            return true;
        }
    }

    /// <summary>Snappy.DecompressToMemory (Snappy.cs:223-235).</summary>
    public static IMemoryOwner<byte> DecompressToMemory(ReadOnlySpan<byte> input)
    {
        int length = GetUncompressedLength(input);
        byte[] buffer = ArrayPool<byte>.Shared.Rent(Math.Max(length, 1));
        try
        {
            if (!TryDecompress(input, buffer.AsSpan(0, length), out int written) || written != length)
                throw new InvalidDataException("Incomplete Snappy block.");      // Snappy.cs:229-232
            return new PooledOwner(buffer, length);
        }
        catch
        {
            ArrayPool<byte>.Shared.Return(buffer);
            throw;
        }
    }

    /// <summary>Snappy.DecompressToMemory(ReadOnlySequence) (Snappy.cs:246-262): segments pinned in place, snp_try_decompress_segments.</summary>
    public static IMemoryOwner<byte> DecompressToMemory(ReadOnlySequence<byte> input)
    {
        if (input.IsSingleSegment) return DecompressToMemory(input.FirstSpan);
        Span<byte> head = stackalloc byte[NativeMethods.VarintMax];
        int headLen = (int)Math.Min(input.Length, head.Length);
        input.Slice(0, headLen).CopyTo(head);
        int length = GetUncompressedLength(head.Slice(0, headLen));              // InvalidDataException on a bad preamble
        if (length < MinGpuDecompressBytes || !GpuContext.IsAvailable) return global::Snappier.Snappy.DecompressToMemory(input);
        byte[] buffer = ArrayPool<byte>.Shared.Rent(Math.Max(length, 1));
        try
        {
            SnpStatus st;
            nuint written;
            using (var segs = new PinnedSegments(input))
            fixed (byte* pout = buffer)
                st = NativeMethods.snp_try_decompress_segments(GpuContext.Current.Handle, segs.Pointers, segs.Lengths, (uint)segs.Count, pout, (nuint)length, out written);
            ThrowIfFailed(st);
            if ((int)written != length) throw new InvalidDataException("Incomplete Snappy block.");      // Snappy.cs:229-232
            return new PooledOwner(buffer, length);
        }
        catch
        {
            ArrayPool<byte>.Shared.Return(buffer);
            throw;
        }
    }

    /// <summary>Snappy.Decompress(ReadOnlySequence, IBufferWriter) (Snappy.cs:194-212).</summary>
    public static void Decompress(ReadOnlySequence<byte> input, IBufferWriter<byte> output)
    {
        ArgumentNullException.ThrowIfNull(output);
        using IMemoryOwner<byte> owner = DecompressToMemory(input);
        owner.Memory.Span.CopyTo(output.GetSpan(owner.Memory.Length));
        output.Advance(owner.Memory.Length);
    }

    /// <summary>Snappy.DecompressToArray (Snappy.cs:273-283).</summary>
    public static byte[] DecompressToArray(ReadOnlySpan<byte> input)
    {
        using IMemoryOwner<byte> owner = DecompressToMemory(input);
        return owner.Memory.ToArray();
    }

    /// <summary>Status code -> the exception the reference throws for the same condition (ThrowHelper.cs:8-36).</summary>
    internal static void ThrowIfFailed(SnpStatus st)
    {
        switch (st)
        {
            case SnpStatus.Ok: return;
            case SnpStatus.OutputTooSmall: throw new ArgumentException("Output buffer is too small.");
            case SnpStatus.Overlap: throw new InvalidOperationException("Input and output spans must not overlap.");
            case SnpStatus.BadOffset:
            case SnpStatus.TooLong:
            case SnpStatus.Incomplete:
            case SnpStatus.BadLength:
            case SnpStatus.CrcMismatch:
            case SnpStatus.ChunkType:
            case SnpStatus.TruncatedStream: throw new InvalidDataException(NativeMethods.StatusString(st));
            case SnpStatus.BadArg: throw new ArgumentException(NativeMethods.StatusString(st));
            default: throw new InvalidOperationException("libsnappier_hip: " + NativeMethods.StatusString(st) + " " + (GpuContext.TryGetCurrent(out GpuContext? c) ? c!.LastError : string.Empty));
        }
    }

    /// <summary>The segments of a ReadOnlySequence pinned for the duration of one native call: pointer and length arrays in
    /// unmanaged memory (what snp_try_*_segments takes), one MemoryHandle per segment.</summary>
    private sealed class PinnedSegments : IDisposable
    {
        private readonly MemoryHandle[] _handles;
        private readonly IntPtr _block;
        public int Count { get; }
        public byte** Pointers => (byte**)_block;
        public nuint* Lengths => (nuint*)((byte*)_block + Count * sizeof(IntPtr));

        public PinnedSegments(ReadOnlySequence<byte> input)
        {
            int n = 0;
            foreach (ReadOnlyMemory<byte> _ in input) ++n;
            Count = n;
            _handles = new MemoryHandle[n];
            _block = System.Runtime.InteropServices.Marshal.AllocHGlobal(Math.Max(1, n) * (sizeof(IntPtr) + sizeof(nuint)));
            int i = 0;
            try
            {
                foreach (ReadOnlyMemory<byte> m in input)
                {
                    _handles[i] = m.Pin();
                    Pointers[i] = (byte*)_handles[i].Pointer;
                    Lengths[i] = (nuint)m.Length;
                    ++i;
                }
            }
            catch
            {
                // Pin() of a custom MemoryManager may throw: release what was pinned so far and the native block (MemoryHandle.Dispose on a
                // default handle is a no-op, so the unpinned tail of _handles is safe to dispose too)
                Dispose();
                throw;
            }
        }

        public void Dispose()
        {
            foreach (MemoryHandle h in _handles) h.Dispose();
            System.Runtime.InteropServices.Marshal.FreeHGlobal(_block);
        }
    }

    private sealed class PooledOwner(byte[] buffer, int length) : IMemoryOwner<byte>
    {
        private byte[]? _buffer = buffer;
        public Memory<byte> Memory => _buffer is null ? throw new ObjectDisposedException(nameof(PooledOwner)) : _buffer.AsMemory(0, length);
        public void Dispose()
        {
            byte[]? b = _buffer;
            _buffer = null;
            if (b is not null) ArrayPool<byte>.Shared.Return(b);
        }
    }
}
