// Snappier.Gpu.Snappy -- the block API of Snappier.Snappy (Snappier/Snappy.cs:20-283) with the codec work done by
// libsnappier_hip.so.  Same method names, argument meaning, return values and exception types/messages; allocation
// (arrays, ArrayPool owners) stays managed, the native side never keeps a pointer past the call.
using System;
using System.Buffers;
using System.IO;

namespace Snappier.Gpu;

public static unsafe class Snappy
{
    /// <summary>Snappy.GetMaxCompressedLength (Snappy.cs:20-24).</summary>
    public static int GetMaxCompressedLength(int inputLength)
    {
        long v = NativeMethods.snp_max_compressed_length(inputLength);
        if (v < 0) throw new ArgumentOutOfRangeException(nameof(inputLength));
        return checked((int)v);
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

    /// <summary>Snappy.Compress(ReadOnlySequence, IBufferWriter) (Snappy.cs:82-89): segments are gathered, then one call.</summary>
    public static void Compress(ReadOnlySequence<byte> input, IBufferWriter<byte> output)
    {
        ArgumentNullException.ThrowIfNull(output);
        byte[] flat = ArrayPool<byte>.Shared.Rent(checked((int)input.Length));
        try
        {
            input.CopyTo(flat);
            int max = GetMaxCompressedLength((int)input.Length);
            int written = Compress(flat.AsSpan(0, (int)input.Length), output.GetSpan(max));
            output.Advance(written);
        }
        finally { ArrayPool<byte>.Shared.Return(flat); }
    }

    /// <summary>Snappy.GetUncompressedLength (Snappy.cs:142-143): InvalidDataException("Invalid stream length") on a bad preamble.</summary>
    public static int GetUncompressedLength(ReadOnlySpan<byte> input)
    {
        fixed (byte* pin = input)
        {
            SnpStatus st = NativeMethods.snp_get_uncompressed_length(pin, (nuint)input.Length, out uint length, out _);
            ThrowIfFailed(st);
            if (length > int.MaxValue) throw new InvalidDataException("Invalid stream length");
            return (int)length;
        }
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

    /// <summary>Snappy.DecompressToMemory(ReadOnlySequence) (Snappy.cs:246-262).</summary>
    public static IMemoryOwner<byte> DecompressToMemory(ReadOnlySequence<byte> input)
    {
        if (input.IsSingleSegment) return DecompressToMemory(input.FirstSpan);
        byte[] flat = ArrayPool<byte>.Shared.Rent(checked((int)input.Length));
        try
        {
            input.CopyTo(flat);
            return DecompressToMemory(flat.AsSpan(0, (int)input.Length));
        }
        finally { ArrayPool<byte>.Shared.Return(flat); }
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
