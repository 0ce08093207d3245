// The chunk step of SnappyStream on the GPU: what SnappyStreamCompressor.CompressBlock (SnappyStreamCompressor.cs:194-230)
// and the chunk branches of SnappyStreamDecompressor.Decompress (SnappyStreamDecompressor.cs:53-199) compute, for MANY
// chunks per call.  The Stream plumbing (buffering writes to 64 KiB, async I/O, leaveOpen) stays in the managed
// SnappyStream; it hands whole runs of chunks to these two methods.
using System;
using System.IO;

namespace Snappier.Gpu;

public static unsafe class SnappyStreamChunkCodec
{
    public const int StreamHeaderLength = NativeMethods.StreamHeaderLength;   // ff 06 00 00 "sNaPpY"

    /// <summary>Upper bound of EncodeChunks' output for n raw bytes (includes the 10-byte stream identifier).</summary>
    public static long GetMaxEncodedLength(long n) => NativeMethods.snp_frame_max_encoded_length(n);

    /// <summary>
    /// Frames raw[0..n) as consecutive chunks of at most 65536 raw bytes: [type][len24][masked CRC-32C][payload], payload
    /// compressed when that is smaller (type 0x00) else raw (type 0x01).  The output always starts with the stream identifier;
    /// a SnappyStream that has already written it drops the first StreamHeaderLength bytes (includeStreamHeader = false).
    /// </summary>
    public static int EncodeChunks(ReadOnlySpan<byte> raw, Span<byte> output, bool includeStreamHeader)
    {
        fixed (byte* pin = raw)
        fixed (byte* pout = output)
        {
            Snappy.ThrowIfFailed(NativeMethods.snp_frame_encode(GpuContext.Current.Handle, pin, (nuint)raw.Length, pout, (nuint)output.Length, out nuint written));
            int n = checked((int)written);
            if (includeStreamHeader) return n;
            output.Slice(StreamHeaderLength, n - StreamHeaderLength).CopyTo(output);
            return n - StreamHeaderLength;
        }
    }

    /// <summary>Sum of the chunks' declared lengths of a framed byte run (header walk only, no device work).</summary>
    public static long GetDecodedLength(ReadOnlySpan<byte> framed)
    {
        fixed (byte* pin = framed)
        {
            Snappy.ThrowIfFailed(NativeMethods.snp_frame_decoded_length(pin, (nuint)framed.Length, out ulong length));
            return checked((long)length);
        }
    }

    /// <summary>
    /// Decodes a run of complete chunks (stream identifier, padding and skippable chunks allowed anywhere): every data chunk is
    /// decompressed or copied and its CRC verified; the first failing chunk in stream order decides the exception, as in the
    /// sequential reference ("Chunk CRC mismatch.", "Unknown chunk type", "Invalid copy offset", ...).
    /// </summary>
    public static int DecodeChunks(ReadOnlySpan<byte> framed, Span<byte> output)
    {
        fixed (byte* pin = framed)
        fixed (byte* pout = output)
        {
            byte dummy = 0;
            Snappy.ThrowIfFailed(NativeMethods.snp_frame_decode(GpuContext.Current.Handle, pin, (nuint)framed.Length,
                                                                output.IsEmpty ? &dummy : pout, (nuint)output.Length, out nuint written));
            return checked((int)written);
        }
    }

    /// <summary>Crc32CAlgorithm.Compute + ApplyMask (Crc32CAlgorithm.cs:41-49,156-158) of one buffer.</summary>
    public static uint MaskedCrc32C(ReadOnlySpan<byte> data)
    {
        fixed (byte* pin = data)
        {
            Snappy.ThrowIfFailed(NativeMethods.snp_crc32c(GpuContext.Current.Handle, pin, (nuint)data.Length, 1, out uint crc));
            return crc;
        }
    }
}
