// Several GPUs behind ONE SnappyStream: the chunk step (SnappyStreamCompressor.cs:166-230, SnappyStreamDecompressor.cs:53-199) striped over a
// list of devices.  Chunks are independent and one process sees every GPU of the node, so no collective is needed: the host cuts the run into
// contiguous ranges of whole chunks, each range goes through its own GpuContext on its own thread, and the results are concatenated in range
// order (the "length directory" is the list of bytes each range produced).  `devices` may repeat an ordinal ({0, 0}: two contexts and two HIP
// streams on one GPU).  Nothing new in the C-ABI: the same snp_frame_encode / snp_frame_decode the single-device codec calls.
// Python twin: snappier_amd/multidevice.py (tests/test_gpu_multidevice.py runs it with [0, 0] against the oracle).
using System;
using System.Collections.Generic;
using System.Threading.Tasks;

namespace Snappier.Gpu;

public sealed unsafe class MultiDeviceChunkCodec : IDisposable
{
    private const int ChunkBytes = 65536;
    private readonly GpuContext[] _contexts;
    private readonly int _minChunksPerRange;

    /// <summary>Bytes each range produced in the last call, in range order.</summary>
    public IReadOnlyList<int> LastDirectory { get; private set; } = Array.Empty<int>();

    public MultiDeviceChunkCodec(IReadOnlyList<int> devices, SnpHash hash = SnpHash.Crc32C, int minChunksPerRange = 16)
    {
        if (devices is null || devices.Count == 0) throw new ArgumentException("at least one device ordinal", nameof(devices));
        _contexts = new GpuContext[devices.Count];
        for (int i = 0; i < devices.Count; i++)
            _contexts[i] = GpuContext.Create(devices[i], hash) ?? throw new InvalidOperationException($"libsnappier_hip: no context on device {devices[i]}");
        _minChunksPerRange = Math.Max(1, minChunksPerRange);
    }

    /// <summary>Contiguous ranges of whole chunks, one per context (fewer when the run is short).</summary>
    internal static (int First, int Count)[] SplitRanges(int chunks, int workers, int minChunks)
    {
        int k = Math.Min(workers, Math.Max(1, chunks / Math.Max(1, minChunks)));
        var r = new (int, int)[k];
        int first = 0;
        for (int i = 0; i < k; i++)
        {
            int cnt = chunks / k + (i < chunks % k ? 1 : 0);
            r[i] = (first, cnt);
            first += cnt;
        }
        return r;
    }

    /// <summary>Frames raw as consecutive chunks; the bytes are those of SnappyStreamChunkCodec.EncodeChunks on one device.</summary>
    public int EncodeChunks(ReadOnlyMemory<byte> raw, Memory<byte> output, bool includeStreamHeader)
    {
        int chunks = (raw.Length + ChunkBytes - 1) / ChunkBytes;
        var ranges = SplitRanges(Math.Max(chunks, 1), _contexts.Length, _minChunksPerRange);
        var parts = new byte[ranges.Length][];
        var lens = new int[ranges.Length];
        Parallel.For(0, ranges.Length, new ParallelOptions { MaxDegreeOfParallelism = ranges.Length }, i =>
        {
            int lo = ranges[i].First * ChunkBytes, hi = Math.Min(raw.Length, (ranges[i].First + ranges[i].Count) * ChunkBytes);
            ReadOnlySpan<byte> piece = raw.Span.Slice(lo, Math.Max(hi - lo, 0));
            var buf = new byte[checked((int)SnappyStreamChunkCodec.GetMaxEncodedLength(piece.Length))];
            fixed (byte* pin = piece)
            fixed (byte* pout = buf)
            {
                Snappy.ThrowIfFailed(NativeMethods.snp_frame_encode(_contexts[i].Handle, pin, (nuint)piece.Length, pout, (nuint)buf.Length, out nuint w));
                lens[i] = checked((int)w);
            }
            parts[i] = buf;
        });
        int at = 0;
        var dir = new int[ranges.Length];
        for (int i = 0; i < ranges.Length; i++)
        {
            int skip = (i > 0 || !includeStreamHeader) ? SnappyStreamChunkCodec.StreamHeaderLength : 0;   // the identifier is written once  SnappyStreamCompressor.cs:148-157
            parts[i].AsSpan(skip, lens[i] - skip).CopyTo(output.Span.Slice(at));
            dir[i] = lens[i] - skip;
            at += dir[i];
        }
        LastDirectory = dir;
        return at;
    }

    /// <summary>Decodes a run of complete chunks; the first failing chunk in stream order decides the exception, as in the sequential reference.</summary>
    public int DecodeChunks(ReadOnlyMemory<byte> framed, Memory<byte> output)
    {
        // header walk on the host: 4 bytes per chunk (SnappyStreamDecompressor.cs:53-75)
        var offs = new List<int>();
        int pos = 0;
        ReadOnlySpan<byte> f = framed.Span;
        while (pos + 4 <= f.Length)
        {
            int size = f[pos + 1] | (f[pos + 2] << 8) | (f[pos + 3] << 16);
            if (pos + 4 + size > f.Length) break;
            offs.Add(pos);
            pos += 4 + size;
        }
        if (pos != f.Length || offs.Count <= _minChunksPerRange)      // short, or a malformed tail: one context reports exactly what the reference would
        {
            fixed (byte* pin = f)
            fixed (byte* pout = output.Span)
            {
                byte dummy = 0;
                Snappy.ThrowIfFailed(NativeMethods.snp_frame_decode(_contexts[0].Handle, pin, (nuint)f.Length, output.IsEmpty ? &dummy : pout, (nuint)output.Length, out nuint w));
                LastDirectory = new[] { checked((int)w) };
                return checked((int)w);
            }
        }
        offs.Add(pos);
        var ranges = SplitRanges(offs.Count - 1, _contexts.Length, _minChunksPerRange);
        var outAt = new int[ranges.Length + 1];
        for (int i = 0; i < ranges.Length; i++)
        {
            int lo = offs[ranges[i].First], hi = offs[ranges[i].First + ranges[i].Count];
            outAt[i + 1] = outAt[i] + checked((int)SnappyStreamChunkCodec.GetDecodedLength(f.Slice(lo, hi - lo)));
        }
        if (outAt[ranges.Length] > output.Length) throw new ArgumentException("Output buffer is too small.", nameof(output));
        var status = new SnpStatus[ranges.Length];
        var dir = new int[ranges.Length];
        Parallel.For(0, ranges.Length, new ParallelOptions { MaxDegreeOfParallelism = ranges.Length }, i =>
        {
            int lo = offs[ranges[i].First], hi = offs[ranges[i].First + ranges[i].Count];
            fixed (byte* pin = framed.Span.Slice(lo, hi - lo))
            fixed (byte* pout = output.Span.Slice(outAt[i], outAt[i + 1] - outAt[i]))
            {
                byte dummy = 0;
                status[i] = NativeMethods.snp_frame_decode(_contexts[i].Handle, pin, (nuint)(hi - lo), outAt[i + 1] == outAt[i] ? &dummy : pout,
                                                           (nuint)(outAt[i + 1] - outAt[i]), out nuint w);
                dir[i] = checked((int)w);
            }
        });
        foreach (SnpStatus st in status) Snappy.ThrowIfFailed(st);      // range order = stream order
        LastDirectory = dir;
        return outAt[ranges.Length];
    }

    public void Dispose()
    {
        foreach (GpuContext c in _contexts) c.Dispose();
    }
}
