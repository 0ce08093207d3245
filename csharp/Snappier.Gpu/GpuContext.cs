// One snp_ctx per calling thread: Snappier's static Snappy.* methods are re-entrant because they create a compressor
// per call (Snappier/Snappy.cs:64,174,225); here the per-thread context owns a HIP stream and its HBM scratch.
using System;
using System.Runtime.InteropServices;

namespace Snappier.Gpu;

public sealed class GpuContext : SafeHandle
{
    [ThreadStatic] private static GpuContext? t_current;

    /// <summary>Device ordinal and hash variant new per-thread contexts are created with.</summary>
    public static int DefaultDevice { get; set; }
    public static SnpHash DefaultHash { get; set; } = SnpHash.Crc32C;

    private GpuContext() : base(IntPtr.Zero, ownsHandle: true) { }

    public override bool IsInvalid => handle == IntPtr.Zero;

    /// <summary>False when no HIP device is usable: callers keep the managed Snappier path (there is no CPU fallback in the library).</summary>
    public static bool IsAvailable => TryGetCurrent(out _);

    public static GpuContext Current =>
        TryGetCurrent(out GpuContext? c) ? c! : throw new InvalidOperationException("libsnappier_hip: no HIP device (snp_ctx_create returned Device)");

    // A failed snp_ctx_create (library missing, no HIP device) is remembered for the process: every later routed call then costs one
    // volatile read instead of another P/Invoke that fails the same way.  ResetAvailability() forgets it (a device was added, tests).
    private static volatile bool s_unavailable;

    public static void ResetAvailability() => s_unavailable = false;

    public static bool TryGetCurrent(out GpuContext? ctx)
    {
        ctx = t_current;
        if (ctx is { IsInvalid: false, IsClosed: false }) return true;
        if (s_unavailable) { ctx = null; return false; }
        ctx = Create(DefaultDevice, DefaultHash, default, out bool permanent);
        t_current = ctx;
        if (ctx is null && permanent) s_unavailable = true;     // (a transient failure -- out of memory on one thread, a busy device -- is retried by the next call)
        return ctx is not null;
    }

    public static GpuContext? Create(int device, SnpHash hash, IntPtr hipStream = default) => Create(device, hash, hipStream, out _);

    /// <summary>permanent: the library or its entry point is missing, or snp_ctx_create answered Device / BadArg (no such device, no HIP runtime):
    /// nothing a retry would change.  Any other failure is left to the next call.</summary>
    private static GpuContext? Create(int device, SnpHash hash, IntPtr hipStream, out bool permanent)
    {
        permanent = false;
        try
        {
            SnpStatus st = NativeMethods.snp_ctx_create(device, (int)hash, hipStream, out IntPtr h);
            if (st == SnpStatus.Ok) { var c = new GpuContext(); c.SetHandle(h); return c; }
            permanent = st == SnpStatus.Device || st == SnpStatus.BadArg;
        }
        catch (DllNotFoundException) { permanent = true; }
        catch (EntryPointNotFoundException) { permanent = true; }
        return null;
    }

    public string LastError => Marshal.PtrToStringUTF8(NativeMethods.snp_ctx_last_error(handle)) ?? string.Empty;

    /// <summary>which: 0 large blocks decoded one wavefront per 64 KiB fragment, 1 fell back to one wavefront, 2/3 workspace probe.</summary>
    public ulong Counter(int which) => NativeMethods.snp_ctx_counter(handle, which);

    /// <summary>snp_ctx_set_option: e.g. (SnpOption.TableProbeMaxBytes, 32L &lt;&lt; 30) in a process that shares the GPU, or
    /// (SnpOption.DecodeLayout, 1) before a batch of 64 KiB blocks when the previous batch was small blocks.</summary>
    public void SetOption(SnpOption option, long value) => Snappy.ThrowIfFailed(NativeMethods.snp_ctx_set_option(handle, (int)option, value));

    public long GetOption(SnpOption option)
    {
        Snappy.ThrowIfFailed(NativeMethods.snp_ctx_get_option(handle, (int)option, out long v));
        return v;
    }

    /// <summary>snp_ctx_reserve_compress: builds the DEVICE's hash-table workspace (shared by every context on it) for batches of up to <paramref name="fragments"/>
    /// 64 KiB fragments now -- call once at service start-up, before other allocations crowd the device, so that the first large
    /// request does not pay for the placement search.</summary>
    public void ReserveCompress(uint fragments) => Snappy.ThrowIfFailed(NativeMethods.snp_ctx_reserve_compress(handle, fragments));

    public void Synchronize() => Snappy.ThrowIfFailed(NativeMethods.snp_ctx_synchronize(handle));

    internal IntPtr Handle => handle;

    protected override bool ReleaseHandle()
    {
        NativeMethods.snp_ctx_destroy(handle);
        return true;
    }
}
