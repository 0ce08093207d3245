"""Kernel layouts of the PRODUCT library (libsnappier_hip.so), selected per context through snp_ctx_set_option -- what the parity and fuzz tests
parametrise over.  No test sets a SNAPPIER_HIP_* variable: the product reads no environment, and every path it contains has an option."""
from snappier_amd import _native as N

COMPRESS_LAYOUTS = ["win", "win-np2", "wing", "wind", "lanes", "lanes-exact", "lanes-opts7", "lanes-opts31", "lanes-opts87-slots1", "lanes-opts215-slots1",
                    "lanes-opts151-slots2", "lanes-per16", "lanes-slots4"]
# chains = the default (decode_chains.hip after the small-block policy); wave-only = no pre-pass; serial = decompress.hip's tag-by-tag kernel;
# small* = every block through the small-block pre-pass first (decompress_small.hip), leftovers to the list kernel (or one workgroup each: -grid)
DECODE_LAYOUTS = ["chains", "wave-only", "serial", "small", "small-grid", "small-lanes", "small-team4", "small-team16"]


def set_compress_layout(ctx, layout: str):
    """win = one fragment per wavefront, table in LDS (-np2: two window positions per lane); wing = the same with the table in a global slot;
    wind = both forms side by side (two streams, one ticket counter);
    lanes = one fragment per lane, tables in the HBM workspace (-exact: exact-length stores only; -optsN: SNP_OPT_COMPRESS_LANE_STORES = N;
    -slotsN: N probes per trip; -perN: N fragments per wavefront)."""
    base = layout.split("-")[0]
    ctx.set_option(N.OPT_COMPRESS_LAYOUT, {"auto": N.COMPRESS_AUTO, "win": N.COMPRESS_WINDOW_LDS, "win2": N.COMPRESS_WINDOW_LDS,
                                           "wing": N.COMPRESS_WINDOW_GLOBAL, "wind": N.COMPRESS_WINDOW_DUAL, "lanes": N.COMPRESS_LANES}[base])
    ctx.set_option(N.OPT_COMPRESS_WINDOW_POSITIONS, 2 if (base == "win2" or "-np2" in layout) else 1)
    if layout.endswith("-exact"):
        ctx.set_option(N.OPT_COMPRESS_LANE_STORES, 0)
    for part in layout.split("-")[1:]:
        if part.startswith("opts"):
            ctx.set_option(N.OPT_COMPRESS_LANE_STORES, int(part[4:]))
        elif part.startswith("slots"):
            ctx.set_option(N.OPT_COMPRESS_LANE_PROBES, int(part[5:]))
        elif part.startswith("per"):
            ctx.set_option(N.OPT_COMPRESS_LANES_PER_WAVEFRONT, int(part[3:]))
    return ctx


def set_decode_layout(ctx, decode: str, fenced=None, small_max: int = 65536):
    if fenced is not None:
        ctx.set_option(N.OPT_FENCED, int(fenced))
    if decode == "chains":
        ctx.set_option(N.OPT_DECODE_LAYOUT, N.DECODE_AUTO)
    elif decode == "wave-only":
        ctx.set_option(N.OPT_DECODE_LAYOUT, N.DECODE_WAVE_ONLY)
    elif decode == "serial":
        ctx.set_option(N.OPT_DECODE_LAYOUT, N.DECODE_SERIAL)
    elif decode.startswith("small"):
        kind = decode.split("-")[1] if "-" in decode else ""
        ctx.set_option(N.OPT_DECODE_LAYOUT, {"": N.DECODE_AUTO, "grid": N.DECODE_AUTO, "lanes": N.DECODE_SMALL_LANES, "team4": N.DECODE_SMALL_TEAM4,
                                             "team8": N.DECODE_SMALL_TEAM8, "team16": N.DECODE_SMALL_TEAM16}[kind])
        ctx.set_option(N.OPT_SMALL_BLOCK_MIN_BATCH, 1)
        ctx.set_option(N.OPT_SMALL_BLOCK_MAX, small_max)
        ctx.set_option(N.OPT_DECODE_LEFTOVERS, 1 if kind == "grid" else 2)   # always the pre-pass, whatever the previous batch was like
    else:
        raise ValueError(decode)
    return ctx
