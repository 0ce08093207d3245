#!/bin/bash
# Builds snappier_amd/variants/libsnappier_hip_<name>.so: the product sources with extra -D flags applied to ONE of them (default
# decode_chains.hip); the other sources are compiled once into a cache of objects.
#     scripts/build_variant.sh <name> [-DFOO=1 ...]        SRC=compress_lanes.hip to vary another file
#     LAB=1 scripts/build_variant.sh lab                   the LAB library: scripts/lab/decompress_r04.hip in place of decompress.hip (every decoder front
#                                                          end that was measured and lost), and -DSNAPPIER_HIP_DEBUG_ENV on every source (the
#                                                          SNAPPIER_HIP_* knobs act: the product library reads no environment)
#     LAB=1 CLAB=1 SRC=compress_lanes.hip scripts/build_variant.sh clablate -DSNP_CL_ABLATE_RT=1
#                                                          ... with scripts/lab/compress_lanes_r04.hip (the lane compressor with its timing-only ablations)
#     PATCH=scripts/lab_patches/decode_chains_prof.patch scripts/build_variant.sh prof -DSNP_DC_PROF=1
#                                                          the varied source is a patched COPY (/tmp): instrumentation that never enters the product
#                                                          source (per-phase shader-clock budget: scripts/r5_decode_prof.py; -DSNP_DC_ABL=mask: timing-only removals)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
SRC=${SRC:-decode_chains.hip}
LAB=${LAB:-0}
OBJ=${OBJ_CACHE:-/tmp/snp_obj}
if [ "$LAB" = 1 ]; then OBJ=${OBJ}_lab; fi
mkdir -p $OBJ snappier_amd/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fconstexpr-steps=100000000 -Wno-sometimes-uninitialized -Wno-unused-function -Isnappier_amd/csrc"
DEC=decompress
CL=compress_lanes
if [ "$LAB" = 1 ]; then FLAGS="$FLAGS -DSNAPPIER_HIP_DEBUG_ENV"; DEC=../../scripts/lab/decompress_r04; fi
if [ "${CLAB:-0}" = 1 ]; then CL=../../scripts/lab/compress_lanes_r04; fi
if [ "$SRC" = decompress.hip ] && [ "$LAB" = 1 ]; then SRC=../../scripts/lab/decompress_r04.hip; fi
if [ "$SRC" = compress_lanes.hip ] && [ "${CLAB:-0}" = 1 ]; then SRC=../../scripts/lab/compress_lanes_r04.hip; fi
FILES="decode_chains $DEC decompress_small tag_index $CL compress_win crc32c framing frame_scan capi_ctx capi_pool capi_batch capi_host capi_frame"
newest_header=$(ls -t snappier_amd/csrc/*.h include/*.h | head -1)
for f in $FILES; do
  o=$OBJ/$(basename $f).o
  if [ "$f.hip" != "$SRC" ]; then
    if [ ! -f $o ] || [ snappier_amd/csrc/$f.hip -nt $o ] || [ $newest_header -nt $o ]; then
      /opt/rocm/bin/hipcc $FLAGS -c snappier_amd/csrc/$f.hip -o $o
    fi
  fi
done
VSRC=snappier_amd/csrc/$SRC
if [ -n "${PATCH:-}" ]; then
  VSRC=$OBJ/patched_${name}_$(basename $SRC)
  cp snappier_amd/csrc/$SRC $VSRC
  patch -s $VSRC < $PATCH
  FLAGS="$FLAGS -Isnappier_amd/csrc"
fi
/opt/rocm/bin/hipcc $FLAGS "$@" -c $VSRC -o $OBJ/variant_$name.o
objs=""
for f in $FILES; do
  if [ "$f.hip" == "$SRC" ]; then objs="$objs $OBJ/variant_$name.o"; else objs="$objs $OBJ/$(basename $f).o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib $objs -o snappier_amd/variants/libsnappier_hip_$name.so
echo snappier_amd/variants/libsnappier_hip_$name.so
