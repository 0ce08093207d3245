#!/bin/bash
# Builds snappier_amd/variants/libsnappier_hip_<name>.so with extra -D flags applied to ONE source (default decompress.hip); the other
# sources are compiled once into a cache of objects.    scripts/build_variant.sh <name> [-DFOO=1 ...]      SRC=compress_lanes.hip to vary another file
set -e
cd "$(dirname "$0")/.."
name=$1; shift
SRC=${SRC:-decompress.hip}
OBJ=${OBJ_CACHE:-/tmp/snp_obj}
mkdir -p $OBJ snappier_amd/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fconstexpr-steps=100000000 -Wno-sometimes-uninitialized -Wno-unused-function"
for f in decompress decompress_small tag_index compress_lanes compress_win crc32c framing frame_scan capi; do
  if [ "$f.hip" != "$SRC" ]; then
    if [ ! -f $OBJ/$f.o ] || [ snappier_amd/csrc/$f.hip -nt $OBJ/$f.o ] || [ snappier_amd/csrc/snp_device.h -nt $OBJ/$f.o ] || [ include/snappier_hip.h -nt $OBJ/$f.o ]; then
      /opt/rocm/bin/hipcc $FLAGS -c snappier_amd/csrc/$f.hip -o $OBJ/$f.o
    fi
  fi
done
/opt/rocm/bin/hipcc $FLAGS "$@" -c snappier_amd/csrc/$SRC -o $OBJ/variant_$name.o
objs=""
for f in decompress decompress_small tag_index compress_lanes compress_win crc32c framing frame_scan capi; do
  if [ "$f.hip" == "$SRC" ]; then objs="$objs $OBJ/variant_$name.o"; else objs="$objs $OBJ/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib $objs -o snappier_amd/variants/libsnappier_hip_$name.so
echo snappier_amd/variants/libsnappier_hip_$name.so
