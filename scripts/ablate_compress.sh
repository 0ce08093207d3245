#!/bin/bash
# Build compile-time variants of the compress kernel (here) and time each on the GPU box:
#   scripts/ablate_compress.sh build      (in the build container)
#   gpurun -- 'bash scripts/ablate_compress.sh run'
set -e
cd "$(dirname "$0")/.."
V=snappier_amd/variants
SRC="snappier_amd/csrc/decompress.hip snappier_amd/csrc/decompress_lanes.hip snappier_amd/csrc/tag_index.hip snappier_amd/csrc/compress.hip snappier_amd/csrc/compress_lanes.hip snappier_amd/csrc/crc32c.hip snappier_amd/csrc/framing.hip snappier_amd/csrc/capi.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fconstexpr-steps=100000000 -Wno-unused-function -Wl,-rpath,/opt/rocm/lib"
declare -A VARIANTS=(
  [stage0]="-DSNP_D_STAGE=0"
  [stage2048]="-DSNP_D_STAGE=2048"
  [stage2048w8]="-DSNP_D_STAGE=2048 -DSNP_D_WAVES=8"
  [stage1024w8]="-DSNP_D_STAGE=1024 -DSNP_D_WAVES=8"
  [stage3072w8]="-DSNP_D_STAGE=3072 -DSNP_D_WAVES=8"
)
if [ "$1" = prof ]; then
  mkdir -p $V
  /opt/rocm/bin/hipcc $FLAGS -DSNP_C_PROF=1 ${PROF_DEFS:-} $SRC -o $V/libsnappier_hip_prof.so
  ls -la $V/libsnappier_hip_prof.so
elif [ "$1" = build ]; then
  mkdir -p $V
  for k in "${!VARIANTS[@]}"; do
    /opt/rocm/bin/hipcc $FLAGS ${VARIANTS[$k]} $SRC -o $V/libsnappier_hip_$k.so &
  done
  wait
  ls -la $V
else
  mkdir -p gpurun_out
  for k in "${!VARIANTS[@]}"; do
    r=$(SNAPPIER_HIP_LIB=$PWD/$V/libsnappier_hip_$k.so timeout 120 python bench.py --blocks ${BLOCKS:-163840} --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['compress_GBps'], d['decompress_GBps'])")
    echo "$k $r" | tee -a gpurun_out/ablate_compress.log
  done
fi
