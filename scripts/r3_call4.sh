#!/bin/bash
# Round 3, GPU call 4: phase timers of the two decoder forms, new bench.py end to end, GPU suite on the rebuilt library.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=${TAG:-r03d}
for v in r2prof v4prof; do
  echo "== $v" >> gpurun_out/${T}_decode_phase_timers.txt
  SNAPPIER_HIP_LIB=$PWD/snappier_amd/variants/libsnappier_hip_$v.so BLOCKS=65536 timeout 300 python scripts/prof_decompress.py >> gpurun_out/${T}_decode_phase_timers.txt 2>&1
done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${T}_pytest.txt
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
cat gpurun_out/${T}_decode_phase_timers.txt | grep -v "per block$" ; tail -3 gpurun_out/${T}_pytest.txt; tail -c 2500 gpurun_out/${T}_bench.json; tail -3 gpurun_out/${T}_bench.err
