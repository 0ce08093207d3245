"""bench.py through its one-process-per-GPU path on the one-GPU box: launched by torch.distributed.run with ONE rank, so the RCCL process group, the
barriers, the max-over-ranks timing and the length/status directory gather all execute (with N = 1 they are trivial, but they are the code the
driver runs at N = 2, 4, 8).  A small batch, no CPU baseline, no PMC child runs."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_bench_runs_under_torch_distributed_run_with_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_NO_PLAIN="1", BENCH_NO_DEFAULT_SEARCH="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--blocks", "16384", "--no-cpu-baseline", "--no-live-traffic",
           "--no-other-configs", "--config5-lines"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["verified"].startswith("decode(encode(x)) == x")
    assert line["config"]["blocks_per_gpu"] == 16384 and "16384" in line["config"]["workload"]
    c5 = line["config5_lines"]                                      # the mixed-corpus lines with the directory gather and the payload gather
    assert c5["n_gpus"] == 1 and c5["codec_plus_directory_GBps"] > 0 and c5["verified"]
    assert set(line["per_rank"]) >= {"compress_ms", "decompress_ms", "directory_gather_ms"}
