"""bench.py --gpus N without a launcher around it: the script starts its own N ranks (bench.launch_ranks), one process per GPU -- the
replacement of the reference's single-process nn.DataParallel (main_both.py:386-388).  On this one-GPU box the two ranks share the card
and gloo stands in for RCCL (MMRCA_DIST_BACKEND), as in the torchrun rehearsal of test_engine_gpu.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_starts_its_own_two_ranks_and_prints_one_line():
    env = dict(os.environ, MMRCA_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no_cpu_baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]       # ONE line on stdout: library chatter is relayed to stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["comm"]["ranks_seen"] == 2 and d["comm"]["bytes_per_step"] > 0
    assert d["config"]["global_batch"] == 512
