"""cProfile of the host side of the configs[1] train step (where do the ~34 ms of Python per step go?).
python tools/host_profile.py [--steps 8] [--dtype bf16]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--sort", default="tottime")
    a = ap.parse_args()
    import bench
    sys.argv = ["bench.py", "--steps", str(a.steps), "--warmup", "3", "--no_cpu_baseline", "--dtype", a.dtype]
    pr = cProfile.Profile()
    # profile only hip_train_step calls: wrap it
    from garbage_classification_rca_amd import training as MB
    orig = MB.hip_train_step
    calls = {"n": 0, "t": 0.0}

    def wrapped(*args, **kw):
        calls["n"] += 1
        if calls["n"] > 3:
            t0 = time.perf_counter()
            pr.enable()
            try:
                return orig(*args, **kw)
            finally:
                pr.disable()
                calls["t"] += time.perf_counter() - t0
        return orig(*args, **kw)

    MB.hip_train_step = wrapped            # bench.main() imports it from the module at call time
    bench.main()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(a.sort).print_stats(35)
    print(s.getvalue(), file=sys.stderr)
    print(f"profiled calls: {calls['n'] - 3}, host s/step under the profiler: {calls['t'] / max(calls['n'] - 3, 1):.4f}", file=sys.stderr)


if __name__ == "__main__":
    main()
