#!/usr/bin/env python3
"""A/B of two builds of librpe_amd.so on ONE GPU box (MI355X_MICROARCH.md: never rank builds by timings taken on
different devices): alternates `bench.py` runs with RPE_LIB pointing at each library and prints the per-stage
milliseconds side by side.

    python tools/ab.py path/to/libA.so path/to/libB.so [rounds] [extra bench args ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(lib, extra):
    env = dict(os.environ, RPE_LIB=os.path.abspath(lib))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-extra",
                          "--no-calibrate", "--data-cache", "/tmp/rpe_ab"] + extra, env=env, capture_output=True, text=True, check=True)
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    extra = sys.argv[4:]
    res = {a: [], b: []}
    for _ in range(rounds):
        for lib in (a, b):
            res[lib].append(run(lib, extra))
    stages = list(res[a][0]["stage_ms_per_launch"])
    print(f"{'stage':12s} {'A min':>8s} {'A med':>8s} {'B min':>8s} {'B med':>8s}   (A = {a}, B = {b}, {rounds} rounds each, interleaved)")
    for s in stages + ["ms_per_step"]:
        va = sorted(r["stage_ms_per_launch"][s] if s != "ms_per_step" else r["ms_per_step"] for r in res[a])
        vb = sorted(r["stage_ms_per_launch"].get(s, float("nan")) if s != "ms_per_step" else r["ms_per_step"] for r in res[b])
        print(f"{s:12s} {va[0]:8.3f} {va[len(va) // 2]:8.3f} {vb[0]:8.3f} {vb[len(vb) // 2]:8.3f}")


if __name__ == "__main__":
    main()
