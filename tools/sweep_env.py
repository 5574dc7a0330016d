#!/usr/bin/env python3
"""interleaved bench runs of ONE build under different values of an environment switch, on one box:
    python tools/sweep_env.py VAR v1,v2,... rounds [bench args ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var, vals, rounds, extra = sys.argv[1], sys.argv[2].split(","), int(sys.argv[3]), sys.argv[4:]
res = {v: [] for v in vals}
for _ in range(rounds):
    for v in vals:
        env = dict(os.environ); env[var] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--no-calibrate"] + extra,
                             env=env, capture_output=True, text=True, check=True)
        res[v].append(json.loads(out.stdout.strip().splitlines()[-1]))
stages = list(res[vals[0]][0]["stage_ms_per_launch"])
print(f"{var:12s}", *[f"{v:>10s}" for v in vals], "  (min over rounds)")
for s in stages + ["ms_per_step"]:
    print(f"{s:12s}", *[f"{min((r['stage_ms_per_launch'][s] if s != 'ms_per_step' else r['ms_per_step']) for r in res[v]):10.3f}" for v in vals])
