#!/usr/bin/env python3
"""interleaved bench runs of N builds on one box: python tools/abn.py rounds libA,libB,... [bench args]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds, libs, extra = int(sys.argv[1]), sys.argv[2].split(","), sys.argv[3:]
res = {l: [] for l in libs}
for _ in range(rounds):
    for l in libs:
        env = dict(os.environ, RPE_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extra", "--no-calibrate"] + extra,
                             env=env, capture_output=True, text=True, check=True)
        res[l].append(json.loads(out.stdout.strip().splitlines()[-1]))
stages = list(res[libs[0]][0]["stage_ms_per_launch"])
print(f"{'stage':12s}", *[os.path.basename(l).replace('librpe_', '').replace('.so', '')[:10].rjust(10) for l in libs], "  (min over rounds)")
for s in stages + ["ms_per_step"]:
    print(f"{s:12s}", *[f"{min((r['stage_ms_per_launch'][s] if s != 'ms_per_step' else r['ms_per_step']) for r in res[l]):10.3f}" for l in libs])
