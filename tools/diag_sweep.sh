for lib in "$@"; do
RPE_LIB=$lib python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_ab > gpurun_out/diag.json 2> gpurun_out/diag.err || { tail -3 gpurun_out/diag.err; continue; }
python - <<PY
import json
d=json.loads(open('gpurun_out/diag.json').read().strip().splitlines()[-1])
print("$lib".split('/')[-1], round(d['stage_ms_per_launch']['match'],3))
PY
done
