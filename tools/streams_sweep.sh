for s in 1 2 3 4 6 8; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_ab --streams $s > gpurun_out/st_$s.json 2> gpurun_out/st_$s.err || exit 1
python - <<PY
import json
d=json.loads(open('gpurun_out/st_$s.json').read().strip().splitlines()[-1])
print($s, d['value'], d['ms_per_step'])
PY
done
