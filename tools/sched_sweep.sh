# RANSAC chunk-schedule sweep on one box (experiment hook RPE_RANSAC_SCHEDULE)
for s in "32,96,384,512" "32,64,416,512" "32,128,352,512" "32,96,512,512" "64,448,512" "32,480,512" "32,160,320,512" "32,96,384,512" "32,96,128,256,512" "96,416,512"; do
RPE_RANSAC_SCHEDULE=$s python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_ab > gpurun_out/sched.json 2> gpurun_out/sched.err || exit 1
python - <<PY
import json
d=json.loads(open('gpurun_out/sched.json').read().strip().splitlines()[-1])
print("$s", d['stage_ms_per_launch']['ransac'], d['ms_per_step'], d['median_rotation_error_deg'], d['pairs_ok'])
PY
done
