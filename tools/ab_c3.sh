# A/B of environment switches on the configs[2] shape, one box (run on the GPU box from the repo root): bash tools/ab_c3.sh "VAR=val" "VAR=val2" ...
B3="--config 3 --unique 16 --no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_c3 --steps 3 --warmup 1"
python bench.py $B3 > /dev/null 2>&1
for rep in 1 2; do
for v in "$@"; do
  env $v python bench.py $B3 2>/dev/null | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('$v', round(o['value'],1), o['ms_per_step'].__round__(2), o['stage_ms_per_launch']['pyramid'])"
done; done
