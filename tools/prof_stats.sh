# kernel-trace stats of the headline config (run on the GPU box from the repo root): bash tools/prof_stats.sh TAG [extra bench args]
set -e
TAG=${1:-x}; shift || true
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_c $@"
python bench.py --steps 1 --warmup 1 $B > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o run -- python3 $R/bench.py --steps 10 --warmup 3 $B > $R/gpurun_out/${TAG}_stats.log 2>&1
cd $R
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/${TAG}_stats/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:24]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), ('%.3f'%(float(r['TotalDurationNs'])/1e6)).rjust(10), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(10), r['Percentage'])
PY
