# kernel trace (per-launch CSV + stats) of the default ORB workload: bash tools/trace_orb.sh TAG
set -e
TAG=${1:-tr}
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_c"
python bench.py --steps 1 --warmup 1 $B > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o run -- python3 $R/bench.py --steps 5 --warmup 2 $B > $R/gpurun_out/${TAG}_stats.log 2>&1
echo done
