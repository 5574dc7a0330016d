# ad-hoc: LDS / VALU counters of the config-3 kernels (run on the GPU box from the repo root)
set -e
TAG=${1:-c3lds}
R=$GRAFT_REPO_ROOT
B3="--config 3 --unique 16 --no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_c3"
python bench.py --steps 1 --warmup 1 $B3 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/${TAG}_sq -o run -- python3 $R/bench.py --steps 1 --warmup 1 $B3 > $R/gpurun_out/${TAG}_sq.log 2>&1
echo done
