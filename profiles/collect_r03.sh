# Round-2 profile set (run on the GPU box from the repo root: bash profiles/collect_r03.sh TAG).  Tracing and
# counter collection are separate runs; every --pmc pass is its own run.  Outputs land in gpurun_out/TAG_*.
set -e
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_c"
python bench.py --steps 1 --warmup 1 $B > /dev/null 2>&1           # renders + caches the synthetic batch (forks happen here, not under the profiler)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -o run -- python3 $R/bench.py --steps 10 --warmup 3 $B > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 $B > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -o run -- python3 $R/bench.py --steps 2 --warmup 1 $B > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${TAG}_sq -o run -- python3 $R/bench.py --steps 2 --warmup 1 $B > $R/gpurun_out/${TAG}_sq.log 2>&1
echo done_c2

# BASELINE configs[2] shape: 128 HD pairs, SIFT(2048) + L2, one launch group (16 distinct rendered pairs tiled 8x)
cd $R
B3="--config 3 --unique 16 --no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_c3"
python bench.py --steps 1 --warmup 1 $B3 > /dev/null 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_c3_stats -o run -- python3 $R/bench.py --steps 2 --warmup 1 $B3 > $R/gpurun_out/${TAG}_c3_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_c3_fetch -o run -- python3 $R/bench.py --steps 1 --warmup 1 $B3 > $R/gpurun_out/${TAG}_c3_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_c3_write -o run -- python3 $R/bench.py --steps 1 --warmup 1 $B3 > $R/gpurun_out/${TAG}_c3_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${TAG}_c3_sq -o run -- python3 $R/bench.py --steps 1 --warmup 1 $B3 > $R/gpurun_out/${TAG}_c3_sq.log 2>&1
echo done_c3
