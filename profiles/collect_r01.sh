set -e
R=$GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --data-cache /tmp/rpe_c > /dev/null 2>&1
python bench.py --steps 10 --warmup 3 --data-cache /tmp/rpe_c > $R/gpurun_out/r01c_bench.json 2> $R/gpurun_out/r01c_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c_stats -o run -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --data-cache /tmp/rpe_c > $R/gpurun_out/r01c_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01c_fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --data-cache /tmp/rpe_c > $R/gpurun_out/r01c_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01c_write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --data-cache /tmp/rpe_c > $R/gpurun_out/r01c_write.log 2>&1

# config 3 (HD SIFT + L2), 128 pairs in one launch group
cd $R
python bench.py --config 3 --batch 128 --sub-batch 128 --steps 1 --warmup 1 --no-cpu-baseline --data-cache /tmp/rpe_c3 > /dev/null 2>&1
python bench.py --config 3 --batch 128 --sub-batch 128 --steps 2 --warmup 1 --data-cache /tmp/rpe_c3 > $R/gpurun_out/r01c_c3_bench.json 2> $R/gpurun_out/r01c_c3_bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c_c3_stats -o run -- python3 $R/bench.py --config 3 --batch 128 --sub-batch 128 --steps 2 --warmup 1 --no-cpu-baseline --data-cache /tmp/rpe_c3 > $R/gpurun_out/r01c_c3_stats.log 2>&1
cd $R
python bench.py --stream --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r01c_stream_bench.json 2> $R/gpurun_out/r01c_stream_bench.err
python bench.py --streams 2 --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r01c_streams2_bench.json 2> $R/gpurun_out/r01c_streams2_bench.err
echo done2
