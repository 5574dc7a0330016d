# kernel stats + SQ counters of the configs[2] shape (run on the GPU box from the repo root): bash tools/prof_c3.sh TAG
set -e
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
B3="--config 3 --unique 16 --no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_c3"
python bench.py --steps 1 --warmup 1 $B3 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_c3_stats -o run -- python3 $R/bench.py --steps 2 --warmup 1 $B3 > $R/gpurun_out/${TAG}_c3_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/${TAG}_c3_sq -o run -- python3 $R/bench.py --steps 1 --warmup 1 $B3 > $R/gpurun_out/${TAG}_c3_sq.log 2>&1
cd $R
python - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/${TAG}_c3_stats/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), ('%.3f'%(float(r['TotalDurationNs'])/1e6)).rjust(10), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(10), r['Percentage'])
f=glob.glob('gpurun_out/${TAG}_c3_sq/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:44]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in acc.items():
    if 'sift' in k or 'match' in k: print(k.ljust(44), {a:round(b/1e6,1) for a,b in v.items()})
PY
