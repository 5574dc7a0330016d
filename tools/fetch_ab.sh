# FETCH_SIZE per kernel for two library builds (same box): bash tools/fetch_ab.sh libA libB
set -e
R=$GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-extra --no-calibrate --data-cache /tmp/rpe_ab"
cd /tmp && export TMPDIR=/tmp
for L in $1 $2; do
  N=$(basename $L .so)
  RPE_LIB=$R/$L rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fetch_$N -o run -- python3 $R/bench.py --steps 2 --warmup 1 $B > $R/gpurun_out/fetch_$N.log 2>&1
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda:[0.0,0])
d=collections.defaultdict(float); nm={}
for r in csv.DictReader(open("$R/gpurun_out/fetch_$N/run_counter_collection.csv")):
    d[r["Dispatch_Id"]]+=float(r["Counter_Value"]); nm[r["Dispatch_Id"]]=r["Kernel_Name"].split("(")[0].replace("void ","").split("<")[0]
for k,v in d.items():
    acc[nm[k]][0]+=v; acc[nm[k]][1]+=1
print("$N", {k: round(v[0]/v[1]*2048/1e9,3) for k,v in acc.items() if k in ("orient_describe_kernel","harris_kernel","fast_nms_kernel","pyr_resize_kernel")}, "GB fetched per launch (x2 corrected)")
PY
done
