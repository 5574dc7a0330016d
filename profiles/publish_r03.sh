# copies one collect_r03.sh TAG set from gpurun_out/ into profiles/r03_* and rebuilds r03_counters.json (run from the repo root)
set -e
T=${1:?tag}; G=gpurun_out; H=$(git rev-parse --short HEAD)
python profiles/make_counters.py profiles/r03_counters.json ORB_640x480_1000_1024 3 "1024 640x480 pairs, ORB 1000 + Hamming, one launch group (2048 images); commit $H" $G/${T}_fetch/run_counter_collection.csv $G/${T}_write/run_counter_collection.csv $G/${T}_sq/run_counter_collection.csv $G/${T}_stats/run_kernel_stats.csv
python profiles/make_counters.py profiles/r03_counters.json SIFT_1920x1080_2048_128 2 "128 1920x1080 pairs (16 distinct), SIFT cap 2048 + L2, one launch group (256 images); commit $H" $G/${T}_c3_fetch/run_counter_collection.csv $G/${T}_c3_write/run_counter_collection.csv $G/${T}_c3_sq/run_counter_collection.csv $G/${T}_c3_stats/run_kernel_stats.csv
cp $G/${T}_stats/run_kernel_stats.csv profiles/r03_kernel_stats.csv; cp $G/${T}_c3_stats/run_kernel_stats.csv profiles/r03_config3_kernel_stats.csv
cp $G/${T}_fetch/run_counter_collection.csv profiles/r03_pmc_fetch_size.csv; cp $G/${T}_write/run_counter_collection.csv profiles/r03_pmc_write_size.csv; cp $G/${T}_sq/run_counter_collection.csv profiles/r03_pmc_sq.csv
cp $G/${T}_c3_fetch/run_counter_collection.csv profiles/r03_config3_pmc_fetch_size.csv; cp $G/${T}_c3_write/run_counter_collection.csv profiles/r03_config3_pmc_write_size.csv; cp $G/${T}_c3_sq/run_counter_collection.csv profiles/r03_config3_pmc_sq.csv
