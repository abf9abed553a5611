cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_map or test_gpu_frame or test_gpu_steady or test_gpu_loop or bench_configs" > gpurun_out/r06j_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r06j_tests.log | cut -c1-300
AB_SPECS="d" bash tools/gpu_visit.sh r06j ab2
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_d -o st --output-format csv -- python $R/tools/ab_bench.py --steps 100 --reps 1 d > /dev/null 2>&1
f=$(find /tmp/st_d -name "st_kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/r06j_kernel_stats.csv
python - <<PY
import csv
for r in csv.DictReader(open('$f')):
    print(r['Name'][:70].replace('efm::(anonymous namespace)::','').replace('eft::(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs'])/1e3,2), r['MinNs'], r['MaxNs'])
PY
