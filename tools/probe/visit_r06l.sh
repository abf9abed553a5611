cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for spec in d nolut; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_$spec -o st --output-format csv -- python $R/tools/ab_bench.py --steps 100 --reps 1 $spec 2>&1 | grep "rep 0"
  f=$(find /tmp/st_$spec -name "st_kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/r06l_kernel_stats_$spec.csv
done
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_map or test_gpu_frame" > gpurun_out/r06l_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06l_tests.log | cut -c1-300
