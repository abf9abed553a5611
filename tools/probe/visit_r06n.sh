cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_d -o st --output-format csv -- python $R/tools/ab_bench.py --steps 100 --reps 1 d 2>&1 | grep "rep 0"
cp $(find /tmp/st_d -name "st_kernel_stats.csv" | head -1) $R/gpurun_out/r06n_kernel_stats.csv
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/st_big -o st --output-format csv -- python $R/tools/ab_bench.py --big --steps 60 --reps 1 d 2>&1 | grep "rep 0"
cp $(find /tmp/st_big -name "st_kernel_stats.csv" | head -1) $R/gpurun_out/r06n_1280x960_kernel_stats.csv
cd $R
timeout 600 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_map or test_gpu_frame" > gpurun_out/r06n_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06n_tests.log | cut -c1-300
