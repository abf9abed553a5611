cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or ops_tracking or vs_reference or test_gpu_loop or test_gpu_global or gpu_replay or bench_configs" > gpurun_out/r06r_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06r_tests.log | cut -c1-300
AB_SPECS="d d+ov1" bash tools/gpu_visit.sh r06r ab2
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_d -o st --output-format csv -- python $R/tools/ab_bench.py --steps 100 --reps 1 d 2>&1 | grep "rep 0"
cp $(find /tmp/st_d -name "st_kernel_stats.csv" | head -1) $R/gpurun_out/r06r_kernel_stats.csv
