cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 800 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_tracking or bench_configs or vs_reference or test_gpu_frame" > gpurun_out/r07c_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r07c_tests_k.log | cut -c1-300
AB_ARGS="--big --steps 60 --reps 3" AB_SPECS="d shallow d@0 shallow@0" bash tools/gpu_visit.sh r07c_big ab2
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_d -o st --output-format csv -- python $R/tools/ab_bench.py --big --steps 40 --reps 1 d d@0 2>&1 | grep "rep 0"
cp $(find /tmp/st_d -name "st_kernel_stats.csv" | head -1) $R/gpurun_out/r07c_1280x960_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_s -o st --output-format csv -- python $R/tools/ab_bench.py --big --steps 40 --reps 1 shallow shallow@0 2>&1 | grep "rep 0"
cp $(find /tmp/st_s -name "st_kernel_stats.csv" | head -1) $R/gpurun_out/r07c_1280x960_kernel_stats_shallow.csv
grep "se3_accum\|k_track_ref<" $R/gpurun_out/r07c_1280x960_kernel_stats.csv $R/gpurun_out/r07c_1280x960_kernel_stats_shallow.csv | cut -d, -f1-4 | cut -c1-260
