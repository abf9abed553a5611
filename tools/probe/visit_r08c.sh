cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_map or test_gpu_frame or steady" > gpurun_out/r08c_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08c_tests_k.log | cut -c1-300
AB_SPECS="d alltaps" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08c ab2
AB_ARGS="--big --steps 60 --reps 3" AB_SPECS="d alltaps" bash tools/gpu_visit.sh r08c_big ab2
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_d -o st --output-format csv -- python $R/tools/ab_bench.py --big --steps 40 --reps 1 d 2>&1 | grep "rep 0"
cp $(find /tmp/st_d -name "st_kernel_stats.csv" | head -1) $R/gpurun_out/r08c_1280x960_kernel_stats.csv
