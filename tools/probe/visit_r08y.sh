cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_map or test_gpu_frame or steady" > gpurun_out/r08y_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08y_tests_k.log | cut -c1-300
AB_SPECS="d resolveall" AB_ARGS="--reps 4" bash tools/gpu_visit.sh r08y ab2
