cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame" > gpurun_out/r08p_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08p_tests_k.log | cut -c1-300
AB_SPECS="d tilesfirst resolvetally" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08p ab2
AB_ARGS="--big --steps 60 --reps 3" AB_SPECS="d tilesfirst resolvetally" bash tools/gpu_visit.sh r08p_big ab2
