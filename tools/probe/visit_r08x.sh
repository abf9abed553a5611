cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_map or test_gpu_frame" > gpurun_out/r08x_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08x_tests_k.log | cut -c1-300
AB_SPECS="d splat1" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08x ab2
AB_ARGS="--big --steps 60 --reps 3" AB_SPECS="d splat1" bash tools/gpu_visit.sh r08x_big ab2
bash tools/gpu_visit.sh r08x prof | grep "k_surface_splat" | cut -c1-60,180-260
