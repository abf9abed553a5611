cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_map or test_gpu_frame or capi" > gpurun_out/r08h_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08h_tests_k.log | cut -c1-300
AB_SPECS="d assoc_late splat_early" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08h ab2
AB_ARGS="--big --steps 60 --reps 3" AB_SPECS="d assoc_late splat_early" bash tools/gpu_visit.sh r08h_big ab2
bash tools/gpu_visit.sh r08h prof | grep "k_associate\|k_surface_splat" | cut -c1-50,150-260
