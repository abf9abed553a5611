cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_map or test_gpu_frame or capi" > gpurun_out/r08f_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08f_tests_k.log | cut -c1-300
AB_SPECS="d pre_pertap" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08f ab2
AB_ARGS="--big --steps 60 --reps 2" AB_SPECS="d pre_pertap" bash tools/gpu_visit.sh r08f_big ab2
bash tools/gpu_visit.sh r08f prof | head -12
