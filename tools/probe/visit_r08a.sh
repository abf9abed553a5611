cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_map or test_gpu_frame or bench_configs or steady or test_gpu_loop or capi" > gpurun_out/r08a_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08a_tests_k.log | cut -c1-300
AB_SPECS="d r6m" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08a ab2
AB_ARGS="--big --steps 60 --reps 2" AB_SPECS="d r6m" bash tools/gpu_visit.sh r08a_big ab2
bash tools/gpu_visit.sh r08a prof
