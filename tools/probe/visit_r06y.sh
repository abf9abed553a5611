cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -q --timeout=300 -x -k "ops_map or test_gpu_frame or vs_reference or bench_configs or test_gpu_steady or gpu_replay" > gpurun_out/r06y_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06y_tests_k.log | cut -c1-300
AB_SPECS="d pre_vpair" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r06y ab2
AB_ARGS="--big --steps 60" AB_SPECS="d pre_vpair" bash tools/gpu_visit.sh r06y_big ab2
bash tools/gpu_visit.sh r06y prof
