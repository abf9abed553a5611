cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or steady or test_gpu_loop or reloc or bench_configs or one_frame" > gpurun_out/r08o_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08o_tests_k.log | cut -c1-300
AB_SPECS="d resolvetally" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08o ab2
AB_ARGS="--big --steps 60 --reps 3" AB_SPECS="d resolvetally" bash tools/gpu_visit.sh r08o_big ab2
bash tools/gpu_visit.sh r08o prof > /dev/null
