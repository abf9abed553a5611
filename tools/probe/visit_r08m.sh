cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or test_gpu_loop or capi or reloc" > gpurun_out/r08m_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08m_tests_k.log | cut -c1-300
AB_SPECS="d noarena" AB_ARGS="--reps 3" bash tools/gpu_visit.sh r08m ab2
AB_ARGS="--big --steps 60 --reps 6" AB_SPECS="d noarena" bash tools/gpu_visit.sh r08m_big ab2
