cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 500 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or fallback or vs_reference" > gpurun_out/r08j_tests_k.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r08j_tests_k.log | cut -c1-300
timeout 200 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so elasticfusion_amd/libefusion_hip_norepl_clocks.so 140 > gpurun_out/r08j_clocks.jsonl 2>gpurun_out/r08j_clocks.err; cat gpurun_out/r08j_clocks.jsonl | cut -c1-1200
AB_SPECS="d norepl" AB_ARGS="--reps 4" bash tools/gpu_visit.sh r08j ab2
AB_ARGS="--big --steps 60 --reps 2" AB_SPECS="d norepl" bash tools/gpu_visit.sh r08j_big ab2
