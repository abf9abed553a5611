cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or vs_reference or ops_tracking" > gpurun_out/r06d_tests_frame.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06d_tests_frame.log | cut -c1-300
timeout 200 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > gpurun_out/r06d_clocks_resident.jsonl 2>gpurun_out/r06d_clocks.err; cat gpurun_out/r06d_clocks_resident.jsonl
timeout 200 python tools/fast_clocks.py --nores elasticfusion_amd/libefusion_hip_clocks.so 140 > gpurun_out/r06d_clocks_streaming.jsonl 2>>gpurun_out/r06d_clocks.err; cat gpurun_out/r06d_clocks_streaming.jsonl
AB_SPECS="d d+nores d@0" bash tools/gpu_visit.sh r06d ab2
timeout 500 python tools/ab_bench.py --big --steps 60 --reps 2 d d+nores 2>&1 | grep -v "^$" | tee gpurun_out/r06d_ab_big.log | tail -6
