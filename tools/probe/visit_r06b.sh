cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or vs_reference" > gpurun_out/r06b_tests_frame.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r06b_tests_frame.log | cut -c1-300
timeout 200 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > gpurun_out/r06b_clocks_resident.jsonl 2>gpurun_out/r06b_clocks.err; cat gpurun_out/r06b_clocks_resident.jsonl
timeout 200 python tools/fast_clocks.py --nores elasticfusion_amd/libefusion_hip_clocks.so 140 > gpurun_out/r06b_clocks_streaming.jsonl 2>>gpurun_out/r06b_clocks.err; cat gpurun_out/r06b_clocks_streaming.jsonl
tail -3 gpurun_out/r06b_clocks.err
AB_SPECS="d d+nores" bash tools/gpu_visit.sh r06b ab2
