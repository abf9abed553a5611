cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > gpurun_out/r06e_clocks_resident.jsonl 2>gpurun_out/r06e_clocks.err; cat gpurun_out/r06e_clocks_resident.jsonl
timeout 200 python tools/fast_clocks.py --nores elasticfusion_amd/libefusion_hip_clocks.so 140 > gpurun_out/r06e_clocks_streaming.jsonl 2>>gpurun_out/r06e_clocks.err; cat gpurun_out/r06e_clocks_streaming.jsonl
AB_SPECS="d d+nores" bash tools/gpu_visit.sh r06e ab2
