cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --timeout=300 -x -k "test_gpu_frame or vs_reference" > gpurun_out/r06o_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r06o_tests.log | cut -c1-300
timeout 200 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > gpurun_out/r06o_clocks.jsonl 2>gpurun_out/r06o_clocks.err; cat gpurun_out/r06o_clocks.jsonl
AB_SPECS="d" bash tools/gpu_visit.sh r06o ab2
