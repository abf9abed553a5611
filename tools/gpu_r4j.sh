#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 120 python tools/debug_fast.py 6 > $out/r04j_debug_fast.log 2>&1; grep "frames differ" $out/r04j_debug_fast.log
timeout 100 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > $out/r04j_fast_clocks.jsonl 2>$out/r04j_fast_clocks.err; cat $out/r04j_fast_clocks.jsonl
timeout 200 python -m pytest tests/test_gpu_fallback.py tests/test_gpu_frame.py -m gpu -q --timeout=150 -k "fallback or two_contexts or persistent_and or 1280 or odd" > $out/r04j_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/r04j_gpu_tests.log
tail -12 $out/r04j_gpu_tests.log | cut -c1-300
