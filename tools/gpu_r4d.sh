#!/bin/bash
# round 4 visit d: phase clocks of the persistent tracker, the two repaired tests
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 120 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > $out/r04d_fast_clocks.jsonl 2>$out/r04d_fast_clocks.err; cat $out/r04d_fast_clocks.jsonl
timeout 400 python -m pytest tests/test_gpu_one_frame.py tests/test_gpu_frame.py -m gpu -q --timeout=380 -k "one_frame or checkpoint or two_contexts or persistent_and" --durations=5 > $out/r04d_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/r04d_gpu_tests.log
tail -25 $out/r04d_gpu_tests.log | cut -c1-400
