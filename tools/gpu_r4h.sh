#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 120 python tools/debug_fast.py 6 > $out/r04h_debug_fast.log 2>&1
grep "frames differ" $out/r04h_debug_fast.log
timeout 100 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > $out/r04h_fast_clocks.jsonl 2>$out/r04h_fast_clocks.err; cat $out/r04h_fast_clocks.jsonl
if [ "$(grep -c ', 0 frames differ' $out/r04h_debug_fast.log)" != "2" ]; then echo "parity broken: suite skipped"; grep DIFF $out/r04h_debug_fast.log | head -5 | cut -c1-300; exit 0; fi
timeout 560 python -m pytest tests -m gpu -q --timeout=300 --durations=6 > $out/r04h_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/r04h_gpu_tests.log
tail -14 $out/r04h_gpu_tests.log | cut -c1-300
