#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 120 python tools/debug_fast.py 6 > $out/r04l_debug_fast.log 2>&1; grep "frames differ" $out/r04l_debug_fast.log
timeout 100 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > $out/r04l_fast_clocks.jsonl 2>$out/r04l_fast_clocks.err; cat $out/r04l_fast_clocks.jsonl
timeout 200 python -m pytest tests/test_gpu_fallback.py tests/test_gpu_frame.py tests/test_gpu_one_frame.py -m gpu -q --timeout=150 -k "fallback or two_contexts or persistent_and or checkpoint" > $out/r04l_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/r04l_gpu_tests.log
tail -6 $out/r04l_gpu_tests.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-side-legs --steps 200 --warmup 20 --frames-cache /tmp/efframes"
run() { t=$1; shift; "$@" 2>$out/r04l_$t.err | tee $out/r04l_$t.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}; t = d.get('roofline_tracker') or {}
print('[$t]', d['value'], 'fps | accum L0', r.get('avg_us'), 'us | tracker', t.get('avg_us'), 'us | frame', (d.get('frame_time_ms') or {}).get('median'))" | tee -a $out/r04l_ab.log; }
run fast_persistent timeout 150 $B
EF_HIP_LIB=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_reforder.so run r3_product timeout 150 $B
run fast_persistent_2 timeout 150 $B
