#!/bin/bash
# round 4 visit c: same-box A/B (persistent fast / per-step fast / round-3 product), then the whole -m gpu suite
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-side-legs --steps 200 --warmup 20 --frames-cache /tmp/efframes"
run() { tag=$1; shift; "$@" 2>$out/r04c_$tag.err | tee $out/r04c_$tag.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}; t = d.get('roofline_tracker') or {}
print('[$tag]', d['value'], 'fps | accum L0', r.get('avg_us'), 'us frac', r.get('frac'), '| tracker', t.get('avg_us'), 'us | frame', d.get('frame_time_ms'), '| calib', d.get('box_calibration'))" | tee -a $out/r04c_ab.log; }
run fast_persistent timeout 150 $B
run fast_per_step timeout 150 $B --per-step-tracker
EF_HIP_LIB=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_reforder.so run r3_product timeout 150 $B
run fast_persistent_2 timeout 150 $B
run fast_close_loops timeout 150 $B --close-loops
timeout 500 python -m pytest tests -m gpu -q --timeout=200 --durations=5 > $out/r04c_gpu_tests.log 2>&1; echo "pytest rc=$?" >> $out/r04c_gpu_tests.log
tail -30 $out/r04c_gpu_tests.log
