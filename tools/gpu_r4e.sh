#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-r04e}
timeout 100 python tools/debug_fast.py 4 persistent 2>&1 | grep -c SAME
timeout 120 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so 140 > $out/${tag}_fast_clocks.jsonl 2>$out/${tag}_fast_clocks.err; cat $out/${tag}_fast_clocks.jsonl
B="python bench.py --no-cpu-baseline --no-side-legs --steps 100 --warmup 10 --frames-cache /tmp/efframes"
run() { t=$1; shift; "$@" 2>$out/${tag}_$t.err | tee $out/${tag}_$t.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}; t = d.get('roofline_tracker') or {}
print('[$t]', d['value'], 'fps | accum L0', r.get('avg_us'), 'us | tracker', t.get('avg_us'), 'us | frame', (d.get('frame_time_ms') or {}).get('median'), '| calib', (d.get('box_calibration') or {}).get('empty_kernel_us'))" | tee -a $out/${tag}_ab.log; }
run fast_persistent timeout 150 $B
EF_HIP_LIB=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_reforder.so run r3_product timeout 150 $B
