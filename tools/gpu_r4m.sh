#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-side-legs --steps 200 --warmup 20 --frames-cache /tmp/efframes"
run() { t=$1; shift; "$@" 2>$out/r04m_$t.err | tee $out/r04m_$t.json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}; t = d.get('roofline_tracker') or {}
print('[$t]', d['value'], 'fps | tracker', t.get('avg_us'), 'us | frame', (d.get('frame_time_ms') or {}).get('median'), '| err', d['config'].get('pose_err_vs_generating_traj_m'), d['config'].get('surfels_end'))" | tee -a $out/r04m_ab.log; }
run default timeout 150 $B
run overlap_4 timeout 150 $B --input-overlap 4
run overlap_2 timeout 150 $B --input-overlap 2
run overlap_8 timeout 150 $B --input-overlap 8
run overlap_1 timeout 150 $B --input-overlap 1
run default_2 timeout 150 $B
