#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
L=elasticfusion_amd/libefusion_hip
timeout 200 python tools/fast_clocks.py ${L}_clocks.so ${L}_clocksna.so ${L}_clocks.so ${L}_clocksna.so 140 > $out/r04k_fast_clocks.jsonl 2>$out/r04k_fast_clocks.err; cut -c1-900 $out/r04k_fast_clocks.jsonl
