#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 150 python tools/fast_clocks.py elasticfusion_amd/libefusion_hip_clocks.so elasticfusion_amd/libefusion_hip_clockslate.so 140 > $out/r04g_fast_clocks.jsonl 2>$out/r04g_fast_clocks.err; cat $out/r04g_fast_clocks.jsonl
