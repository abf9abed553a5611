#!/bin/bash
# round 3, third GPU visit: fern coding on the device + deferred end-of-frame record (closure tests), the one-frame parity test, the
# half-gather persistent kernel (parity, A/B against the full gather, per-phase clocks of both), and the default bench line with its
# new side legs.
tag=${1:-r03c}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_frame.py tests/test_gpu_global.py tests/test_gpu_loop.py tests/test_gpu_reloc.py tests/test_gpu_replay.py tests/test_gpu_one_frame.py -m gpu -q --timeout=200 -k "persistent or tracking_and_fusion or configurations or small_and_odd or global or loop or reloc or replay or one_frame or checkpoint or degenerate" > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
tail -12 $out/${tag}_tests.log
timeout 200 python tools/small_clocks.py elasticfusion_amd/libefusion_hip_clocks.so elasticfusion_amd/libefusion_hip_ptfullclocks.so > $out/${tag}_small_clocks.jsonl 2>$out/${tag}_small_clocks.err
cat $out/${tag}_small_clocks.jsonl; tail -2 $out/${tag}_small_clocks.err
timeout 300 bash tools/gpu_ab.sh ${tag} - ptfull
timeout 300 python bench.py --frames-cache /tmp/efframes > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
cat $out/${tag}_bench.json; tail -3 $out/${tag}_bench.err
