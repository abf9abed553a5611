#!/bin/bash
# round 3, fourth GPU visit: the persistent kernel's SE(3) loop on tagged granules (no barrier): parity, clocks, A/B against the barrier
# version (libefusion_hip_ptfull.so, built from the previous commit's kernel); the closure tests again (deferred end-of-frame record).
tag=${1:-r03d}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_frame.py tests/test_gpu_global.py tests/test_gpu_loop.py tests/test_gpu_reloc.py tests/test_gpu_replay.py tests/test_gpu_one_frame.py -m gpu -q --timeout=200 -k "persistent or tracking_and_fusion or configurations or small_and_odd or global or loop or reloc or replay or one_frame or checkpoint or degenerate or two_contexts or another_thread" > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
tail -12 $out/${tag}_tests.log
timeout 200 python tools/small_clocks.py elasticfusion_amd/libefusion_hip_clocks.so > $out/${tag}_small_clocks.jsonl 2>$out/${tag}_small_clocks.err
cat $out/${tag}_small_clocks.jsonl; tail -2 $out/${tag}_small_clocks.err
timeout 300 bash tools/gpu_ab.sh ${tag} - ptfull
for f in "--close-loops" "--width 1280 --height 960 --preseed 1048576 --steps 60 --warmup 10"; do
  timeout 200 python bench.py --no-cpu-baseline --no-side-legs $f 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$f]', d['value'], 'fps', d['roofline']['avg_us'], d['roofline']['frac'], d['roofline_index_splat']['avg_us'], d['roofline_index_splat']['frac'], d['config']['surfels_end'])" | tee -a $out/${tag}_side.log
done
