#!/bin/bash
# round 3, second GPU visit: the persistent small-level tracker (k_track_small).  Parity first (frame tier + closed loop + the new
# one-frame / checkpoint tests), then persistent vs per-step on this ONE box, then kernel stats.
tag=${1:-r03b}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_frame.py tests/test_gpu_one_frame.py tests/test_gpu_global.py tests/test_gpu_loop.py tests/test_gpu_reloc.py tests/test_gpu_replay.py tests/test_gpu_vs_reference.py -m gpu -q -x --timeout=200 --durations=6 > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
tail -25 $out/${tag}_tests.log
for rep in 1 2; do
  for v in persistent per-step; do
    flag=""; [ $v = per-step ] && flag="--per-step-tracker"
    timeout 200 python bench.py --no-cpu-baseline --no-side-legs --steps 200 --warmup 20 --frames-cache /tmp/efframes $flag 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('[$v]', d['value'], 'fps', r['avg_us'], 'us accum L0 (frac', r['frac'], ') splat', d['roofline_index_splat']['avg_us'], 'us')" | tee -a $out/${tag}_ab.log
  done
done
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --frames-cache /tmp/efframes > $out/${tag}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/${tag}_bench_kernel_stats.csv \;
head -12 $out/${tag}_bench_kernel_stats.csv | cut -c1-60,200-300
