#!/bin/bash
# round-2 visit F: sampling inside vs after the timed region, PMC traffic of the roofline kernels, the whole -m gpu suite
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=r02f
for rep in 1 2; do
  for flag in "" "--probe-inside"; do
    timeout 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --frames-cache /tmp/efframes $flag 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('[$flag]', d['value'], 'fps', r['avg_us'], 'us accum L0 (frac', r['frac'], ') splat', d['roofline_index_splat']['avg_us'], 'us')" | tee -a $out/${tag}_probe_ab.log
  done
done
timeout 420 bash tools/pmc_traffic.sh ${tag}_pmc > $out/${tag}_pmc.log 2>&1; tail -12 $out/${tag}_pmc.log | cut -c1-200
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -x --timeout=300 --durations=8 > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
tail -16 $out/${tag}_tests.log | cut -c1-200
