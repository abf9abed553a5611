#!/bin/bash
# A/B on ONE box: bench with and without the input-stage overlap (+ kernel stats of both), same build.
# usage: bash tools/gpu_ab.sh <tag>
tag=${1:-ab}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for mode in on off on off; do
  if [ $mode = off ]; then export EF_NO_OVERLAP=1; else unset EF_NO_OVERLAP; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap=$mode', d['value'], 'fps', d['roofline']['avg_us'], 'us accum')" | tee -a $out/${tag}_ab.log
done
cd /tmp
for mode in on off; do
  if [ $mode = off ]; then export EF_NO_OVERLAP=1; else unset EF_NO_OVERLAP; fi
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o ${tag}_$mode --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
  find /tmp/prof_$mode -name "${tag}_${mode}_kernel_stats.csv" -exec cp {} $out/ \;
done
