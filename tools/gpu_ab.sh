#!/bin/bash
# A/B on ONE box (box-to-box variance is larger than most effects): the same build under different environment knobs.
# usage: bash tools/gpu_ab.sh <tag> "<env assignments | ->" ...     e.g.  bash tools/gpu_ab.sh ab - "EF_OVERLAP=1" "EF_OVERLAP=2 EF_OVERLAP_PRIO=1"
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do
  for cfg in "$@"; do
    [ "$cfg" = "-" ] && cfg=""
    env $cfg timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], 'fps', d['roofline']['avg_us'], 'us accum')" | tee -a $out/${tag}_ab.log
  done
done
