#!/bin/bash
# A/B on ONE GPU box (box-to-box variance is larger than most effects): the same bench on several builds of libefusion_hip
# (EF_HIP_LIB selects the library the Python harness loads; the product library itself reads no environment variable).
# usage: bash tools/gpu_ab.sh <tag> <lib-suffix | -> ...     e.g.  bash tools/gpu_ab.sh ab - r02a quadloads valu
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2; do   # two passes: the first also warms the frame cache
  for v in "$@"; do
    lib=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip.so
    [ "$v" != "-" ] && lib=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_$v.so
    [ -f $lib ] || { echo "[$v] missing"; continue; }
    EF_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-side-legs --steps 200 --warmup 20 --frames-cache /tmp/efframes 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('[$v]', d['value'], 'fps', r['avg_us'], 'us accum L0 (frac', r['frac'], ') splat', d['roofline_index_splat']['avg_us'], 'us')" | tee -a $out/${tag}_ab.log
  done
done
