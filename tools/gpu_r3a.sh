#!/bin/bash
# round 3, first GPU visit: the three finished variants.  Full -m gpu suite on the combined variant, then an
# A/B of default / quadmaps / readlane / both on this ONE box, then the kernel stats of default and both.
tag=${1:-r03a}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
lib=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_both.so
EF_HIP_LIB=$lib timeout 420 python -m pytest tests -m gpu -q --timeout=200 > $out/${tag}_both_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_both_tests.log
tail -4 $out/${tag}_both_tests.log
timeout 420 bash tools/gpu_ab.sh ${tag} - quadmaps readlane both
cd /tmp
for v in default both; do
  lib=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip.so
  [ $v = both ] && lib=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_both.so
  EF_HIP_LIB=$lib timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o ${tag}_$v --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-legs --frames-cache /tmp/efframes > $out/${tag}_${v}_prof_stdout.log 2>&1
  find /tmp/prof_$v -name "${tag}_${v}_kernel_stats.csv" -exec cp {} $out/${tag}_${v}_bench_kernel_stats.csv \;
  head -14 $out/${tag}_${v}_bench_kernel_stats.csv | cut -c1-150
done
