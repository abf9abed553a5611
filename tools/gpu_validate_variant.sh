#!/bin/bash
# Validate a compile-time variant of libefusion_hip on ONE GPU box before it becomes the default: the full -m gpu suite on the variant
# library (the Python harness loads what EF_HIP_LIB names), then the A/B bench against the default build and kernel stats of the variant.
# Build the variant first, where hipcc is:   python -m elasticfusion_amd.build --variant quadmaps -DEF_MODEL_MAPS_QUAD
# usage (repo root on the GPU box):          bash tools/gpu_validate_variant.sh <tag> <variant>
tag=${1:-run}; v=${2:?variant suffix}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
lib=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_$v.so
[ -f $lib ] || { echo "missing $lib"; exit 2; }
EF_HIP_LIB=$lib timeout 400 python -m pytest tests -m gpu -q --timeout=200 > $out/${tag}_${v}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_${v}_tests.log
tail -4 $out/${tag}_${v}_tests.log
timeout 200 bash tools/gpu_ab.sh ${tag}_${v} - $v
cd /tmp
EF_HIP_LIB=$lib timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag}_$v --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --frames-cache /tmp/efframes > $out/${tag}_${v}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_${v}_kernel_stats.csv" -exec cp {} $out/${tag}_${v}_bench_kernel_stats.csv \;
head -12 $out/${tag}_${v}_bench_kernel_stats.csv | cut -c1-160
