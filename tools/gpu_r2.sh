#!/bin/bash
# One GPU-box visit of round 2.  usage (repo root on the GPU box): bash tools/gpu_r2.sh <tag> [quick|full|bench]
# quick: the tracking-operator / frame parity tests on the product library, and on the test-only variants when they exist
# full : the whole -m gpu suite, smoke, bench line, rocprofv3 kernel stats of the same bench command
tag=${1:-r02}
mode=${2:-full}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
quick_tests="tests/test_gpu_ops_tracking.py tests/test_gpu_frame.py::test_tracking_and_fusion_sequence tests/test_gpu_frame.py::test_small_and_odd_resolutions_match_oracle tests/test_gpu_frame.py::test_tracker_configurations_match_oracle"
timeout 600 python -m pytest $quick_tests -m gpu -x -q > $out/${tag}_quick.log 2>&1; qrc=$?
echo "quick rc=$qrc"; tail -4 $out/${tag}_quick.log
if [ $qrc -ne 0 ]; then
  for v in valu swap; do
    lib=$GRAFT_REPO_ROOT/elasticfusion_amd/libefusion_hip_$v.so
    [ -f $lib ] || continue
    EF_HIP_LIB=$lib timeout 600 python -m pytest $quick_tests -m gpu -x -q > $out/${tag}_quick_$v.log 2>&1
    echo "variant $v rc=$?"; tail -4 $out/${tag}_quick_$v.log
  done
fi
[ "$mode" = "quick" ] && exit 0
if [ "$mode" = "ab" ]; then   # bash tools/gpu_r2.sh <tag> ab <variants...>
  shift 2
  bash tools/gpu_ab.sh $tag "$@"
  exit 0
fi
if [ "$mode" = "full" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
  tail -25 $out/${tag}_tests.log
  timeout 300 python __graft_entry__.py --smoke > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> $out/${tag}_smoke.log
  tail -3 $out/${tag}_smoke.log
fi
timeout 600 python bench.py --steps 200 --warmup 20 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
cat $out/${tag}_bench.json
tail -3 $out/${tag}_bench.err
timeout 300 python bench.py --no-cpu-baseline > $out/${tag}_bench_default.json 2>/dev/null   # what the driver runs (default K / W)
cut -c1-400 $out/${tag}_bench_default.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $out/${tag}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/ \;
head -30 $out/${tag}_kernel_stats.csv | cut -c1-160
