#!/bin/bash
# One GPU-box visit: (optionally) the parity suite and smoke, the bench line, rocprofv3 kernel stats of the same bench command,
# the side benches, the streaming probe.  Every process runs on a short leash (a wedged runtime must not eat the GPU budget).
# usage (repo root on the GPU box): bash tools/gpu_round.sh <tag> [tests]
tag=${1:-run}
with_tests=${2:-}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ "$with_tests" = "tests" ]; then
  timeout 600 python -m pytest tests -m gpu -q -x --timeout=300 --durations=8 > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
  tail -5 $out/${tag}_tests.log
  timeout 200 python __graft_entry__.py --smoke > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> $out/${tag}_smoke.log
  tail -3 $out/${tag}_smoke.log
fi
# the driver's command (default K / W), with the CPU baseline legs
timeout 300 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
cat $out/${tag}_bench.json
tail -3 $out/${tag}_bench.err
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-legs > $out/${tag}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/${tag}_bench_kernel_stats.csv \;
head -12 $out/${tag}_bench_kernel_stats.csv | cut -c1-200
# the streaming probe's kernels under the same profiler (exact dispatch durations: what a kernel of this shape can reach)
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o ${tag}p --output-format csv -- $GRAFT_REPO_ROOT/tools/probe/stream_probe > $out/${tag}_stream_probe_events.json 2>/dev/null
find /tmp/prof2 -name "${tag}p_kernel_stats.csv" -exec cp {} $out/${tag}_stream_probe_kernel_stats.csv \;
cat $out/${tag}_stream_probe_kernel_stats.csv | cut -c1-160
# side measurements (not the headline): frames handed over as host buffers (PCIe inclusive), closed-loop mode, configs[2]
cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --no-cpu-baseline --no-side-legs --host-frames > $out/${tag}_hostframes_bench.json 2>/dev/null
cut -c1-220 $out/${tag}_hostframes_bench.json
timeout 200 python bench.py --no-cpu-baseline --no-side-legs --close-loops > $out/${tag}_closeloops_bench.json 2>/dev/null
cut -c1-220 $out/${tag}_closeloops_bench.json
timeout 300 python bench.py --no-cpu-baseline --no-side-legs --width 1280 --height 960 --steps 100 --warmup 10 > $out/${tag}_1280x960_bench.json 2>/dev/null
cat $out/${tag}_1280x960_bench.json
