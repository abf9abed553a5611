#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench line, rocprofv3 kernel stats of the same bench command.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag>
tag=${1:-run}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
tail -5 $out/${tag}_tests.log
timeout 300 python __graft_entry__.py --smoke > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> $out/${tag}_smoke.log
tail -3 $out/${tag}_smoke.log
timeout 600 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
cat $out/${tag}_bench.json
tail -3 $out/${tag}_bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/${tag}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/ \;
head -12 $out/${tag}_kernel_stats.csv | cut -c1-200
# side measurements (not the headline): frames handed over as host buffers (PCIe inclusive), and closeLoops
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --no-cpu-baseline --host-frames --steps 200 --warmup 20 > $out/${tag}_hostframes_bench.json 2>/dev/null
cut -c1-220 $out/${tag}_hostframes_bench.json
timeout 300 python bench.py --no-cpu-baseline --close-loops --steps 200 --warmup 20 > $out/${tag}_closeloops_bench.json 2>/dev/null
cut -c1-220 $out/${tag}_closeloops_bench.json
# configs[2]: 1280x960, ~1 M surfels
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --width 1280 --height 960 --steps 100 --warmup 10 > $out/${tag}_1280x960_bench.json 2>/dev/null
cat $out/${tag}_1280x960_bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof3 -o ${tag}c3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --width 1280 --height 960 --steps 50 --warmup 5 > /dev/null 2>&1
find /tmp/prof3 -name "${tag}c3_kernel_stats.csv" -exec cp {} $out/${tag}_1280x960_kernel_stats.csv \;
