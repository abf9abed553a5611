#!/bin/bash
# round 3: rocprofv3 kernel stats of the CLOSED-LOOP bench leg (closeLoops = true at the reference's default window), the breakdown behind DESIGN 5.4's
# "why not 1000 frames/s in the reference's default mode"
tag=${1:-r03m}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --close-loops --no-cpu-baseline --no-side-legs > $out/${tag}_closeloops_stdout.log 2>&1
echo "rc=$?"
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/${tag}_closeloops_kernel_stats.csv \;
tail -1 $out/${tag}_closeloops_stdout.log | cut -c1-300
head -8 $out/${tag}_closeloops_kernel_stats.csv | cut -c1-70,200-330
