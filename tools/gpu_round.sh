#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench line, rocprofv3 kernel stats of the same bench command.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag>
tag=${1:-run}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
tail -5 $out/${tag}_tests.log
timeout 300 python __graft_entry__.py --smoke > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> $out/${tag}_smoke.log
tail -3 $out/${tag}_smoke.log
timeout 600 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
cat $out/${tag}_bench.json
tail -3 $out/${tag}_bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/${tag}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/ \;
find /tmp/prof -name "${tag}_domain_stats.csv" -exec cp {} $out/ \;
head -12 $out/${tag}_kernel_stats.csv | cut -c1-200
