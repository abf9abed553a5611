#!/bin/bash
# round 3: the local closure's second tracker initialised in five launches (eft::init_model_pair) — closed-loop bench under rocprofv3, then the
# closed-loop parity tests (194 frames at default thresholds with a loop closing on its own; the local-closure tests)
tag=${1:-r03n}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout 60 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --close-loops --no-cpu-baseline --no-side-legs > $out/${tag}_closeloops_stdout.log 2>&1
echo "prof rc=$?"
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/${tag}_closeloops_kernel_stats.csv \;
cd $GRAFT_REPO_ROOT
(timeout 30 python bench.py --close-loops --no-cpu-baseline --no-side-legs | cut -c1-160) > $out/${tag}_closeloops_bench.log 2>&1; cat $out/${tag}_closeloops_bench.log
timeout 110 python -m pytest tests/test_gpu_closed_steady.py tests/test_gpu_loop.py -x -q --timeout=100 > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?"
tail -4 $out/${tag}_tests.log
