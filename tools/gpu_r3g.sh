#!/bin/bash
# round 3, last GPU visit at HEAD: smoke, a parity subset, the driver's bench command and the rocprofv3 kernel stats of the headline workload
tag=${1:-r03g}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 60 python __graft_entry__.py --smoke > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> $out/${tag}_smoke.log; tail -2 $out/${tag}_smoke.log
timeout 240 python -m pytest tests/test_gpu_frame.py tests/test_gpu_ops_map.py tests/test_gpu_ops_tracking.py tests/test_gpu_ops_linalg.py -m gpu -q --timeout=150 > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
tail -4 $out/${tag}_tests.log
timeout 200 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
python -c "
import json; d = json.load(open('$out/${tag}_bench.json')); print(d['value'], d['roofline']['frac'], {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d['side_legs'].items()})"
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-legs > $out/${tag}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/${tag}_bench_kernel_stats.csv \;
head -6 $out/${tag}_bench_kernel_stats.csv | cut -c1-60,200-330
