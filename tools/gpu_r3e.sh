#!/bin/bash
# round 3, fifth GPU visit: the whole -m gpu suite at HEAD (persistent tracker with barriers + half gather, fern coding on the device,
# tile-binned surface splat, closed-loop steady-state test), A/B of the tiled splat against the all-global one, the default bench line
# and its rocprofv3 kernel stats.
tag=${1:-r03e}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q --timeout=400 --durations=8 > $out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> $out/${tag}_tests.log
tail -16 $out/${tag}_tests.log
timeout 300 bash tools/gpu_ab.sh ${tag} - splatglobal
timeout 300 python bench.py --frames-cache /tmp/efframes > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
python -c "
import json; d = json.load(open('$out/${tag}_bench.json')); print(d['value'], d['roofline']['frac'], {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d['side_legs'].items()})"
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${tag} --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-legs --frames-cache /tmp/efframes > $out/${tag}_prof_stdout.log 2>&1
find /tmp/prof -name "${tag}_kernel_stats.csv" -exec cp {} $out/${tag}_bench_kernel_stats.csv \;
head -8 $out/${tag}_bench_kernel_stats.csv | cut -c1-60,200-330
