#!/bin/bash
tag=${1:-r03h}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 240 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
python -c "
import json; d = json.load(open('$out/${tag}_bench.json')); print(d['value'], d['roofline']['frac'], {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d['side_legs'].items()}); print(d['side_legs']['config2_1280x960_1M'].get('ms_per_step'))"
