#!/bin/bash
tag=${1:-r03k}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 170 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc=$?"
python -c "
import json; d = json.load(open('$out/${tag}_bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], {k: (v.get('value') if isinstance(v, dict) else v) for k, v in d['side_legs'].items()}); c = d['side_legs']['config2_1280x960_1M']; print(c['roofline']['traffic'], c['roofline_index_splat']['traffic'], c['roofline']['frac'], c['roofline_index_splat']['frac'])"
tail -3 $out/${tag}_bench.err
