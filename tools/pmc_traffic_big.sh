#!/bin/bash
# HBM-side traffic of the level-0 normal equations and the IndexMap splat on BASELINE configs[2] (1280x960, map pre-seeded with ~1 M
# surfels): FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md); the calibration constants are those of
# tools/pmc_traffic.sh (FETCH_SIZE counts half the bytes on gfx950: 2047.3 B per counted KB; WRITE_SIZE 1024).
# usage (GPU box, repo root): bash tools/pmc_traffic_big.sh <tag>
tag=${1:-pmcbig}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$ctr
  timeout 150 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmcb_$ctr -o p --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py --width 1280 --height 960 --preseed 1048576 --preroll 10 --steps 8 --warmup 2 --no-cpu-baseline --no-side-legs --frames-cache /tmp/efframes_pmcb > $out/${tag}_${ctr}_stdout.log 2>&1
  f=$(find /tmp/pmcb_$ctr -name "p_counter_collection.csv" | head -1)
  python - "$f" "$ctr" > $out/${tag}_${ctr}_per_kernel.csv <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != ctr:
        continue
    k = r["Kernel_Name"]
    acc[k][0] += 1
    acc[k][1] += float(r["Counter_Value"])
print("kernel,dispatches,%s_total,%s_per_dispatch" % (ctr, ctr))
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%.1f,%.3f' % (k.replace('"', "'"), n, v, v / n))
PY
  head -8 $out/${tag}_${ctr}_per_kernel.csv | cut -c1-60,150-400
done
