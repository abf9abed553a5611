#!/bin/bash
# HBM-side traffic of the dominant kernels from the L2's memory-side request counters, as MI355X_MICROARCH.md §HBM
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not fit one pass), only --kernel-trace beside them.
# usage (GPU box, repo root): bash tools/pmc_traffic.sh <tag> [extra bench.py arguments, e.g. --per-step-tracker: the launch-per-step script,
# in which the level-0 normal equations are a launch of their own (k_se3_accum_fast); without it the tracker is k_track_fast]
tag=${1:-pmc}
shift
extra="$@"
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout 200 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$ctr -o p --output-format csv -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-side-legs --frames-cache /tmp/efframes_pmc $extra > $out/${tag}_${ctr}_stdout.log 2>&1
  f=$(find /tmp/pmc_$ctr -name "p_counter_collection.csv" | head -1)
  echo "$ctr -> $f"
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
  head -6 $out/${tag}_${ctr}_per_kernel.csv | cut -c1-220
  # calibration on a known byte count in the same access pattern (4 B / lane planar maps); once per visit is enough (PMC_SKIP_CAL=1)
  [ -n "$PMC_SKIP_CAL" ] && continue
  rm -rf /tmp/cal_$ctr
  timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/cal_$ctr -o c --output-format csv -- python $GRAFT_REPO_ROOT/tools/pmc_calibrate.py > $out/${tag}_${ctr}_cal_stdout.log 2>&1
  f=$(find /tmp/cal_$ctr -name "c_counter_collection.csv" | head -1)
  python - "$f" "$ctr" <<'PY' | tee $out/${tag}_${ctr}_calibration.txt
import csv, sys
f, ctr = sys.argv[1], sys.argv[2]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r.get("Counter_Name") == ctr and "k_transform_maps" in r["Kernel_Name"]]
print("calibration %s: k_transform_maps dispatches %d, mean counter value %.1f (KB) for 75497472 B => bytes per counted KB = %.1f"
      % (ctr, len(v), sum(v) / len(v), 75497472.0 / (sum(v) / len(v))))
PY
done
