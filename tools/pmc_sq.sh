#!/bin/bash
# SQ counters of a frame's kernels (MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ slots a pass): where the wavefronts' cycles go —
# WAIT_ANY (parked: s_waitcnt / barrier / s_sleep), WAIT_INST_ANY (issue stall), ACTIVE_INST_VALU / _LDS, and the LDS bank conflicts.
# usage (GPU box, repo root): bash tools/pmc_sq.sh <tag> [bench.py arguments]
tag=${1:-sq}; shift
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT \
  -d /tmp/pmc_sq -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-side-legs "$@" > $out/${tag}_sq_stdout.log 2>&1
f=$(find /tmp/pmc_sq -name "p_counter_collection.csv" | head -1)
python - "$f" > $out/${tag}_sq_per_kernel.csv <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z0-9_]+(?:<[^>]*>)?)", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
cols = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"]
print("kernel,dispatches," + ",".join(c + "_per_dispatch" for c in cols) + ",wait_any_frac,wait_inst_frac,active_valu_frac,active_lds_frac")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = v.get("SQ_WAVE_CYCLES", 0) or 1.0
    print('"%s",%d,' % (k, n[k]) + ",".join("%.0f" % (v.get(c, 0) / n[k]) for c in cols) + ",%.3f,%.3f,%.3f,%.3f" % (v.get("SQ_WAIT_ANY", 0) / w, v.get("SQ_WAIT_INST_ANY", 0) / w, v.get("SQ_ACTIVE_INST_VALU", 0) / w, v.get("SQ_ACTIVE_INST_LDS", 0) / w))
PY
head -20 $out/${tag}_sq_per_kernel.csv | cut -c1-260
