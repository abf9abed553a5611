#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-kernel PMC summaries tools/pmc_traffic.sh leaves in gpurun_out/.

    python tools/pmc_json.py <tag> [<tag2>]   # reads gpurun_out/<tag>_{FETCH,WRITE}_SIZE_per_kernel.csv and *_calibration.txt
                                              # tag: a run of the launch-per-step script (k_se3_accum_fast is a launch of its own there),
                                              # tag2: a default run (the persistent tracker launch k_track_fast), same calibration

HBM-side bytes per launch = FETCH_SIZE x (bytes per counted KB, from the calibration kernel) + WRITE_SIZE x (same).
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out")


def per_kernel(ctr, tag=tag):
    out = {}
    with open(os.path.join(src, f"{tag}_{ctr}_per_kernel.csv")) as f:
        for r in csv.DictReader(f):
            out[r["kernel"]] = (int(r["dispatches"]), float(r[f"{ctr}_per_dispatch"]))
    return out


def calibration(ctr):
    txt = open(os.path.join(src, f"{tag}_{ctr}_calibration.txt")).read()
    return float(re.search(r"bytes per counted KB = ([0-9.]+)", txt).group(1))


fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
cf, cw = calibration("FETCH_SIZE"), calibration("WRITE_SIZE")


def pick(prefix, level0_only=False):
    cands = [k for k in fetch if prefix in k]
    if not cands:
        return None
    # the level-0 instance of a templated kernel is the one with the most bytes per dispatch
    k = max(cands, key=lambda k: fetch[k][1])
    return k, fetch[k][1], write.get(k, (0, 0.0))[1]


k, f, w = pick("k_se3_accum")
doc = {"kernel": re.sub(r"^.*?(k_se3_accum\w*<[^>]*>).*$", r"\1", k), "FETCH_SIZE_KB_per_dispatch": round(f, 3),
       "WRITE_SIZE_KB_per_dispatch": round(w, 3), "fetch_bytes_per_counted_KB": cf, "write_bytes_per_counted_KB": cw,
       "traffic_bytes_per_launch": int(round(f * cf + w * cw)), "source": f"tools/pmc_traffic.sh {tag}", "also": {}}
for name in ("k_index_splat", "k_index_resolve"):
    p = pick(name)
    if p:
        _, f2, w2 = p
        doc["also"][name] = {"FETCH_SIZE_KB_per_dispatch": round(f2, 3), "WRITE_SIZE_KB_per_dispatch": round(w2, 3),
                             "traffic_bytes_per_launch": int(round(f2 * cf + w2 * cw))}
if len(sys.argv) > 2:   # the persistent tracker launch, from a default run
    tag2 = sys.argv[2]
    f2, w2 = per_kernel("FETCH_SIZE", tag2), per_kernel("WRITE_SIZE", tag2)
    ks = [k for k in f2 if "k_track_fast" in k]
    if ks:
        k = max(ks, key=lambda k: f2[k][1])
        doc["also"]["k_track_fast"] = {"FETCH_SIZE_KB_per_dispatch": round(f2[k][1], 3), "WRITE_SIZE_KB_per_dispatch": round(w2.get(k, (0, 0.0))[1], 3),
                                       "traffic_bytes_per_launch": int(round(f2[k][1] * cf + w2.get(k, (0, 0.0))[1] * cw)), "source": f"tools/pmc_traffic.sh {tag2}"}
doc["how"] = (f"tools/pmc_traffic.sh {tag}: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
              "`bench.py --steps 12 --warmup 3`; calibrated on k_transform_maps (75497472 B read + written, 4 B/lane planar): "
              "FETCH_SIZE counts half the bytes (gfx950), WRITE_SIZE is exact")
with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as fo:
    json.dump(doc, fo, indent=1)
print(json.dumps(doc, indent=1))
