#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the per-kernel PMC summaries tools/pmc_traffic.sh leaves in gpurun_out/.

    python tools/pmc_json.py <tag> [<tag_per_step>] [--out profiles/pmc_traffic.json]

<tag>: a run of the default script (gpurun_out/<tag>_{FETCH,WRITE}_SIZE_per_kernel.csv + *_calibration.txt): the persistent tracker launch
(k_track_ref; k_track_fast in the fast build), the IndexMap splat and every other kernel of a frame — and, from the frames bench.py replays
with the launch-per-step script behind its timed region, the level-0 normal equations as a launch of their own (k_se3_accum<5, ...>);
<tag_per_step> (optional): take k_se3_accum from a run with --per-step-tracker instead, same calibration.

HBM-side bytes per launch = FETCH_SIZE x (bytes per counted KB, from the calibration kernel) + WRITE_SIZE x (same); bench.py reads
kernels[<name>].traffic_bytes_per_launch by kernel-name prefix (pmc_traffic_of).
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
if "--out" in sys.argv:
    out_path = sys.argv[sys.argv.index("--out") + 1]
    args = [a for a in args if a != out_path]
tag = args[0]
src = os.path.join(ROOT, "gpurun_out")


def per_kernel(ctr, tag):
    out = {}
    with open(os.path.join(src, f"{tag}_{ctr}_per_kernel.csv")) as f:
        for r in csv.DictReader(f):
            out[r["kernel"]] = (int(r["dispatches"]), float(r[f"{ctr}_per_dispatch"]))
    return out


def calibration(ctr):
    txt = open(os.path.join(src, f"{tag}_{ctr}_calibration.txt")).read()
    return float(re.search(r"bytes per counted KB = ([0-9.]+)", txt).group(1))


def short(k):
    m = re.search(r"(k_[a-z0-9_]+(?:<[^>]*>)?)", k)
    return m.group(1) if m else k


cf, cw = calibration("FETCH_SIZE"), calibration("WRITE_SIZE")
doc = {"source": f"tools/pmc_traffic.sh {tag}", "fetch_bytes_per_counted_KB": cf, "write_bytes_per_counted_KB": cw, "kernels": {}}
for t, wanted in ((tag, None), (args[1] if len(args) > 1 else None, ("k_se3_accum",))):
    if t is None:
        continue
    fetch, write = per_kernel("FETCH_SIZE", t), per_kernel("WRITE_SIZE", t)
    for k, (n, f) in fetch.items():
        name = short(k)
        if not name.startswith("k_") or name.startswith("k_calib") or (wanted and not name.startswith(wanted)) or (not wanted and len(args) > 1 and name.startswith("k_se3_accum")):
            continue
        w = write.get(k, (0, 0.0))[1]
        rec = {"dispatches": n, "FETCH_SIZE_KB_per_dispatch": round(f, 3), "WRITE_SIZE_KB_per_dispatch": round(w, 3),
               "traffic_bytes_per_launch": int(round(f * cf + w * cw)), "source": f"tools/pmc_traffic.sh {t}"}
        if name.startswith("k_se3_accum<"):
            # bench.py runs the launch-per-step script on the frames behind its timed region.  Round 6's instances: <6, ...> = a level of ONE round of
            # six steps (640x480: level 0; 1280x960: level 1), <4, ...> = the multi-round path (1280x960: level 0), <2, ...> = the small levels;
            # the instance with the most bytes per dispatch is kept below: the level-0 launches at either size
            rec["launches_are"] = "level 0 (the instance with the most bytes per dispatch: <6, ...> at 640x480, <4, ...> at 1280x960)"
        if name in doc["kernels"] and doc["kernels"][name]["traffic_bytes_per_launch"] >= rec["traffic_bytes_per_launch"]:
            continue   # (templated kernels: keep the instance with the most bytes per dispatch)
        doc["kernels"][name] = rec
doc["how"] = (f"tools/pmc_traffic.sh {tag}: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
              "`bench.py --steps 12 --warmup 3`; per-kernel means over the run's dispatches; calibrated on k_transform_maps (75497472 B read + written, "
              "4 B/lane planar): FETCH_SIZE counts half the bytes (gfx950), WRITE_SIZE is exact")
with open(out_path, "w") as fo:
    json.dump(doc, fo, indent=1)
print(json.dumps({k: v["traffic_bytes_per_launch"] for k, v in doc["kernels"].items()}, indent=1))
