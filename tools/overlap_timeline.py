#!/usr/bin/env python
"""Per-frame kernel timeline from a `rocprofv3 --kernel-trace` CSV (start / end timestamps per dispatch, not --stats): which kernels ran on which
hardware queue, when, and which of them overlapped ("no timeline says whether anything actually ran concurrently").

    rocprofv3 --kernel-trace -d /tmp/tl -o tl --output-format csv -- python tools/ab_bench.py --steps 40 --reps 1 d+ov1
    python tools/overlap_timeline.py /tmp/tl/.../tl_kernel_trace.csv [--frames 3] > profiles/r06_overlap_timeline.txt

A frame starts at its k_preprocess dispatch (input stage) — frames are cut at consecutive k_track_ref dispatches on the main queue; the LAST
`--frames` complete frames are printed: offset from the frame's first dispatch, duration, queue, short kernel name, and '*' where a dispatch
overlaps a dispatch of ANOTHER queue in time.  The summary gives, per frame: wall time between consecutive k_track_ref starts, the sum of
kernel durations, and the time during which two queues were busy at once."""
import csv
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def main():
    path = [a for a in sys.argv[1:] if not a.startswith("--")][0]
    nframes = 3
    if "--frames" in sys.argv:
        nframes = int(sys.argv[sys.argv.index("--frames") + 1])
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append(dict(name=short(r["Kernel_Name"]), q=r.get("Queue_Id", "?"), t0=int(r["Start_Timestamp"]), t1=int(r["End_Timestamp"])))
    rows.sort(key=lambda r: r["t0"])
    track = [i for i, r in enumerate(rows) if r["name"] == "k_track_ref"]
    if len(track) < nframes + 2:
        print("not enough k_track_ref dispatches", len(track))
        return
    queues = sorted(set(r["q"] for r in rows))
    print(f"# {path}: {len(rows)} dispatches, queues {queues}")
    summary = []
    for fi in range(len(track) - nframes - 1, len(track) - 1):
        a, b = track[fi], track[fi + 1]
        # the frame = everything from this tracker launch up to (not including) the next one
        fr = rows[a:b]
        base = fr[0]["t0"]
        wall = rows[b]["t0"] - base
        busy = sum(r["t1"] - r["t0"] for r in fr)
        # time with >= 2 queues busy
        ev = []
        for r in fr:
            ev.append((r["t0"], 1, r["q"]))
            ev.append((r["t1"], -1, r["q"]))
        ev.sort()
        act = {}
        both = 0
        last = None
        for t, d, q in ev:
            if last is not None and sum(1 for v in act.values() if v > 0) >= 2:
                both += t - last
            act[q] = act.get(q, 0) + d
            last = t
        idle = 0
        cover = []
        for r in sorted(fr, key=lambda r: r["t0"]):
            if cover and r["t0"] <= cover[-1][1]:
                cover[-1][1] = max(cover[-1][1], r["t1"])
            else:
                cover.append([r["t0"], r["t1"]])
        covered = sum(c[1] - c[0] for c in cover)
        summary.append((wall, busy, both, wall - covered))
        print(f"\n## frame {fi}: wall {wall / 1e3:.1f} us (tracker start to tracker start), sum of kernel durations {busy / 1e3:.1f} us, two queues busy at once "
              f"{both / 1e3:.1f} us, no kernel running {(wall - covered) / 1e3:.1f} us")
        for r in fr:
            ov = any(o is not r and o["q"] != r["q"] and o["t0"] < r["t1"] and r["t0"] < o["t1"] for o in fr)
            print(f"  +{(r['t0'] - base) / 1e3:8.1f} us  {(r['t1'] - r['t0']) / 1e3:7.1f} us  q{r['q']:>3} {'*' if ov else ' '} {r['name']}")
    w = sum(s[0] for s in summary) / len(summary)
    print(f"\n# mean over {len(summary)} frames: wall {w / 1e3:.1f} us, kernels {sum(s[1] for s in summary) / len(summary) / 1e3:.1f} us, "
          f"two queues busy {sum(s[2] for s in summary) / len(summary) / 1e3:.1f} us, idle {sum(s[3] for s in summary) / len(summary) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
