#!/usr/bin/env python
"""Per-kernel VGPR / scratch / LDS / occupancy table for one HIP source (gfx950): python tools/kernel_resources.py <file.hip> [filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", *sys.argv[3:], "-c", src,
       "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.rsplit(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("eft::", "").replace("efm::", "")
    n = re.sub(r"\(.*", "", n)
    if flt and not re.search(flt, n):
        continue
    print(f"{n[:64]:64s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} scratch={r.get('ScratchSize [bytes/lane]','?'):>5} "
          f"occ={r.get('Occupancy [waves/SIMD]','?'):>2} lds={r.get('LDS Size [bytes/block]','?'):>6}")
