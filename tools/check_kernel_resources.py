#!/usr/bin/env python
"""No kernel of either library may use scratch (private memory spilled to HBM), and the register budget of the persistent launches is worth a look
after every change: compiles the kernel sources for gfx950 with -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints, per flavour,
every kernel's VGPRs / LDS / scratch; exit code 1 if any kernel has scratch.

    python tools/check_kernel_resources.py [--all]      (default: only the kernels with > 128 VGPRs or scratch are listed)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elasticfusion_amd import build  # noqa: E402

CSRC = os.path.join(ROOT, "elasticfusion_amd", "csrc")


def resources(src, flags):
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([build._hipcc(), *build.FLAGS, *flags, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, "x.o")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stdout[-3000:])
    out = []
    cur = None
    for line in r.stdout.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()}
            out.append(cur)
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("waves", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return out


def main():
    show_all = "--all" in sys.argv
    bad = 0
    for flavour, flags in (("default (reference rounding)", []), ("fast", ["-DEF_FAST_BUILD"])):
        print(f"== {flavour}")
        for src in build.SOURCES:
            for k in resources(src, flags):
                short = re.sub(r"\(.*", "", k["name"].replace("(anonymous namespace)::", "").replace("void ", ""))
                if k.get("scratch", 0) > 0:
                    bad += 1
                if show_all or k.get("scratch", 0) > 0 or k.get("vgprs", 0) > 128:
                    print(f"  {src:22s} {short:60s} VGPRs {k.get('vgprs', '?'):>3}  LDS {k.get('lds', '?'):>6}  waves/SIMD {k.get('waves', '?')}  scratch {k.get('scratch', '?')}")
    print("kernels with scratch:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
