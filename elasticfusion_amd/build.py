"""Builds libefusion_hip.so (hand-written HIP, gfx950 only) in-tree with hipcc.

    python -m elasticfusion_amd.build [--force]
    python -m elasticfusion_amd.build --variant <name> [-D...]      (development / A-B builds: VARIANTS below)

ONE library is shipped (csrc/ef_build.hpp):
    libefusion_hip.so        REFERENCE ROUNDING (no fused multiply-add, the reference's summation order) — every result bit for bit the
                             reference's own sources compiled without contraction; what libefusion.so links and bench.py times
Round 4's build (-DEF_FAST_BUILD: fused multiply-adds + the fast summation order; no faster since round 5 and outside the 1e-4 m / 1e-4 rad
bar on 15 of 113 one-frame checkpoints, profiles/r05_parity_factorial.json) is no longer built by build(): it is the development variant
"fast" (`--variant fast` -> libefusion_hip_fast.so), kept for the parity factorial and the `-m "gpu and fastbuild"` tests.

-ffp-contract=off: fused multiply-adds appear only where the kernels spell them out (fmaf), which is what
makes the integer-valued stages (u16/u8/i16 pyramids, correspondences, index maps) bit-exact against the
CPU oracle (DESIGN.md "Numerics").
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libefusion_hip.so")            # reference rounding (the default)
FAST_LIB = os.path.join(HERE, "libefusion_hip_fast.so")   # build_variant("fast"): NOT part of build() since round 6
NOFMA_LIB = LIB                                           # (rounds 1-4 kept the reference rounding in a test-only libefusion_hip_nofma.so: now the default)
VARIANTS = {   # python -m elasticfusion_amd.build --variant <name>: libefusion_hip_<name>.so
    "fast": ["-DEF_FAST_BUILD"],                               # round 4's shipped build: fused multiply-adds + the fast order (DESIGN_fast_build.md)
    "reforder": ["-DEF_FAST_BUILD", "-DEF_REF_ORDER"],         # fused multiply-adds + the reference's order (round 3's product; parity factorial)
    "nofma_fast": ["-DEF_FORCE_FAST_ORDER"],                   # no fused multiply-adds + the fast order (parity factorial)
    "clocks": ["-DEF_STAGE_CLOCKS"],                           # phase clocks of the default build's tracker (tools/small_clocks.py)
    "fast_clocks": ["-DEF_FAST_BUILD", "-DEF_STAGE_CLOCKS"],   # phase clocks of the fast build's persistent tracker (tools/fast_clocks.py)
    "ldlt_wave": ["-DEF_LDLT_WAVE"],                           # A/B: rounds 2-4's 6x6 factorisation, one matrix element per lane (ef_solve_dev.hpp)
    "ldlt_wave_clocks": ["-DEF_LDLT_WAVE", "-DEF_STAGE_CLOCKS"],
    "nopairs": ["-DEF_NO_VISIT_PAIRS"],                        # A/B: one visit per lane in the normal-equation rows (rounds 1-5; round 6 evaluates two, packed)
    "nopairs_clocks": ["-DEF_NO_VISIT_PAIRS", "-DEF_STAGE_CLOCKS"],
    "p_icp": ["-DEF_RT_WITH_PAIRS_ICP"],                       # A/B: the level-resident parts of the persistent launch with two visits per lane (default: one)
    "p_search": ["-DEF_RT_WITH_PAIRS_SEARCH"],
    "p_rgb": ["-DEF_RT_WITH_PAIRS_RGB"],
    "p_all": ["-DEF_RT_WITH_PAIRS_ICP", "-DEF_RT_WITH_PAIRS_SEARCH", "-DEF_RT_WITH_PAIRS_RGB"],
    "shfl": ["-DEF_RT_SHFL_REDUCE"],                           # A/B: the wave-level sums of the persistent launch through ds_bpermute (rounds 1-5) instead of DPP / permlane moves
    "sepsc": ["-DEF_SEPARATE_SIN_COS"],                        # A/B: cos(theta) and sin(theta) of the update step as two calls (rounds 1-5) instead of one sincos
    "pyrstages": ["-DEF_PYR_STAGES"],                          # A/B: one launch per pyramid stage (step l -> l + 1 with the maps + Sobel of level l) instead of two steps + one maps / Sobel launch (measured no faster: off)
    "deep": ["-DEF_DEEP_PIPE"],                                # A/B: k_se3_accum's multi-round path (1280x960) two rounds deep instead of one (measured slower: off)
    "sepinputs": ["-DEF_SEPARATE_INPUTS"],                     # A/B: depth pre-processing and the tracker's model maps as two launches instead of one (k_frame_inputs)
    "sepmerge": ["-DEF_SEPARATE_MERGE"],                       # A/B: the fusion's update pass as its own launch (k_merge) instead of riding on the second index splat
    "pre_vpair": ["-DEF_PRE_VPAIR"],                           # A/B: the bilateral filter's two pixels per lane four rows apart (round 5) instead of side by side
    "prep_late": ["-DEF_RT_PREPARE_LATE"],                     # A/B: the sigma-independent half of the photometric rows behind exchange A (rounds 1-5) instead of beside it
    "p_nostream": ["-DEF_RT_NO_PAIRS_STREAM"],                 # ... and the streaming path with one
    "sepscan": ["-DEF_SEPARATE_SCAN"],                         # A/B: clean()'s scan of the rows' counts as its own launch (rounds 1-5) instead of inside the scatter's workgroups
    "endwave": ["-DEF_END_ONE_WAVE"],                          # A/B: k_track_ref_end's two tails behind resultRt on one wavefront (round 5) instead of two
    "pre_pertap": ["-DEF_PRE_SCALE_PER_TAP"],                  # A/B: the bilateral filter scales the tile value back at every tap instead of once behind the loop
    "assoc_late": ["-DEF_ASSOC_LATE_LOADS"],                   # A/B: k_associate asks for the filtered depth, the colour and the index-map texels behind its test on the raw depth (rounds 1-5)
    "resolveall": ["-DEF_RESOLVE_ALL_MAPS"],                   # A/B: the frame's first predictIndices resolves all four index maps (rounds 1-5) instead of the three the association taps
    "splat_early": ["-DEF_SPLAT_EARLY_LOADS"],                 # A/B: the surface splat asks for all three streams of a surfel at once (default: colour / time and normal only for stable surfels)
    "resolvetally": ["-DEF_RESOLVE_TALLY"],                    # A/B: denseEnough()'s tally by one atomicAdd per sample from the prediction's resolve pass (rounds 1-5) instead of by the next frame's model-map workgroups
    "modeone": ["-DEF_FT_MODE_ONE"],                           # A/B: the admission verdict of the persistent launch polled on one word (rounds 4-5) instead of 64 copies
    "modeone_clocks": ["-DEF_FT_MODE_ONE", "-DEF_STAGE_CLOCKS"],
    "norepl": ["-DEF_FT_REPL=1", "-DEF_FT_REPL_A=1"],          # A/B: one copy of the all-to-all exchange areas of the persistent launch (rounds 4-5) instead of 64 (totals) / 2 (records)
    "norepl_clocks": ["-DEF_FT_REPL=1", "-DEF_FT_REPL_A=1", "-DEF_STAGE_CLOCKS"],
    "lanesweep": ["-DEF_FT_LANE_SWEEP"],                       # A/B: the sweeps of the exchanges with every lane on its own four granules (rounds 4-5) instead of every load instruction on 64 consecutive ones
    "lanesweep_clocks": ["-DEF_FT_LANE_SWEEP", "-DEF_STAGE_CLOCKS"],
    "pipepoll": ["-DEF_FT_PIPELINED_POLL"],                    # A/B: the exchanges of the persistent launch keep two polls in flight instead of one (measured 6 % slower: off)
    "pipepoll_clocks": ["-DEF_FT_PIPELINED_POLL", "-DEF_STAGE_CLOCKS"],
    "alltaps": ["-DEF_CLEAN_ALL_TAPS"],                        # A/B: clean()'s keep-test asks for the taps of elements the time rules decide anyway (rounds 1-5)
    "r6m": ["-DEF_SEPARATE_SCAN", "-DEF_END_ONE_WAVE"],        # A/B: both of the above = the launches of commit 7b89629
}
SHIM_LIB = os.path.join(HERE, "libefusion.so")          # class ElasticFusion (include/ElasticFusion.h) over the C ABI
SOURCES = ["ef_track_kernels.hip", "ef_map_kernels.hip", "ef_context.hip", "ef_ferns.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(SHIM_LIB):
        return True
    t = os.path.getmtime(LIB)
    root = os.path.dirname(HERE)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".inc", ".hpp", ".h"))]
    deps += [os.path.join(root, "include", "ef_hip.h"), os.path.join(root, "include", "ElasticFusion.h"), os.path.join(root, "include", "efusion_klg.hpp"), os.path.join(root, "include", "efusion_jpeg.hpp"),
             os.path.join(root, "tools", "efusion_replay.cpp"), __file__]
    replay = os.path.join(HERE, "efusion_replay")
    return not os.path.exists(replay) or any(os.path.getmtime(d) > min(t, os.path.getmtime(replay)) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [_hipcc(), *FLAGS, *os.environ.get("EF_HIPCC_FLAGS", "").split(), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    # libefusion.so: host-only C++ (compiled by hipcc for the shared __host__ __device__ linear-algebra header)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", SHIM_LIB,
           os.path.join(CSRC, "efusion_shim.hip"), "-L" + HERE, "-lefusion_hip", "-lz", "-ldl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"libefusion.so build failed:\n{r.stdout}")
    # headless replay front-end (plain g++: it only sees include/ElasticFusion.h)
    replay = os.path.join(HERE, "efusion_replay")
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(os.path.dirname(HERE), "tools", "efusion_replay.cpp"), "-o", replay, "-L" + HERE,
           "-lefusion", "-lefusion_hip", "-lz", "-ldl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link," + HERE]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"efusion_replay build failed:\n{r.stdout}")
    return LIB


def build_variant(name: str, extra_flags: list[str]) -> str:
    """TEST / DEVELOPMENT builds of libefusion_hip under another name (libefusion_hip_<name>.so, selected with EF_HIP_LIB in the Python
    harness; never loaded by the product).  A name of VARIANTS brings its flags; extra_flags are added."""
    extra_flags = VARIANTS.get(name, []) + list(extra_flags)
    target = os.path.join(HERE, f"libefusion_hip_{name}.so")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", f"_{name}.o"))
        procs.append((src, subprocess.Popen([_hipcc(), *FLAGS, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj], stdout=subprocess.PIPE,
                                            stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target, *objs, "-Wl,-rpath,/opt/rocm/lib"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return target


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m elasticfusion_amd.build --variant valu -DEF_ACCUM_VALU
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
