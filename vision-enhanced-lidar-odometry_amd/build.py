"""Build recipe of the HIP shared library (the product).  The CPU checker has its own Makefile next to its source
and is driven from tests/ and __graft_entry__.build(); nothing in this package touches it.

    python -m velo_amd.build            # via the root shim:  python -c "import velo_amd.build as b; b.build_all()"

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the authoring container; the built
.so travels to the GPU box with the gpurun snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libvelo_hip.so")
LIB_DIAG = os.path.join(CSRC, "libvelo_hip_diag.so")     # the tools' build: -DVELO_DIAGNOSTICS (stamps, counters, VELO_DEBUG_SKIP)
# Translation units, each with the flags only it gets, compiled side by side and linked into ONE shared library (round 6: ~10 s instead of
# 27 s for the product build).  velo_hip.hip is the host side of the C-ABI and defines no kernel; every kernel family is defined -- its device
# code generated -- in exactly one velo_unit_*.hip (velo_kernels.h, "translation units"); velo_lm_ag.hip is the one-launch Levenberg-Marquardt
# solve, built without machine-level loop-invariant code motion (velo_lm_ag_kernels.h says why).
SOURCES = {"velo_hip.hip": [], "velo_unit_load.hip": [], "velo_unit_assoc.hip": [], "velo_unit_lm_a.hip": [], "velo_unit_lm_b.hip": [],
           "velo_lm_ag.hip": ["-mllvm", "-disable-machine-licm"]}
KERNEL_UNITS = ["velo_unit_load.hip", "velo_unit_assoc.hip", "velo_unit_lm_a.hip", "velo_unit_lm_b.hip", "velo_lm_ag.hip"]   # the units that hold device code (tools/kernel_resources.py)
HEADERS = ["velo_kernels.h", "velo_lm_ag_kernels.h", "velo_depth_kernels.h", "velo_tri_kernels.h", "velo_device_math.h", os.path.join(ROOT, "include", "velo_hip.h"),
           # the parts of the host side (velo_hip.hip includes them in this order: one translation unit)
           "velo_host_types.inl", "velo_host_index.inl", "velo_host_assoc.inl", "velo_host_lm.inl", "velo_host_loaders.inl", "velo_host_pool.inl",
           "velo_api_context.inl", "velo_api_solve.inl", "velo_host_chain.inl", "velo_host_batch.inl", "velo_api_pose_comm.inl", "velo_api_next_rows.inl"]

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # the association compares float distances bit-for-bit with the CPU restatement: no FMA contraction anywhere
    "-ffp-contract=off", "-fno-fast-math",
    # the SLP vectoriser re-packs the plain f32 candidate sweep into v_pk_* instructions and costs 7 more VGPRs (measured: slower)
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function", "-Wno-unused-result",
]
LINK_FLAGS = [
    "--offload-arch=gfx950", "-fPIC", "-shared",
    # calls between the library's own entry points bind inside the library: the product build and the diagnostics build export the
    # same C names, and a process that loads both (the variant tests) must not have one build's batch driver call the other's loaders
    "-Wl,-Bsymbolic",
]


def _rocm() -> str:
    return os.environ.get("ROCM_PATH", "/opt/rocm")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False, extra_flags=(), diagnostics: bool = False, out: str | None = None) -> str:
    """The product library; diagnostics=True builds the tools' variant next to it (same source, -DVELO_DIAGNOSTICS: the
    VELO_DEBUG_SKIP hooks exist only there, so a leaked environment variable cannot corrupt a product registration)."""
    hipcc = shutil.which("hipcc") or os.path.join(_rocm(), "bin", "hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES]          # (the units share the kernel headers: any change rebuilds all of them)
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    out = out or (LIB_DIAG if diagnostics else LIB)      # out: an A/B build of the same source somewhere else (tools/ab_env.py, VELO_LIB_PATH)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if diagnostics:
        extra_flags = (*extra_flags, "-DVELO_DIAGNOSTICS")
    if not force and not _stale(out, deps):
        return out
    objdir = os.path.join(CSRC, "_obj", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src, own in SOURCES.items():
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, *HIPCC_FLAGS, *own, *extra_flags, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(_rocm(), "include"),
               "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((obj, subprocess.Popen(cmd)))
    bad = [obj for obj, p in jobs if p.wait() != 0]
    if bad:
        raise subprocess.CalledProcessError(1, f"hipcc -c ({', '.join(os.path.basename(b) for b in bad)})")
    link = [hipcc, *LINK_FLAGS, *[obj for obj, _ in jobs], "-o", out, "-L", os.path.join(_rocm(), "lib"), "-lrccl", "-lpthread",
            f"-Wl,-rpath,{os.path.join(_rocm(), 'lib')}"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.run(link, check=True)
    return out


def build_all(force: bool = False, verbose: bool = False):
    """product library + the diagnostics build the variant tests and dev tools load (both travel to the GPU box)"""
    build_hip(force, verbose, diagnostics=True)
    return build_hip(force, verbose)


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True, diagnostics="--diag" in sys.argv))
