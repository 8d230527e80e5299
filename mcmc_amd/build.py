"""Builds mcmc_amd/libmcmc_amd.so (HIP kernels + C ABI) for gfx950 with hipcc.

In-tree, explicit hipcc line (no JIT cache): the .so travels to the GPU box.
-ffp-contract=off is part of the numerical contract (bit-exact CPU/GPU parity).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmcmc_amd.so")
SOURCES = ["capi.hip", "capi_multiview.hip", "capi_overseg.hip", "frame_kernels.hip", "chain_kernels.hip", "multiview_kernels.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-fno-fast-math", "-Wall", "-Wno-unused-function",
         # matrix-core results straight into VGPRs (no AGPR copies): the L1 pass of Path B reads every result once
         "-mllvm", "-amdgpu-mfma-vgpr-form"]
# per-source additions.  The chain kernel is short of SGPRs (a step's wave-uniform state spills into VGPR lanes):
# cache rows fetched by vector loads instead of 26-SGPR scalar loads, and a scheduler that does not chase an
# occupancy the launch bounds already fix, measured -0.8 % (two-wave form) and -2.5 % (eight-chain form) per step.
EXTRA = {"chain_kernels.hip": ["-mllvm", "-amdgpu-scalarize-global-loads=0",
                               "-mllvm", "-amdgpu-schedule-relaxed-occupancy=true"]}
OBJ = os.path.join(HERE, "_obj")
# experiments only: MCMC_BUILD_DEFS="-DMC_SPEC_STATS" MCMC_BUILD_TAG=stats builds libmcmc_amd_stats.so beside the product
# (objects in _obj_stats); host.load() takes MCMC_AMD_LIB=<path> to load it.  The product build never sets these.
_TAG = os.environ.get("MCMC_BUILD_TAG", "")
if _TAG:
    LIB = os.path.join(HERE, "libmcmc_amd_%s.so" % _TAG)
    OBJ = os.path.join(HERE, "_obj_%s" % _TAG)
    FLAGS = FLAGS + os.environ.get("MCMC_BUILD_DEFS", "").split()


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "mcmc_amd.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """One object per source (each with its own flags, only the stale ones recompiled, in parallel), then one link."""
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdr.append(os.path.join(os.path.dirname(HERE), "include", "mcmc_amd.h"))
    hdr.append(os.path.abspath(__file__))
    newest_hdr = max(os.path.getmtime(h) for h in hdr)
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_hdr):
            continue
        cmd = [HIPCC] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in jobs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
