"""Build libse3conv_hip.so (gfx950) in-tree with hipcc.

    python -m se3conv3d_amd.build [--force]

The shared library lands in ``se3conv3d_amd/lib/`` (git-ignored, travels with gpurun snapshots).
hipcc cross-compiles for gfx950 without a GPU, so this also is the CPU-side "does it build" check.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
# SE3_LIB_SUFFIX=_x builds (and _lib.py loads) lib/libse3conv_hip_x.so with its objects under lib/obj_x/: variant and
# ablation builds (SE3_CXXFLAGS) of tools/*.sh never overwrite the library the tests and the bench ship with
SUFFIX = os.environ.get("SE3_LIB_SUFFIX", "")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(LIBDIR, "obj" + SUFFIX) if SUFFIX else LIBDIR
LIB = os.path.join(LIBDIR, f"libse3conv_hip{SUFFIX}.so")
SOURCES = ["geometry.hip", "edge_kernels.hip", "edge_bf16.hip", "edge_dx.hip", "gemm.hip", "gemm_bf16.hip", "prep.hip", "frames.hip", "glue.hip", "api.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "edge_bf16_body.h"), os.path.join(os.path.dirname(PKG), "include", "se3conv.h")]
ARCH = "gfx950"
# -fno-slp-vectorize: the SLP pass packs adjacent fp32 ops into v_pk_fma_f32 / v_pk_mul_f32, which issue slower than the
# two scalar ops they replace on gfx950 (edge_t_fwd at the headline shape: 0.55 ms packed, 0.44 ms scalar) and need
# 64-bit aligned register pairs (spills at 128 VGPRs).
FLAGS = os.environ.get("SE3_CXXFLAGS", "").split() + ["-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built on this machine")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _fingerprint() -> str:
    """What the objects were compiled with: a change of flags (SE3_CXXFLAGS carries the ablation / variant defines of
    tools/*.sh, some of which give deliberately wrong results) must rebuild everything, whatever the mtimes say."""
    return hashlib.sha256(" ".join([ARCH, *FLAGS]).encode()).hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    stamp = os.path.join(OBJDIR, "build_flags.sha256")
    fp = _fingerprint()
    try:
        with open(stamp) as fh:
            same_flags = fh.read().strip() == fp
    except OSError:
        same_flags = False
    force = force or not same_flags
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if jobs:
        if os.path.exists(stamp):
            os.remove(stamp)  # an interrupted build must not look up to date
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        # --no-undefined: a declaration that ran ahead of its definition must fail HERE, not as an undefined symbol when
        # the GPU box loads the library
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-Wl,--no-undefined", *objs, "-o", LIB])
    with open(stamp, "w") as fh:
        fh.write(fp + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
