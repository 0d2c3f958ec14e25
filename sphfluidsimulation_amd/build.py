"""Builds lib/libsph_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo).

    python -m sphfluidsimulation_amd.build [--force]

-ffp-contract=off and the HIP default of correctly rounded fp32 / and sqrt are part of the parity
contract: the exact-order kernels must evaluate one IEEE operation per source
operation, like oracle/sph_oracle.c.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libsph_hip.so")
SOURCES = ["sph_engine.hip", "sph_kernels_grid.hip", "sph_kernels_sweep.hip", "sph_kernels_brick_density.hip", "sph_kernels_brick_velpos.hip", "sph_kernels_slab.hip", "sph_kernels_init.hip", "sph_kernels_mesh.hip", "sph_kernels_list.hip", "sph_kernels_selftest.hip", "sph_slab_frame.hip"]
HEADERS = ["sph_internal.hpp", "sph_sweep_common.hpp", "sph_list_common.hpp", "sph_fastdiv.hpp", "sph_context.hpp", "sph_brick_impl.inc", os.path.join("..", "..", "include", "sph_hip.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    # the SLP vectoriser packs pairs of independent fp32 operations into v_pk_* instructions, which take two issue slots
    # and schedule worse than the scalar pair on gfx950 (k_velpos_list 2.11 -> 2.00 ms)
    "-fno-slp-vectorize",
    "-Wall", "-Wno-unused-function",
]

LINK_LIBS = ["-ldl"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsph_hip.so cannot be built (there is no CPU fallback)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    extra = os.environ.get("SPH_HIPCC_EXTRA", "").split()   # diagnostic builds, e.g. -DSPH_ABLATE=1
    cc = hipcc()
    # one hipcc per translation unit, in parallel (8 host cores here, 16 on the GPU box), then one link
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        cmd = [cc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, obj, subprocess.Popen(cmd)))
    objs = []
    for cmd, obj, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
        objs.append(obj)
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + LINK_LIBS + ["-o", LIB]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
