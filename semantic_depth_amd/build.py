"""Build libsemdepth.so (HIP, gfx950) in-tree.  `python -m semantic_depth_amd.build` or __graft_entry__.build().

hipcc cross-compiles without a GPU.  fuse.hip / pcl.hip are compiled with -ffp-contract=off because their
arithmetic must round exactly like numpy / OpenCV (no fused multiply-add); the conv engine uses the default.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libsemdepth.so")
ARCH = "gfx950"

SOURCES = [
    ("conv_igemm.hip", []),
    ("conv_split.hip", []),
    ("conv_dma.hip", []),
    ("conv_direct.hip", []),
    ("conv_stem.hip", []),
    ("ops_misc.hip", []),
    ("resize.hip", []),
    ("fuse.hip", ["-ffp-contract=off"]),
    ("pcl.hip", ["-ffp-contract=off"]),
    ("plan.cpp", []),
    ("capi.cpp", []),
]
HEADERS = ["kernels.hpp", "plan.hpp", "split_fmt.hpp", os.path.join("..", "..", "include", "semdepth.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    # a library newer than every source needs nothing, even when the object directory did not travel with the tree
    # (gpurun ships the .so but not csrc/build/): no one-minute rebuild at the start of every GPU-box command
    if not force and not _stale(LIB, [os.path.join(CSRC, src) for src, _ in SOURCES] + hdrs):
        return LIB
    jobs = []
    objs = []
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        objs.append(op)
        if force or _stale(op, [sp] + hdrs):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", sp, "-o", op] + extra
            if src.endswith(".cpp"):
                cmd += ["-x", "hip"]
                cmd = cmd[:1] + ["-x", "hip"] + [c for c in cmd[1:] if c not in ("-x", "hip")]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or _stale(LIB, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
