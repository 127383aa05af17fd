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
    ("conv_dma_v1.hip", []),
    ("conv_dma_v2.hip", []),
    ("conv_dma_v3.hip", []),
    ("conv_dma_v4.hip", []),
    ("conv_dma_v5.hip", []),
    ("conv_dma3.hip", []),
    ("conv_direct.hip", []),
    ("conv_direct3.hip", []),
    ("dec_tail.hip", []),
    ("conv_stem.hip", []),
    ("ops_misc.hip", []),
    ("resize.hip", []),
    ("fuse.hip", ["-ffp-contract=off"]),
    ("pcl.hip", ["-ffp-contract=off"]),
    ("plan.cpp", []),
    ("capi.cpp", []),
    ("host_png.cpp", []),
    ("host_jpeg.cpp", []),
    ("host_ply.cpp", []),
]
HEADERS = ["kernels.hpp", "plan.hpp", "split_fmt.hpp", os.path.join("..", "..", "include", "semdepth.h")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_hash() -> str:
    """sha256 over every file the library is built from (csrc sources + headers + the C-ABI header + the compile recipe
    below), in a fixed order.  Embedded in the library (sd_version()) at build time; _lib.load() refuses a library whose
    hash differs from the tree it is loaded from, so a stale prebuilt .so can never be used silently."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp")))
    for f in files:
        h.update(f.encode() + b"\0")
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "semdepth.h"), "rb").read())
    h.update(repr(SOURCES).encode() + ARCH.encode() + repr(LINK_FLAGS).encode() + (repr(DEV_FLAGS).encode() if DEV_FLAGS else b""))
    return h.hexdigest()[:16]


LINK_FLAGS = ["-lz", "-lpthread"]      # zlib: host_png.cpp inflates PNG streams natively
# SEMDEPTH_DEV_BUILD=1: -DSD_DEV_VARIANTS -- the closed A/B variants and the s_memtime / no-store / no-MFMA decomposition copies of the kernels
# (scripts/decompose_x3.py, dma3_timed.py, direct3_timed.py).  The shipped library carries none of them; the flag is part of the source hash.
DEV_FLAGS = ["-DSD_DEV_VARIANTS"] if os.environ.get("SEMDEPTH_DEV_BUILD") == "1" else []


def _deps(path: str, seen=None) -> list:
    """the file and every header it includes with quotes, recursively: an object is rebuilt when one of THESE changed (editing the
    C-ABI header recompiles capi / host_* only, not the ten kernel files)"""
    import re
    seen = seen if seen is not None else []
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.append(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path, encoding="utf-8", errors="replace").read(), flags=re.M):
        _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    # a library built from exactly this tree (hash sidecar written after a successful link; _lib.load() checks the hash
    # embedded in the library itself) needs nothing, even when the object directory did not travel with the tree
    # (gpurun ships the .so but not csrc/build/): no one-minute rebuild at the start of every GPU-box command
    digest = source_hash()
    stamp = LIB + ".hash"
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return LIB
    hash_obj = os.path.join(OBJ, "capi.hash")
    hash_changed = not os.path.exists(hash_obj) or open(hash_obj).read().strip() != digest
    jobs = []
    objs = []
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        objs.append(op)
        if force or _stale(op, _deps(sp)) or (src.endswith(".hip") and not os.path.exists(op + ".remarks")) or (src == "capi.cpp" and hash_changed):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", sp, "-o", op] + extra + DEV_FLAGS
            if src.endswith(".hip"):
                cmd.append("-Rpass-analysis=kernel-resource-usage")      # registers / spills / LDS per kernel -> <obj>.remarks
            if src == "capi.cpp":
                cmd.append(f'-DSD_SOURCE_HASH="{digest}"')
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
        if "-Rpass-analysis=kernel-resource-usage" in cmd:
            with open(cmd[cmd.index("-o") + 1] + ".remarks", "w") as f:
                f.write("\n".join(ln for ln in r.stderr.splitlines() if "remark:" in ln))
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(max(4, min(os.cpu_count() or 4, 8)), len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or _stale(LIB, objs) or hash_changed:
        tmp = LIB + f".tmp{os.getpid()}"          # link aside, then rename: no process ever maps a half-written library
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs + LINK_FLAGS)
        os.replace(tmp, LIB)
    with open(hash_obj, "w") as f:
        f.write(digest)
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


def kernel_resources() -> dict:
    """{mangled kernel name: {"VGPRs": n, "VGPRs Spill": n, "SGPRs Spill": n, "ScratchSize [bytes/lane]": n, "LDS Size [bytes/block]": n, ...}}
    from the resource-usage remarks hipcc wrote while the objects of THIS tree were compiled (csrc/build/*.remarks; empty when the
    object directory did not travel, e.g. on the GPU box).  tests/test_abi.py holds the conv kernels to zero spilled VGPRs: a spill
    in a register-capped instantiation halves a layer's speed without failing any numerics test (round 3)."""
    import re
    out = {}
    if not os.path.isdir(OBJ):
        return out
    for f in sorted(os.listdir(OBJ)):
        if not f.endswith(".remarks"):
            continue
        cur = None
        for line in open(os.path.join(OBJ, f)):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {"file": f[:-len(".o.remarks")]})
                continue
            m = re.search(r"remark:\s+([\w \[\]/]+?): (\d+)", line)
            if m and cur is not None:
                cur[m.group(1)] = int(m.group(2))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
