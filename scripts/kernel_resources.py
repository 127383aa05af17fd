#!/usr/bin/env python3
"""dev tool: VGPRs / spills / scratch / LDS of every kernel of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/kernel_resources.py conv_dma.hip [--all]   (default: only kernels that spill or use scratch)"""
import os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "semantic_depth_amd", "csrc")
src = sys.argv[1]
extra = ["-ffp-contract=off"] if src in ("fuse.hip", "pcl.hip") else []
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(root, src), "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([\w \[\]/]+?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    spill = v.get("VGPRs Spill", 0) + v.get("SGPRs Spill", 0) + v.get("ScratchSize [bytes/lane]", 0)
    if "--all" in sys.argv or spill:
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{name[:110]:110s} VGPR {v.get('VGPRs', 0):3d} AGPR {v.get('AGPRs', 0):3d} vspill {v.get('VGPRs Spill', 0):3d} sspill {v.get('SGPRs Spill', 0):3d} "
              f"scratch {v.get('ScratchSize [bytes/lane]', 0):4d} occ {v.get('Occupancy [waves/SIMD]', 0)} LDS {v.get('LDS Size [bytes/block]', 0)}")
if r.returncode:
    print(r.stderr[-3000:])
