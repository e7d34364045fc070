#!/usr/bin/env python3
"""Register / scratch / spill table of every kernel (hipcc -Rpass-analysis=kernel-resource-usage; runs
without a GPU).  Usage: python scripts/kernel_resources.py > profiles/<tag>_kernel_resources.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pantr-cdc2023-experiments_amd")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950",
       "-Rpass-analysis=kernel-resource-usage", '-DPANTR_SOURCE_HASH="x"', "-c", os.path.join(PKG, "csrc/pantr_capi.hip"),
       "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|"
                  r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s+(\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": demangle(v).replace("pantr::", "")}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" [")[0]] = v
print("# hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage csrc/pantr_capi.hip")
print("# static LDS only: the solve kernels add their dynamic LDS at launch (DESIGN.md section 3).  SGPR spills go to")
print("# VGPR lanes (v_writelane / v_readlane), VGPR spills beyond the AGPRs to scratch memory.")
print(f"{'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'sgprSp':>7s} {'vgprSp':>7s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    print(f"{r['name'][:78]:78s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} "
          f"{r.get('ScratchSize','?'):>8s} {r.get('SGPRs Spill','?'):>7s} {r.get('VGPRs Spill','?'):>7s} "
          f"{r.get('Occupancy','?'):>4s} {r.get('LDS Size','?'):>7s}")
