#!/usr/bin/env python3
"""Dev tool (no GPU needed): registers, scratch and LDS of every kernel of the product build, from hipcc's resource-usage remarks.
A kernel that uses scratch memory at all pays ~11-15 us per launch on this system (measured twice: the tube kernel with 4 spilled
VGPRs 59 -> 76 us, the batched sweep with a 96-byte stack 18 -> 33 us), so the hot kernels must show ScratchSize 0.
Usage: python tools/kernel_resources.py [name filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import velo_amd
from velo_amd import build
flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
out = ""
for unit in build.KERNEL_UNITS:                                  # every unit that holds device code, with the flags only it gets
    src = os.path.join(build.CSRC, unit)
    cmd = ["hipcc", *flags, *build.SOURCES[unit], "-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-c", "-o", "/dev/null", src, "-Rpass-analysis=kernel-resource-usage"]
    out += subprocess.run(cmd, capture_output=True, text=True).stderr
filt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]|TotalSGPRs): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(2))
print(f"{'kernel':70s} VGPR AGPR SGPR scratch occ  LDS")
for r in rows:
    if filt in r["name"]:
        sc = r.get("ScratchSize [bytes/lane]", 0)
        flag = "  <-- scratch" if sc else ""
        print(f"{r['name'][:70]:70s} {r.get('VGPRs', 0):4d} {r.get('AGPRs', 0):4d} {r.get('TotalSGPRs', 0):4d} {sc:7d} {r.get('Occupancy [waves/SIMD]', 0):3d} {r.get('LDS Size [bytes/block]', 0):6d}{flag}")
