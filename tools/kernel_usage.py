#!/usr/bin/env python3
"""Compiles csrc/atc_step.hip with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
VGPRs / SGPR spills / scratch / occupancy.  Usage: python tools/kernel_usage.py [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "atc-reinforcement-learning_amd", "csrc", "atc_step.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-fPIC", "-shared",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/atc_usage.so", SRC] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([\w \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    if "error" in line:
        print(line)
for k, r in rows.items():
    print("%-34s VGPR %3d  AGPR %2d  SGPR-spill %3d  scratch %3d  occ %d  LDS %d" % (
        k, r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("SGPRs Spill", 0), r.get("ScratchSize [bytes/lane]", 0),
        r.get("Occupancy [waves/SIMD]", 0), r.get("LDS Size [bytes/block]", 0)))
