#!/usr/bin/env python3
"""Developer tool: static instruction attribution of a kernel by source function.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -gline-tables-only \
        -Iinclude atc-reinforcement-learning_amd/csrc/atc_step.hip -o /tmp/lines.s
  python tools/isa_attribution.py /tmp/lines.s _Z6k_stepILi16ELb0E
Every instruction is charged to the innermost inlined function of its .loc (and, second table, to the statement of the
step body it was inlined into)."""
import collections
import os
import re
import sys

asm, kernel = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "atc-reinforcement-learning_amd", "csrc")
files = {"atc_step.hip": os.path.join(root, "atc_step.hip"), "atc_device.h": os.path.join(root, "atc_device.h"),
         "atc_wave.h": os.path.join(root, "atc_wave.h")}
func_of = {}
for short, path in files.items():
    cur, table = None, {}
    for n, line in enumerate(open(path), 1):
        m = re.match(r"^(?:template.*>\s*)?(?:static\s+)?(?:__device__|__global__)[^;{]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", line)
        if m and not line.startswith(" "):
            cur = m.group(1)
        m2 = re.match(r"^\s*(?:static\s+)?(?:void|float|int|bool|[A-Za-z_0-9]+)\s+(run|[a-z_0-9]+)\s*\(.*\)\s*\{\s*$", line)
        if m2 and line.startswith("    ") and "run" == m2.group(1):
            cur = "PairScan16::run"
        if line.startswith("k_step(") or line.startswith("k_step_pipe("):
            cur = line.split("(")[0]
        table[n] = cur
    func_of[short] = table
loc_re = re.compile(r"; ((?:[\w/.\-]+):(\d+):\d+)((?: @\[ [^\]]+ \])*)")
inner, outer, kinds = collections.Counter(), collections.Counter(), collections.Counter()
inside, cur_inner, cur_outer = False, "?", "?"
for line in open(asm):
    if line.startswith(kernel):
        inside = True
        continue
    if not inside:
        continue
    s = line.strip()
    if s.startswith(".loc"):
        m = loc_re.search(s)
        if m:
            f, ln = os.path.basename(m.group(1).split(":")[0]), int(m.group(2))
            cur_inner = "%s" % (func_of.get(f, {}).get(ln) or f)
            chain = re.findall(r"@\[ ([\w/.\-]+):(\d+):\d+", m.group(3))
            cur_outer = "top:%d" % ln if f == "atc_step.hip" else cur_inner
            for cf, cl in chain:  # outermost frame inside step_part_a/b
                cf = os.path.basename(cf)
                fn = func_of.get(cf, {}).get(int(cl))
                if fn in ("step_part_a", "step_part_b", "k_step"):
                    cur_outer = "%s:%s" % (fn, cl)
                    break
        continue
    if not s or s.startswith((";", ".", "_Z")) or s.endswith(":"):
        continue
    op = s.split()[0]
    if op == "s_endpgm":
        break
    kind = "VALU" if op.startswith("v_") else "SALU/SMEM" if op.startswith("s_") else "MEM/LDS"
    kinds[kind] += 1
    inner[(cur_inner, kind)] += 1
    outer[(cur_outer, kind)] += 1
print("totals:", dict(kinds))
for title, table in (("by innermost function", inner), ("by statement of the step body", outer)):
    print("\n" + title)
    names = sorted({k[0] for k in table}, key=lambda n: -table[(n, "VALU")])
    for n in names[:40]:
        print("  %-34s VALU %4d  SALU %4d  MEM %3d" % (n, table[(n, "VALU")], table[(n, "SALU/SMEM")], table[(n, "MEM/LDS")]))
