#!/usr/bin/env python3
"""Turns the passes of tools/pmc_traffic_all.sh into the entries of profiles/pmc_traffic.json (stdout): per workload the average
FETCH_SIZE / WRITE_SIZE of the k_step dispatches, HBM bytes per launch = (FETCH_SIZE x 2 + WRITE_SIZE) KB — FETCH_SIZE counts
64-byte units on gfx950 against the KB the tool prints (MI355X_MICROARCH.md, calibrated on this kernel's known read bytes) —
the algorithmic bytes of the same launch and their ratio, keyed by the ABI version of the library that was measured."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "atc-reinforcement-learning_amd"))
import bench  # noqa: E402
from atc_hip import layout as L  # noqa: E402

tag = sys.argv[1]
WORK = [("n16_65536", 65536, 16, 0, True), ("n16_65536_nohint", 65536, 16, 0, False), ("n16_65536_roll", 65536, 16, 20, False),
        ("n1_65536", 65536, 1, 0, True), ("n1_65536_roll", 65536, 1, 20, False), ("n16_8192", 8192, 16, 0, True),
        ("n16_8192_roll", 8192, 16, 20, False), ("n64_4096", 4096, 64, 0, True), ("n64_4096_roll", 4096, 64, 20, False),
        ("n64_32768", 32768, 64, 0, True), ("n16_262144", 262144, 16, 0, True),
        # round-5 review, next #3: 65 536 x 1 single steps on the 0.125 nm lookup grid (3.2 MB: the eight L2s hold it) instead of the
        # batch's default 0.0625 nm (11.6 MB) — an A/B of traffic against time, reported as a side record
        ("n1_65536_g0125", 65536, 1, 0, True, 0.125)]


def avg(d, counter):
    tot, n = 0.0, 0
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if "k_step" in row.get("Kernel_Name", "") and row["Counter_Name"] == counter:
                tot += float(row["Counter_Value"])
                n += 1
    return (tot / n, n) if n else (None, 0)


out = []
for name, B, N, roll, held, *rest in WORK:
    grid_cell = rest[0] if rest else None
    d = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, name))
    fetch, nf = avg(d, "FETCH_SIZE")
    write, nw = avg(d, "WRITE_SIZE")
    if fetch is None or write is None:
        continue
    hbm = int(round((fetch * 2 + write) * 1024))
    alg = bench.algorithmic_bytes_per_env_step(N, roll or 1, roll or 1) * B * (roll or 1)
    W = 1 << max(0, (N - 1).bit_length())
    ratio = hbm / alg
    # What the x 2 is calibrated on (MI355X_MICROARCH.md, HBM / rocprofv3 section): streaming reads of 16 bytes per lane.  A
    # launch whose reads are mostly narrower (4-byte speed, 12-byte action / last-action records, 8-byte lookup cells) or
    # re-hit lines the Infinity Cache still holds from the previous launch can come out BELOW the algorithmic bytes; such an
    # entry is a lower bound of the traffic, not evidence that algorithmic bytes were skipped.
    note = None
    if ratio < 1.0:
        note = ("ratio < 1: FETCH_SIZE x 2 is calibrated on 16-byte-per-lane streaming reads; this launch's reads are %s, and "
                "write-once outputs / re-read state of a %d MB working set are partly absorbed by the 256 MiB Infinity Cache — "
                "read the figure as a lower bound" % ("mostly 4- / 12-byte records of one-aircraft envs" if N == 1 else
                "16-byte state records plus 12-byte action records once per %d steps" % (roll or 1),
                bench.working_set_bytes(B, N, roll or 1) >> 20))
    out.append({"abi": L.ABI_VERSION, "grid_cell_nm": grid_cell, "read_correction_note": note, "envs": B, "aircraft": N, "rollout": roll, "held_hint": held,
                "kernel": "k_step<%d,false,%s,%s>%s" % (W, "false" if roll else "true", "true" if N == W and (B * W) % 256 == 0 else "false",
                                                     " T=%d hold=%d" % (roll, roll) if roll else ""),
                "FETCH_SIZE_KB_raw": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1), "hbm_bytes_per_launch": hbm,
                "algorithmic_bytes_per_launch": alg, "ratio": round(ratio, 4), "dispatches": min(nf, nw),
                "source": "profiles/%s_pmc_traffic.json (tools/pmc_traffic_all.sh %s: two separate rocprofv3 --pmc passes per workload, "
                          "--kernel-trace only; FETCH_SIZE x 2 + WRITE_SIZE, average over the k_step dispatches%s)"
                          % (tag, tag, "; 19 launches in 20 carry ATC_M_ACTIONS_HELD" if held and not roll else "")})
print(json.dumps({"workloads": out}, indent=1))
