#!/usr/bin/env python3
"""Prints the timing fields of several bench.py JSON lines side by side (developer tool for A/B runs on one box)."""
import json
import sys


def main():
    for f in sys.argv[1:]:
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e:   # noqa: BLE001
            print(f, "ERR", e)
            continue
        c = j["config"]
        row = ["%.4f G/s" % (j["value"] / 1e9), "%.2f us" % (j["ms_per_step"] * 1e3), "frac %.3f" % j["roofline"]["frac"]]
        fr = c.get("fused_rollout")
        if isinstance(fr, dict):
            row.append("fused %.2f us" % fr.get("us_per_step", float("nan")))
        for b in c.get("baseline_configs", []) or []:
            if isinstance(b, dict):
                name = b.get("config", "?")[:2]
                one = b.get("single_steps", {}) if isinstance(b.get("single_steps"), dict) else {}
                fu = b.get("fused_rollout", {}) if isinstance(b.get("fused_rollout"), dict) else {}
                row.append("%s %s/%s" % (name, one.get("us_per_step"), fu.get("us_per_step")))
        print(f.split("/")[-1], " | ".join(str(r) for r in row))


if __name__ == "__main__":
    main()
