"""CPU checks of bench.py's bookkeeping (no GPU): the algorithmic byte counts SURVEY §8(d) defines, the traffic table the
roofline record is filled from, and the command line the driver uses."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_follow_survey_8d():
    b = _bench()
    # single step: 96 N + 13 (N = 1: 109, N = 16: 1 549, N = 64: 6 157)
    assert b.algorithmic_bytes_per_env_step(1) == 109
    assert b.algorithmic_bytes_per_env_step(16) == 1549
    assert b.algorithmic_bytes_per_env_step(64) == 6157
    # T fused steps: the 40-byte state term is paid once per T, a held action block once per `hold`
    assert abs(b.algorithmic_bytes_per_env_step(16, 20, 20) - ((44 + 2 + 0.6) * 16 + 13)) < 1e-9
    assert b.algorithmic_bytes_per_env_step(16, 1, 1) == b.algorithmic_bytes_per_env_step(16)


def test_traffic_table_is_consistent():
    """profiles/pmc_traffic.json: every entry's bytes follow from its two counters (FETCH_SIZE in 64-byte units on gfx950:
    x 2 against the KB the tool prints, + WRITE_SIZE), its ratio from the algorithmic bytes, and the headline entry exists
    for both launch modes."""
    b = _bench()
    sys.path.insert(0, os.path.join(ROOT, "atc-reinforcement-learning_amd"))
    from atc_hip import layout as L
    j = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    seen = set()
    for w in j["workloads"]:
        assert w["abi"] == L.ABI_VERSION, "the reported table must have been measured on the library's current ABI"
        hbm = (w["FETCH_SIZE_KB_raw"] * 2 + w["WRITE_SIZE_KB"]) * 1024
        assert abs(hbm - w["hbm_bytes_per_launch"]) <= 1024, w
        T = w["rollout"] or 1
        alg = b.algorithmic_bytes_per_env_step(w["aircraft"], T, T) * w["envs"] * T
        assert abs(alg - w["algorithmic_bytes_per_launch"]) < 1, w
        assert abs(w["hbm_bytes_per_launch"] / w["algorithmic_bytes_per_launch"] - w["ratio"]) < 2e-3, w
        seen.add((w["envs"], w["aircraft"], w["rollout"], bool(w.get("held_hint", False))))
    assert (65536, 16, 0, True) in seen and (65536, 16, 0, False) in seen and (65536, 16, 20, False) in seen
    # BASELINE.json's other single-GPU configurations, single steps and fused
    for cfg in ((65536, 1), (8192, 16), (4096, 64)):
        assert cfg + (0, True) in seen and cfg + (20, False) in seen
    # entries of other ABI versions are never reported
    assert b.traffic_entry(65536, 16, 0, True)[0] == [w for w in j["workloads"] if (w["envs"], w["aircraft"], w["rollout"],
                                                       w["held_hint"]) == (65536, 16, 0, True)][0]["hbm_bytes_per_launch"]
    for w in j.get("superseded", []):
        assert w["abi"] != L.ABI_VERSION


def test_driver_command_line_parses():
    """`python bench.py --gpus N --steps K --warmup W` (the driver's call) and the no-flag default."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert 'add_argument("%s"' % flag in src
    assert 'default=ENVS_PER_GPU' in src and "ENVS_PER_GPU = 65536" in src.replace("65_536", "65536")


def test_store_only_reference_never_costs_the_bench_line():
    """bench.py runs tools/ubench/write_bw.hip (built by build()) next to the fused launch to say what a store-only launch takes on
    the box at hand.  It is a side record: other shapes than 16 aircraft have none, and a failing executable (here: no GPU) comes back
    as an error note, never as an exception."""
    b = _bench()
    assert b.store_only_reference(65536, 1, 20) is None
    rec = b.store_only_reference(4096, 16, 2)
    exe = os.path.join(ROOT, "atc-reinforcement-learning_amd", "atc_hip", "ubench_write_bw")
    if not os.path.exists(exe):
        assert rec is None
    else:
        assert isinstance(rec, dict) and ("error" in rec or rec["envs"] == 4096)
