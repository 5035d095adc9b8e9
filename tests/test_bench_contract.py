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
        if w.get("grid_cell_nm") is None:   # (entries measured on a non-default lookup grid are A/B side records)
            seen.add((w["envs"], w["aircraft"], w["rollout"], bool(w.get("held_hint", False))))
    assert (65536, 16, 0, True) in seen and (65536, 16, 0, False) in seen and (65536, 16, 20, False) in seen
    # BASELINE.json's other single-GPU configurations, single steps and fused
    for cfg in ((65536, 1), (8192, 16), (4096, 64)):
        assert cfg + (0, True) in seen and cfg + (20, False) in seen
    # entries of other ABI versions are never reported
    assert b.traffic_entry(65536, 16, 0, True)[0] == [w for w in j["workloads"] if (w["envs"], w["aircraft"], w["rollout"],
                                                       w["held_hint"]) == (65536, 16, 0, True) and w.get("grid_cell_nm") is None][0]["hbm_bytes_per_launch"]
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


def test_bench_eight_rank_flow_on_cpu():
    """`bench.py --gpus 8` must not fail for a reason unrelated to scaling (round-5 review, next #7): the script's whole 8-rank
    control flow — self-spawn through torch.distributed.run, contiguous env shards with distinct rank seeds, barrier-bracketed timed
    blocks, MAX over ranks, the ONE packed asynchronous all-gather per report issued after the step window, the measured blocking
    cost of that exchange on this very group, the three aggregate values and the per-rank windows — runs here over gloo on CPU
    tensors with a stand-in env that steps nothing (`--stub-env`; data: "stub").  What the 8-GPU line adds is RCCL and the kernels."""
    import subprocess
    env = dict(os.environ, ATC_DIST_BACKEND="gloo", MASTER_PORT="29547")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "40", "--warmup", "20", "--envs", "512",
                        "--repeats", "3", "--stub-env", "--no-cpu-baseline", "--no-single-env"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line on stdout (rank 0)"
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 8 and line["steps"] == 40 and line["warmup"] == 20 and line["scaling"] == "weak" and line["data"] == "stub"
    c = line["config"]
    assert c["gathered_returns_shape"] == [8, 512] and c["collective_backend"] == "gloo"
    assert len(c["rank_seeds"]) == 8 and len(set(c["rank_seeds"])) == 8
    assert len(c["rank_ms_per_step"]) == 8 and all(v > 0 for v in c["rank_ms_per_step"])
    assert abs(max(c["rank_ms_per_step"]) - line["ms_per_step"]) <= 1e-9 * line["ms_per_step"] + 1e-12
    # value = ranks x envs x steps / MAX over ranks of the step window; the two companions charge the barrier / the exchange
    assert abs(line["value"] - 8 * 512 * 40 / (line["ms_per_step"] * 40 * 1e-3)) <= 1e-6 * line["value"]
    assert 0 < c["value_between_barriers"] <= line["value"] and 0 < c["value_incl_exchange"] <= line["value"]
    assert c["exchange"] == dict(c["exchange"], collectives_per_report=1, issued=3, reports=3, in_step_window=False)
    us = c["collective"]["us"]
    assert us["world_size"] == 8 and c["exchange"]["us_blocking"] == us["packed_blocking"]["wall"] > 0
    # the report really travelled: every rank's statistics arrived in rank order on rank 0
    assert c["stub_report"]["first_return_of_each_rank"] == [1000.0 * r_ for r_ in range(8)]
    assert c["stub_report"]["first_length_of_each_rank"] == [7 * r_ for r_ in range(8)]
    assert c["stub_report"]["launches_rank0"] >= 20 + 3 * 40
