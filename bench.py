#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched AtcGym.step() hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: run under torch.distributed.run, one rank per GPU)

One "step" = one pass of the hot path (one atc_step launch) over this GPU's batch: 65 536 envs x 16 aircraft, the
configuration BASELINE.json's metric is quoted on ("env-steps/sec aggregate @16 aircraft/env, 64k envs").  Weak scaling:
every rank owns 65 536 envs (8 ranks = config C5, 524 288 envs); no collective on the step path, one RCCL all-gather of the
per-env episode returns at the end of the timed rollout.  Inputs (sector, state, a ring of pre-sampled action tensors:
U(-1,1) fp32, re-sampled every 20 steps like learning/atc-gym-demo.py:18-19) are resident in HBM before the timed region;
envs auto-reset on done.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

ENVS_PER_GPU = 65536
AIRCRAFT = 16
HOLD = 20            # action re-sampling interval [steps]
HBM_PEAK_GBS = 8000  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
INFINITY_CACHE_BYTES = 256 << 20  # MI355X_MICROARCH.md: 256 MiB memory-side cache (MALL) in front of HBM


def working_set_bytes(B, N, T=1):
    """Bytes one launch touches: aircraft records (ac 16 + alt 8 + last_act 16), env records (16 + 32 per env), ONE action
    tensor (12 per aircraft; the other ring tensors are cold) and the outputs of its T steps (obs 40 + flags 2 per aircraft,
    reward 4 + done 1 per env)."""
    return B * N * (40 + 12 + 42 * T) + B * (48 + 5 * T)


def algorithmic_bytes_per_env_step(n, fused_steps=1, hold=1):
    """SURVEY.md §8(d) / BASELINE.md §3: 96 B per aircraft (32 read: state 20 + action 12; 64 write: state 20 + obs 40 +
    reward 4) + 13 B per env (timesteps r/w 8, done 1, flags 4).  "In a T-step fused rollout the state term (40 N) is paid
    once per T" (SURVEY §8d); an action block held for `hold` steps is read once per `hold`."""
    return (44 + 40.0 / fused_steps + 12.0 / hold) * n + 13


def store_only_reference(B, N, T):
    """What a launch that ONLY WRITES the fused launch's outputs takes on this box, measured now (tools/ubench/write_bw.hip, a
    separate executable built by build(); boxes of one pool differ by 25 % on it): the ideal write stream, the launch's store pattern
    alone ([T][B][N][10] rows, flags, reward, done), and that pattern with 400 FMAs and one gather per lane-step.  A side record."""
    import json
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "atc-reinforcement-learning_amd", "atc_hip", "ubench_write_bw")
    if N != 16 or not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "--json", str(B), str(T)], capture_output=True, text=True, timeout=120)
        rec = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001 — a reference figure, never a reason to lose the bench line
        return {"error": repr(e)}
    rec["source"] = "tools/ubench/write_bw.hip --json %d %d" % (B, T)
    return rec


def _time_oracle(n_aircraft, B, threads, seconds_target, scn=None):
    import numpy as np
    from envs.atc import scenarios
    from oracle import oracle as O
    if scn is None:
        scn = scenarios.LOWWDense() if n_aircraft > 16 else scenarios.LOWW(random_entrypoints=n_aircraft > 1)
    comp = scenarios.compile_scenario(scn)
    O.set_threads(threads)
    env = O.OracleEnv(comp, B, n_aircraft, O.make_params(auto_reset=True, seed=0), np.float32)
    rng = np.random.default_rng(0)
    acts = [rng.uniform(-1, 1, (B, n_aircraft, 3)).astype(np.float32) for _ in range(4)]
    for t in range(5):
        env.step(acts[0])
    steps = 0
    t0 = time.perf_counter()
    while True:
        env.step(acts[(steps // HOLD) % 4])
        steps += 1
        if steps % 10 == 0 and time.perf_counter() - t0 > seconds_target:
            break
    dt = time.perf_counter() - t0
    O.set_threads(1)
    return B * steps / dt, steps, dt


def cpu_side_baseline(n_aircraft, scn, seconds_target=3.0):
    """The same sampler as `cpu_baseline` for one of BASELINE.json's other configurations: ONE core, ~3 s."""
    Bs = max(64, 32768 // max(1, n_aircraft))
    v, steps, dt = _time_oracle(n_aircraft, Bs, 1, seconds_target, scn)
    return {"value": v, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "%d envs x %d aircraft x %d steps (%.1f s) of the same workload through oracle/ (fp32 C port of the "
                      "reference step, 1 thread)" % (Bs, n_aircraft, steps, dt)}


def single_env_cpu_oracle(n_steps=100000):
    """The reference's own protocol (learning/atc-gym-compute-performance.py:10-19: ONE env x ONE aircraft, 100 000 x step(one
    fixed sampled action), no reset on done, FPS = N / wall time) through the CPU oracle stepped 1 x 1 from Python — the
    compiled-CPU counterpart of `single_env` (the GPU-backed drop-in AtcGym), on 1 core of this host."""
    import numpy as np
    from envs.atc import scenarios
    from oracle import oracle as O
    comp = scenarios.compile_scenario(scenarios.LOWW())
    out = {}
    for name, dt_ in (("f64", np.float64), ("f32", np.float32)):
        env = O.OracleEnv(comp, 1, 1, O.make_params(auto_reset=False, keep_active=True), dt_)
        action = np.random.default_rng(0).uniform(-1, 1, (1, 1, 3)).astype(dt_)
        for _ in range(200):
            env.step(action)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            env.step(action)
        out[name] = n_steps / (time.perf_counter() - t0)
    return out


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box reports 256 logical
    CPUs but the container's cpu.max allows 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(n_aircraft, seconds_target=12.0):
    """The fp32 CPU oracle (a scalar C port of the reference's step, oracle/) timed on the GPU box's host cores on a bounded
    sample of the same workload (same sector, spawn lattice, action protocol, auto-reset).  Primary figure: ONE core
    (scalar port, `cores` = 1); `all_cores` adds the same loop under OpenMP over envs on every host core."""
    v1, steps1, dt1 = _time_oracle(n_aircraft, 2048, 1, seconds_target)
    ncpu = host_cores()
    out = {"value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": "%d envs x %d aircraft x %d steps (%.1f s) of the same workload through oracle/ (fp32 C port of the "
                     "reference step, gcc -O2, 1 thread); host: %d logical CPUs, %d usable (affinity/cgroup quota)"
                     % (2048, n_aircraft, steps1, dt1, os.cpu_count() or 1, ncpu)}
    if ncpu > 1:
        try:
            Bm = 16384
            vm, stepsm, dtm = _time_oracle(n_aircraft, Bm, ncpu, 6.0)
            out["all_cores"] = {"value": vm, "unit": "env-steps/s", "cores": ncpu,
                                "sample": "%d envs x %d aircraft x %d steps (%.1f s), OpenMP over envs" % (Bm, n_aircraft, stepsm, dtm)}
        except Exception as exc:  # the single-core figure is the contract; never fail the bench on the extra leg
            out["all_cores"] = {"error": str(exc)}
    return out


def parity_gate(scn, N, grid_cell, sep_nm, device, held_hint=False):
    """correctness gate of every timed run, executed BEFORE the timed region on rank 0.
    (a) reference fixtures: the scripted LOWW episodes of tests/golden/g2_scripted.npz (captured from the imported reference)
        replayed through the batched kernel — flags / done / counters exact, obs and rewards within 1e-5;
    (b) this run's own workload: its first 256 envs x 40 steps against the fp32 CPU oracle (the checker, never the thing
        measured) — flags / done exact, obs and rewards within 1e-5.
    Raises if the gate fails: a number measured on wrong results is not a number."""
    import numpy as np
    import torch
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import model, scenarios
    from oracle import oracle as O
    out = {}
    # (a) golden fixture
    z = np.load(os.path.join(ROOT, "tests", "golden", "g2_scripted.npz"), allow_pickle=False)
    eps = [e for e in json.loads(str(z["episodes"])) if e["scen"] == "LOWW" and e["dt"] == 1.0 and e["shaping"]
           and e["normalize"] and not e["discrete"]]
    env = AtcVecEnv(len(eps), 1, sim_parameters=model.SimParameters(1), scenario=scenarios.LOWW(), device=device,
                    auto_reset=False, spawn="lattice", keep_active=True, grid_cell=grid_cell)
    for b, e in enumerate(eps):
        env.set_state(b, 0, *e["init_state"])
        env.timesteps[b] = e["init_timesteps"]
        env.set_last_action(b, 0, e["init_last_action"])
    steps = np.array([e["steps"] for e in eps])
    starts = np.array([e["start"] for e in eps])
    n_steps, worst_o, worst_r = 0, 0.0, 0.0
    for t in range(int(steps.max())):
        live = t < steps
        rows = np.where(live, starts + t, starts)
        obs, rew, done, info = env.step(z["action"][rows].astype(np.float32).reshape(-1, 1, 3))
        lr = rows[live]
        ok = (np.array_equal(info["flags"].cpu().numpy()[live, 0].astype(np.uint32), z["flags"][lr])
              and np.array_equal(done.cpu().numpy()[live], z["done"][lr])
              and np.array_equal(env.actions_taken.cpu().numpy()[live], z["actions_taken"][lr]))
        if not ok:
            raise RuntimeError("parity gate (a): integer outputs differ from the reference fixture at step %d" % t)
        worst_o = max(worst_o, float(np.abs(obs.cpu().numpy()[live].astype(np.float64) - z["obs"][lr]).max()))
        gw = z["reward"][lr]
        worst_r = max(worst_r, float((np.abs(rew.cpu().numpy()[live] - gw) / np.maximum(1.0, np.abs(gw))).max()))
        n_steps += int(live.sum())
    env.close()
    if worst_o > 1e-5 or worst_r > 1e-5:
        raise RuntimeError("parity gate (a): obs %.2e / reward %.2e beyond 1e-5" % (worst_o, worst_r))
    out["reference_fixture"] = {"fixture": "tests/golden/g2_scripted.npz", "episodes": len(eps), "steps": n_steps,
                                "integer_outputs_exact": True, "max_obs_err": worst_o, "max_rel_reward_err": worst_r}
    out["reference_fixture_timesteps"] = timestep_gate(device, grid_cell)
    out["fp32_oracle"] = oracle_gate(scn, N, grid_cell, sep_nm, device, held_hint)
    return out


def timestep_gate(device, grid_cell, dt=0.1, max_eps=48):
    """(a2) SimParameters.timestep = 0.1 s: episodes of tests/golden/g12_timesteps.npz (captured from the imported reference: sustained
    descents and altitude TIES the reference decides by its own float64 rounding) replayed through the batched kernel — flags /
    done / action counters exact on every step, sampled observations within 1e-5.  The altitude is the reference's float64 and the
    timestep a double since ABI 20; an fp32 accumulator fails this gate."""
    import numpy as np
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import model, scenarios
    z = np.load(os.path.join(ROOT, "tests", "golden", "g12_timesteps.npz"), allow_pickle=False)
    eps = [e for e in json.loads(str(z["episodes"])) if e["scen"] == "LOWW" and e["dt"] == dt and e["shaping"] and e["normalize"]
           and not e["discrete"] and e["steps"] <= 3100][:max_eps]
    flags, done, acts = z["flags"], z["done"], z["actions_taken"]
    idx = np.zeros(len(flags), np.int64)
    idx[z["act_rows"]] = np.arange(len(z["act_rows"]))
    action = z["act_vals"][np.maximum.accumulate(idx)]
    samp = np.full(len(flags), -1, np.int64)
    samp[z["samp_rows"]] = np.arange(len(z["samp_rows"]))
    env = AtcVecEnv(len(eps), 1, sim_parameters=model.SimParameters(dt), scenario=scenarios.LOWW(), device=device, auto_reset=False,
                    spawn="lattice", keep_active=True, grid_cell=grid_cell)
    for b, e in enumerate(eps):
        env.set_state(b, 0, *e["init_state"])
        env.timesteps[b] = e["init_timesteps"]
        env.set_last_action(b, 0, e["init_last_action"])
    steps = np.array([e["steps"] for e in eps])
    starts = np.array([e["start"] for e in eps])
    n_steps, worst, below = 0, 0.0, 0
    for t in range(int(steps.max())):
        live = t < steps
        rows = np.where(live, starts + t, starts)
        obs, rew, dn, info = env.step(action[rows].astype(np.float32).reshape(-1, 1, 3))
        lr = rows[live]
        fl = info["flags"].cpu().numpy()[live, 0].astype(np.uint8)
        if not (np.array_equal(fl, flags[lr]) and np.array_equal(dn.cpu().numpy()[live], done[lr])
                and np.array_equal(env.actions_taken.cpu().numpy()[live], acts[lr])):
            raise RuntimeError("parity gate (a2): integer outputs differ from the reference at timestep %g s, step %d" % (dt, t))
        si = samp[lr]
        if (si >= 0).any():
            o = obs.cpu().numpy()[live][si >= 0].astype(np.float64)
            g = z["obs"][si[si >= 0]].astype(np.float64)
            d = np.abs(o - g)
            d[:, 9] = np.minimum(d[:, 9], np.abs(d[:, 9] - 2.0))   # +-180 deg are one angle (tests/helpers.py: obs_close)
            worst = max(worst, float(d.max()))
        below += int((fl & 1).sum())
        n_steps += int(live.sum())
    env.close()
    if worst > 1e-5:
        raise RuntimeError("parity gate (a2): obs %.2e beyond 1e-5 at timestep %g s" % (worst, dt))
    return {"fixture": "tests/golden/g12_timesteps.npz", "timestep_s": dt, "episodes": len(eps), "steps": n_steps,
            "below_mva_steps": below, "integer_outputs_exact": True, "max_obs_err": worst}


def oracle_gate(scn, N, grid_cell, sep_nm, device, held_hint=False, B=256, T=40):
    """Single launches of a workload (its first B envs, T steps, the launch mode of the timed loop: held steps carry the hint)
    against the fp32 oracle; raises on any difference beyond the parity bar."""
    import numpy as np
    import torch
    from atc_hip.vec_env import AtcVecEnv
    env = AtcVecEnv(B, N, scenario=scn, device=device, auto_reset=True, seed=0, grid_cell=grid_cell, sep_nm=sep_nm)
    orc = _oracle_for(env, scn, N, grid_cell, sep_nm)
    g = torch.Generator(device="cpu").manual_seed(99)
    worst = [0.0, 0.0]
    for t in range(T):
        if t % HOLD == 0:
            a = (torch.rand((B, N, 3), generator=g) * 2 - 1).numpy()
        obs, rew, done, info = env.step(a, held=held_hint and t % HOLD != 0)
        orc.step(a)
        _compare("single steps", t, obs, rew, done, info["flags"], orc, B, N, worst)
    rec = _finish_gate("single steps", env, orc, worst, {"envs": B, "aircraft": N, "steps": T, "held_hint": bool(held_hint),
                                                          "actions_taken_exact": True, "last_action_exact": True})
    env.close()
    return rec


def _oracle_for(env, scn, N, grid_cell, sep_nm, seed=0):
    """The fp32 CPU oracle configured like `env` (checker only)."""
    import numpy as np
    from envs.atc import scenarios
    from oracle import oracle as O
    comp = scenarios.compile_scenario(scn, grid_cell=grid_cell)
    return O.OracleEnv(comp, env.B, N, O.make_params(auto_reset=True, seed=seed, random_entry=bool(env.params.mode & 16),
                                                     sep_nm=sep_nm), np.float32)


def _compare(tag, t, obs, rew, done, flags, orc, B, N, worst):
    import numpy as np
    if not (np.array_equal(flags.cpu().numpy().astype(np.uint16).reshape(B, N), orc.flags)
            and np.array_equal(done.cpu().numpy(), orc.done)):
        raise RuntimeError("parity gate (%s): flags / done differ from the fp32 oracle at step %d" % (tag, t))
    o = obs.cpu().numpy().reshape(B, N, 10)
    worst[0] = max(worst[0], float((np.abs(o - orc.obs) / np.maximum(1.0, np.abs(orc.obs))).max()))
    worst[1] = max(worst[1], float((np.abs(rew.cpu().numpy() - orc.reward) / np.maximum(1.0, np.abs(orc.reward))).max()))


def _finish_gate(tag, env, orc, worst, extra):
    import numpy as np
    if worst[0] > 1e-5 or worst[1] > 1e-5:
        raise RuntimeError("parity gate (%s): obs %.2e / reward %.2e beyond 1e-5" % (tag, worst[0], worst[1]))
    if not (np.array_equal(env.actions_taken.cpu().numpy(), orc.actions_taken)
            and np.array_equal(env.last_act.cpu().numpy(), orc.last_act)
            and np.array_equal(env.ac[:, 0].cpu().numpy(), orc.px) and np.array_equal(env.ac[:, 1].cpu().numpy(), orc.py)):
        raise RuntimeError("parity gate (%s): actions_taken / last_action / positions differ from the fp32 oracle" % tag)
    out = {"flags_done_exact": True, "state_and_counters_exact": True, "max_rel_obs_err": worst[0], "max_rel_reward_err": worst[1]}
    out.update(extra)
    return out


def rollout_gate(scn, N, grid_cell, sep_nm, device, T=HOLD, hold=HOLD, launches=2, B=256):
    """The multi-step entry (atc_rollout_hold, `launches` launches of T steps, one action block per `hold` steps) on the first
    B envs of the workload against the fp32 oracle stepped once per step: raises unless flags / done are exact, obs / rewards
    within 1e-5 and the state, actions_taken and last-action records identical afterwards."""
    import torch
    from atc_hip.vec_env import AtcVecEnv
    env = AtcVecEnv(B, N, scenario=scn, device=device, auto_reset=True, seed=0, grid_cell=grid_cell, sep_nm=sep_nm)
    orc = _oracle_for(env, scn, N, grid_cell, sep_nm)
    g = torch.Generator(device="cpu").manual_seed(77)
    worst = [0.0, 0.0]
    for j in range(launches):
        blocks = torch.rand((T // hold, B, N, 3), generator=g) * 2 - 1
        out = env.rollout(blocks, hold=hold)
        for t in range(T):
            orc.step(blocks[t // hold].numpy())
            _compare("rollout", j * T + t, out["obs"][t], out["reward"][t], out["done"][t], out["flags"][t], orc, B, N, worst)
    rec = _finish_gate("rollout", env, orc, worst, {"entry": "atc_rollout_hold", "envs": B, "aircraft": N, "T": T, "hold": hold,
                                                     "launches": launches})
    env.close()
    return rec


def multi_stream_gate(scn, N, grid_cell, sep_nm, device, held_hint, steps=2 * HOLD, B_half=128):
    """Two independent sub-batches stepped through atc_step_multi on two HIP streams (the launch form of the `two_streams`
    record), each against its own fp32 oracle."""
    import torch
    from atc_hip.vec_env import AtcVecEnv, make_multi_launcher
    dev = torch.device("cuda", device)
    halves = [AtcVecEnv(B_half, N, scenario=scn, device=device, auto_reset=True, seed=7919 * (s + 1), grid_cell=grid_cell,
                        sep_nm=sep_nm) for s in range(2)]
    orcs = [_oracle_for(e, scn, N, grid_cell, sep_nm, seed=7919 * (s + 1)) for s, e in enumerate(halves)]
    qs = [torch.cuda.Stream(device=dev) for _ in range(2)]
    g = torch.Generator(device="cpu").manual_seed(78)
    worst = [0.0, 0.0]
    for t in range(steps):
        if t % HOLD == 0:
            a = torch.rand((2, B_half, N, 3), generator=g) * 2 - 1
            a_dev = a.to(dev)
            torch.cuda.synchronize(dev)
            first = make_multi_launcher(halves, [a_dev[0], a_dev[1]], qs)
            rest = make_multi_launcher(halves, [a_dev[0], a_dev[1]], qs, held=True) if held_hint else first
        (rest if t % HOLD else first)()
        for q in qs:
            q.synchronize()
        for s, (e, o) in enumerate(zip(halves, orcs)):
            o.step(a[s].numpy())
            _compare("two streams", t, e.obs, e.reward, e.done, e.flags, o, B_half, N, worst)
    recs = [_finish_gate("two streams", e, o, worst, {}) for e, o in zip(halves, orcs)]
    for e in halves:
        e.close()
    rec = recs[0]
    rec.update({"entry": "atc_step_multi", "sub_batches": 2, "envs_per_sub_batch": B_half, "aircraft": N, "steps": steps,
                "held_hint": bool(held_hint)})
    return rec


def traffic_entry(B, N, rollout, held_hint, streams=1, grid_cell=None):
    """HBM bytes per launch of this workload from the committed PMC passes (profiles/pmc_traffic.json: separate rocprofv3 --pmc
    runs, FETCH_SIZE / WRITE_SIZE with the gfx950 corrections) — counters cannot be read from inside the process.  Entries are
    keyed by the ABI version of the library they were measured on: a kernel change cannot silently keep an old figure."""
    from atc_hip import layout as L
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath) or streams != 1:
        return None, None
    try:
        for tj in json.load(open(tpath))["workloads"]:
            if (tj.get("abi") == L.ABI_VERSION and tj["envs"] == B and tj["aircraft"] == N and tj["rollout"] == (rollout or 0)
                    and bool(tj.get("held_hint", False)) == bool(held_hint) and tj.get("grid_cell_nm") == grid_cell):
                return tj["hbm_bytes_per_launch"], tj["source"]
    except Exception:
        pass
    return None, None


def side_config(name, B, N, scn, sep_nm, device, held_hint, n_single=2000, n_fused=100, cpu=True, alt_grid=None):
    """One of BASELINE.json's other single-GPU configurations as a side record of the default line: the launch mode of the
    headline loop (one atc_step per step, a new action tensor every HOLD steps, the held-action hint in between) and the same
    envs with HOLD steps fused per launch — each parity-gated on its own first 256 envs before it is timed."""
    import torch
    from atc_hip.vec_env import AtcVecEnv, auto_grid_cell
    dev = torch.device("cuda", device)
    grid_cell = auto_grid_cell(B, N)   # the library's default for this batch; the gates run on the same grid
    gate = {"single_steps": oracle_gate(scn, N, grid_cell, sep_nm, device, held_hint),
            "fused": rollout_gate(scn, N, grid_cell, sep_nm, device)}
    env = AtcVecEnv(B, N, scenario=scn, device=device, auto_reset=True, seed=11, grid_cell=grid_cell, sep_nm=sep_nm)
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    ring = [torch.rand((B, N, 3), generator=g, device=dev, dtype=torch.float32) * 2 - 1 for _ in range(8)]
    first = [env.make_launcher(a) for a in ring]
    rest = [env.make_launcher(a, held=True) for a in ring] if held_hint else first

    def run(n, t0=0):
        for t in range(t0, t0 + n):
            (rest if t % HOLD else first)[(t // HOLD) % len(ring)]()
    run(3000)
    torch.cuda.synchronize(dev)
    blocks = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(n_single)
        e1.record()
        torch.cuda.synchronize(dev)
        blocks.append(e0.elapsed_time(e1) * 1e3 / n_single)
    us = sorted(blocks)[1]
    alt = None
    if alt_grid is not None:
        # A/B asked for by the round-5 review (next #3): the same single-step loop on a coarser lookup grid — fewer bytes from
        # beyond the L2s (the 0.0625 nm table is 11.6 MB, the 0.125 nm one 3.2 MB) against more aircraft in cells a border cuts
        env2 = AtcVecEnv(B, N, scenario=scn, device=device, auto_reset=True, seed=11, grid_cell=alt_grid, sep_nm=sep_nm)
        f2 = [env2.make_launcher(a) for a in ring]
        r2 = [env2.make_launcher(a, held=True) for a in ring] if held_hint else f2
        for t in range(3000):
            (r2 if t % HOLD else f2)[(t // HOLD) % len(ring)]()
        torch.cuda.synchronize(dev)
        b2 = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in range(n_single):
                (r2 if t % HOLD else f2)[(t // HOLD) % len(ring)]()
            e1.record()
            torch.cuda.synchronize(dev)
            b2.append(e0.elapsed_time(e1) * 1e3 / n_single)
        env2.close()
        tr2, src2 = traffic_entry(B, N, 0, held_hint, grid_cell=alt_grid)
        alt = {"grid_cell_nm": alt_grid, "us_per_step": sorted(b2)[1], "traffic": tr2, "traffic_source": src2,
               "traffic_over_algorithmic": (tr2 / (algorithmic_bytes_per_env_step(N) * B)) if tr2 else None}
    ro = {"obs": torch.empty((HOLD, B, N * 10), dtype=torch.float32, device=dev),
          "reward": torch.empty((HOLD, B), dtype=torch.float32, device=dev),
          "done": torch.empty((HOLD, B), dtype=torch.uint8, device=dev),
          "flags": torch.empty((HOLD, B, N), dtype=torch.int16, device=dev)}
    for j in range(30):
        env.rollout(ring[j % len(ring)][None], out=ro, hold=HOLD)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for j in range(n_fused):
        env.rollout(ring[j % len(ring)][None], out=ro, hold=HOLD)
    e1.record()
    torch.cuda.synchronize(dev)
    usf = e0.elapsed_time(e1) * 1e3 / (n_fused * HOLD)
    lds_tab = bool(getattr(env.sector, "has_lds_table", False))
    env.close()
    lds_ab = None
    if lds_tab:
        # one-aircraft envs (ABI 21): the multi-step launch answers the MVA lookup from a table staged in LDS; the same loop
        # without it (the lookup grid's gather) as the A/B side of the record
        env3 = AtcVecEnv(B, N, scenario=scn, device=device, auto_reset=True, seed=11, grid_cell=grid_cell, sep_nm=sep_nm, lds_table=False)
        for j in range(30):
            env3.rollout(ring[j % len(ring)][None], out=ro, hold=HOLD)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for j in range(n_fused):
            env3.rollout(ring[j % len(ring)][None], out=ro, hold=HOLD)
        e1.record()
        torch.cuda.synchronize(dev)
        lds_ab = {"us_per_step_from_the_lookup_grid": e0.elapsed_time(e1) * 1e3 / (n_fused * HOLD),
                  "note": "atc_scenario_attach_lds_table (include/atc_step.h, ABI 21): k_step<1, ..., LDSG>, one workgroup per CU, "
                          "the sector's two-level code table staged in LDS once per launch; results identical (tests/test_lds_table.py)"}
        env3.close()
    b1, bf = algorithmic_bytes_per_env_step(N), algorithmic_bytes_per_env_step(N, HOLD, HOLD)
    tr1, src1 = traffic_entry(B, N, 0, held_hint)
    trf, srcf = traffic_entry(B, N, HOLD, False)
    return {"config": name, "envs": B, "aircraft": N, "sector": type(scn).__name__, "grid_cell_nm": grid_cell, "parity_gate": gate,
            "working_set_bytes": working_set_bytes(B, N), "fits_infinity_cache": working_set_bytes(B, N) <= INFINITY_CACHE_BYTES,
            "cpu_baseline": cpu_side_baseline(N, scn) if cpu else None,
            "single_steps": {"steps": n_single, "us_per_step": us, "env_steps_per_s": B / (us * 1e-6),
                             "algorithmic_bytes_per_env_step": b1, "hbm_frac": b1 * B / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "traffic": tr1, "traffic_source": src1, "timed_blocks_us_per_step": blocks,
                             "traffic_over_algorithmic": (tr1 / (b1 * B)) if tr1 else None, "coarser_grid": alt},
            "fused_rollout": {"entry": "atc_rollout_hold", "T": HOLD, "hold": HOLD, "launches": n_fused, "us_per_step": usf,
                              "env_steps_per_s": B / (usf * 1e-6), "algorithmic_bytes_per_env_step": bf,
                              "hbm_frac": bf * B / (usf * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": trf, "traffic_source": srcf,
                              "lds_table": lds_tab, "lds_table_ab": lds_ab}}


def single_env_protocol(n_steps=100000):
    """The reference's own benchmark (learning/atc-gym-compute-performance.py:10-19): ONE env, 100 000 x step(one fixed
    sampled action), no reset on done, FPS = N / wall time — through the drop-in envs.atc.atc_gym.AtcGym: by its persistent step
    server (atc_serve_*, the default) and, as a side figure, by one kernel launch per step (atc_step_packet, rounds 3-5)."""
    import numpy as np
    from envs.atc import atc_gym
    out = {"protocol": "learning/atc-gym-compute-performance.py:10-19 (1 env x 1 aircraft, one fixed action, no reset)", "steps": n_steps}
    for key, persistent in (("server", True), ("launch", False)):
        env = atc_gym.AtcGym(persistent=persistent)
        env.reset()
        action = np.random.default_rng(0).uniform(-1, 1, 3).astype(np.float32)
        for _ in range(200):
            env.step(action)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            env.step(action)
        dt = time.perf_counter() - t0
        served = bool(env._serving)
        env.close()
        if key == "server":
            out.update({"steps_per_s": n_steps / dt, "us_per_step": dt / n_steps * 1e6, "stepped_by": "persistent step server (atc_serve_step)"
                        if served else "kernel launches (the server was not running)"})
        else:
            out["launch_per_step"] = {"steps_per_s": n_steps / dt, "us_per_step": dt / n_steps * 1e6, "stepped_by": "atc_step_packet"}
    return out


def _sync(dev):
    """torch.cuda.synchronize on a GPU; nothing to drain for the CPU stub flow (--stub-env)."""
    if dev.type == "cuda":
        import torch
        torch.cuda.synchronize(dev)


class _Mark:
    """A point in time on a stream: a HIP event on the GPU, the host clock in the CPU stub flow."""

    def __init__(self, dev):
        self.dev, self.t = dev, None
        if dev.type == "cuda":
            import torch
            self.ev = torch.cuda.Event(enable_timing=True)

    def record(self, stream=None):
        if self.dev.type == "cuda":
            self.ev.record(stream) if stream is not None else self.ev.record()
        else:
            self.t = time.perf_counter()

    def ms_until(self, other):
        return self.ev.elapsed_time(other.ev) if self.dev.type == "cuda" else (other.t - self.t) * 1e3


class StubEnv:
    """TEST-ONLY stand-in for AtcVecEnv in `--stub-env` runs (tests/test_bench_contract.py: the 8-rank flow of this script over
    gloo on a box without a GPU).  It steps nothing: a launch is a counter and 20 us of sleep; its episode statistics are CPU
    tensors whose values encode (rank, env) so that the gathered report can be checked.  Never reachable without the flag."""

    def __init__(self, B, N, rank, torch):
        self.B, self.N, self.launches = B, N, 0
        self.ep_return = (torch.arange(B, dtype=torch.float32) + 1000.0 * rank)
        self.ep_length = torch.arange(B, dtype=torch.int32) + 7 * rank
        self.episodes = torch.ones(B, dtype=torch.int32)

    def make_launcher(self, actions, held=False):
        def launch():
            self.launches += 1
            time.sleep(20e-6)
        return launch

    def close(self):
        pass


def measure_collective(D, stats, dev, force, block=None, reps=100):
    """Cost of the episode-statistics exchange on the initialised group (microseconds, mean of `reps`; wall clock around a
    stream synchronisation and HIP events on the current stream):
      blocking_two_all_gathers_plus_barrier — round 3's report: all_gather_stats(ep_return, ep_length) (two collectives) and
        a barrier, what every timed block of round 3 contained;
      packed_blocking — ONE packed all-gather of a snapshot taken beforehand, issued and waited for at once;
      barrier — the backend's barrier alone;
      block_with / block_without — a 20-step block with the asynchronous packed exchange issued from the side stream behind its
        launches and waited for, against the same block without any exchange: what the exchange WOULD add to a step window
        if it were issued inside it.  It depends on which hardware queue the backend's stream shares (5-15 us stand-alone,
        ~47 us in this process: tools/exchange_overlap.py) — which is why the timed loop issues it after the window."""
    import torch
    out = {"reps": reps, "world_size": torch.distributed.get_world_size()}

    def timed(fn):
        _sync(dev)
        e0, e1 = _Mark(dev), _Mark(dev)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        _sync(dev)
        return {"wall": (time.perf_counter() - t0) / reps * 1e6, "hip_events": e0.ms_until(e1) * 1e3 / reps}

    def old():
        D.all_gather_stats(*stats(), force=force)
        _sync(dev)
        D.barrier(force=force)
    xch = D.StatsExchange(force=force)

    def packed():
        xch.rearm()      # (re-arm the snapshot taken below: the timed loop takes its snapshots outside the window too)
        xch.issue()
        xch.wait()
        _sync(dev)
    xch.snapshot(*stats())
    for f in (old, packed):
        f()
    out["blocking_two_all_gathers_plus_barrier"] = timed(old)
    out["packed_blocking"] = timed(packed)
    out["barrier"] = timed(lambda: D.barrier(force=force))
    if block is not None:
        def with_x():            # queue the block's launches, issue from the side stream, wait, synchronize
            block()
            xch.rearm()
            xch.issue()
            xch.wait()
            _sync(dev)

        def without():
            block()
            _sync(dev)
        with_x()
        without()
        reps_b = 60
        for name, f in (("block_without", without), ("block_with", with_x), ("block_without_2", without), ("block_with_2", with_x)):
            _sync(dev)
            ts = []
            for _ in range(reps_b):   # (the median of per-block times: one scheduling hiccup in sixty must not decide a 5 us difference)
                t0 = time.perf_counter()
                f()
                ts.append((time.perf_counter() - t0) * 1e6)
            out[name] = sorted(ts)[reps_b // 2]
        if os.environ.get("ATC_BENCH_DEBUG_EXCHANGE"):   # developer: where does the host spend a block with the exchange?
            seg = [[], [], [], []]
            for _ in range(40):
                t0 = time.perf_counter()
                block()
                t1 = time.perf_counter()
                xch.rearm()
                xch.issue()
                t2 = time.perf_counter()
                xch.wait()
                t3 = time.perf_counter()
                _sync(dev)
                t4 = time.perf_counter()
                for k, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                    seg[k].append(v * 1e6)
            out["debug_host_us"] = {n: sorted(v)[20] for n, v in zip(("queue_launches", "issue", "wait", "synchronize"), seg)}
        out["block_steps"] = HOLD
        out["exchange_adds_us_per_block"] = min(out["block_with"], out["block_with_2"]) - min(out["block_without"], out["block_without_2"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=ENVS_PER_GPU, help="envs per GPU")
    ap.add_argument("--aircraft", type=int, default=AIRCRAFT)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true", help="developer knob: skip the pre-timing correctness gate")
    ap.add_argument("--no-single-env", action="store_true", help="skip the single-env compute-performance.py protocol")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps; value = the median block")
    ap.add_argument("--grid-cell", default="auto", help="cell size [nm] of the MVA lookup grid (default: by batch size, "
                    "atc_hip.vec_env.auto_grid_cell)")
    ap.add_argument("--no-held-hint", action="store_true", help="do not tell atc_step that launches 2..%d of an action block "
                    "repeat the previous launch's actions (ATC_M_ACTIONS_HELD)" % HOLD)
    ap.add_argument("--action-ring", type=int, default=64, help="number of pre-generated action tensors the loop cycles through "
                    "(one per %d-step block)" % HOLD)
    ap.add_argument("--prewarm", type=int, default=-1, help="developer knob: untimed steps before the warm-up (default: "
                    "6000 / 12000, the steady episode mix); profiling passes use fewer")
    ap.add_argument("--rollout", type=int, default=0, help="fuse this many steps per launch (0 = one launch per step)")
    ap.add_argument("--sep-nm", type=float, default=3.0, help="developer knob: separation minimum (0 disables conflicts)")
    ap.add_argument("--streams", type=int, default=1, help="step the batch as this many independent sub-batches on "
                    "separate HIP streams (no join between steps): launch ramp / tail of one overlaps the others")
    ap.add_argument("--lib", default=None, help="developer knob: path of a libatcstep.so build variant to A/B (default: the "
                    "in-tree build)")
    ap.add_argument("--no-baseline-configs", action="store_true", help="skip the side records of BASELINE.json's other single-GPU "
                    "configurations (65 536 x 1, 8 192 x 16, 4 096 x 64 + noise areas)")
    ap.add_argument("--no-collective", action="store_true", help="N = 1 only: do not create the one-rank RCCL group that "
                    "exercises the multi-GPU path's collective calls on this GPU (untimed)")
    ap.add_argument("--stub-env", action="store_true", help="TEST ONLY (tests/test_bench_contract.py): run this script's multi-rank "
                    "control flow — sharding, timed blocks, the asynchronous report, the line's fields — on CPU tensors with a "
                    "stand-in env that steps nothing (StubEnv); needs ATC_DIST_BACKEND=gloo.  The line says \"data\": \"stub\"")
    ap.add_argument("--graph", action="store_true", help="replay the %d-step action-hold block as one captured HIP graph "
                    "(removes per-launch host overhead; matters for the small launch-bound configs)" % HOLD)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)]
        cmd += sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner through C stdio
    # when a communicator is created): from here on file descriptor 1 is stderr, and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # a fresh checkout has no native pieces yet: local rank 0 compiles them, the other ranks wait for the files
    import __graft_entry__ as entry
    if args.stub_env:
        assert os.environ.get("ATC_DIST_BACKEND") == "gloo", "--stub-env is the CPU test flow: ATC_DIST_BACKEND=gloo"
    elif int(os.environ.get("LOCAL_RANK", "0")) == 0:
        entry.ensure_built()
    else:
        lib_path = os.path.join(ROOT, "atc-reinforcement-learning_amd", "atc_hip", "libatcstep.so")
        t_wait = time.time()
        while not os.path.exists(lib_path) and time.time() - t_wait < 600:
            time.sleep(1.0)

    import torch
    from atc_hip import dist as D
    if args.lib:
        from atc_hip import lib as _binding
        _binding.use_library(args.lib)
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios

    rank, ws, local = D.init()
    assert ws == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (ws, args.gpus)
    collective = {"world_size": ws, "backend": D.backend_name()}
    if ws == 1 and not args.no_collective:
        # One GPU: create a world of ONE rank on the real backend (RCCL) all the same, so that the exact calls of the multi-GPU
        # path — init_process_group("nccl", device_id=...), all_gather_into_tensor / all_reduce on device tensors — run on
        # this box; the collective is issued once, untimed, below.  A failure is recorded, never fatal at N = 1.
        try:
            D.init(force=True)
            collective["backend"] = D.backend_name()
        except Exception as exc:
            collective["error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
    stub = args.stub_env
    assert stub or torch.cuda.is_available(), "bench.py needs the GPU (no CPU fallback)"
    if not stub and local >= torch.cuda.device_count():  # only when ATC_DIST_BACKEND=gloo lets several ranks share the one visible GPU
        assert torch.cuda.device_count() == 1, "LOCAL_RANK %d >= %d visible devices" % (local, torch.cuda.device_count())
        local = 0
    if not stub:
        torch.cuda.set_device(local)
    dev = torch.device("cpu") if stub else torch.device("cuda", local)
    B, N, K, W = args.envs, args.aircraft, args.steps, args.warmup
    from atc_hip.vec_env import auto_grid_cell
    args.grid_cell = auto_grid_cell(B // max(1, args.streams), N) if args.grid_cell == "auto" else float(args.grid_cell)

    scn = scenarios.LOWWDense() if N > 16 else scenarios.LOWW(random_entrypoints=N > 1)
    S = args.streams
    assert S >= 1 and B % S == 0 and not (S > 1 and (args.graph or args.rollout)), "--streams: B % S == 0, no --graph/--rollout"
    if stub:
        assert S == 1 and not args.graph and not args.rollout, "--stub-env: the plain launch loop only"
        subs = [StubEnv(B, N, rank, torch)]
    else:
        subs = [AtcVecEnv(B // S, N, scenario=scn, device=local, auto_reset=True, seed=D.rank_seed(0, rank) + 7919 * s,
                          grid_cell=args.grid_cell, sep_nm=args.sep_nm) for s in range(S)]
    env = subs[0]

    # action ring resident in HBM before timing (Philox, seed 0 + rank)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    n_ring = max(2, min(args.action_ring, (K + W) // HOLD + 1))
    if stub:
        n_ring = 2
    ring = [torch.rand((B, N, 3), generator=g, device=dev, dtype=torch.float32) * 2 - 1 for _ in range(n_ring)]

    launchers = None
    held_launchers = None
    last_block = [None]   # ring index of the previous launch (the held promise is only made for an immediate repeat)
    if S == 1 and not args.graph and not args.rollout:
        # one pre-bound atc_step call per ring tensor (AtcVecEnv.make_launcher): the timed loop then only launches — a few
        # microseconds of host time per step, so the GPU never waits for Python even in a 20-step timed block
        launchers = [env.make_launcher(a) for a in ring]
        if not args.no_held_hint:   # launches 2..HOLD of a held block carry the promise "same actions as the previous launch"
            held_launchers = [env.make_launcher(a, held=True) for a in ring]
    if S > 1:  # sub-batch s owns envs [s B/S, (s+1) B/S) of every ring tensor and its own stream
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        Bs = B // S
        from atc_hip.vec_env import make_multi_launcher
        launchers = [make_multi_launcher(subs, [a[s * Bs:(s + 1) * Bs] for s in range(S)], streams) for a in ring]
        if not args.no_held_hint:
            held_launchers = [make_multi_launcher(subs, [a[s * Bs:(s + 1) * Bs] for s in range(S)], streams, held=True) for a in ring]

    graph = None
    act_buf = None
    if args.graph:
        assert not args.rollout and (K % HOLD == 0) and (W % HOLD == 0), "--graph needs steps/warmup multiples of %d" % HOLD
        act_buf = ring[0].clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):      # warm-up on a side stream as torch requires before capture
            env.step(act_buf)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(HOLD):
                env.step(act_buf)

    def run(n_steps, t_base):
        if launchers is not None:
            for s in range(n_steps):
                j = ((t_base + s) // HOLD) % n_ring
                if held_launchers is not None and last_block[0] == j and (t_base + s) % HOLD:
                    held_launchers[j]()
                else:
                    launchers[j]()   # one foreign call = one step of every sub-batch
                last_block[0] = j
            return
        if graph is not None:
            for s in range(0, n_steps, HOLD):
                act_buf.copy_(ring[((t_base + s) // HOLD) % n_ring], non_blocking=True)
                graph.replay()
            return
        if args.rollout:
            T = args.rollout
            for s in range(0, n_steps, T):
                env.rollout(roll_actions[((t_base + s) // max(T, HOLD)) % len(roll_actions)], out=roll_out, hold=min(T, HOLD))
        else:
            for s in range(n_steps):
                env.step(ring[((t_base + s) // HOLD) % n_ring])

    roll_out = None
    roll_actions = None
    if args.rollout:
        T = args.rollout
        assert (T <= HOLD and HOLD % T == 0) or T % HOLD == 0, "--rollout must divide or be a multiple of %d" % HOLD
        assert K % T == 0 and W % T == 0, "--steps / --warmup must be multiples of --rollout"
        # action blocks of one launch, resident before the timed region: [max(1, T / HOLD), B, N, 3], each held min(T, HOLD) steps
        nb = max(1, T // HOLD)
        roll_actions = [torch.stack([ring[(j * nb + c) % n_ring] for c in range(nb)]) if nb > 1 else ring[j % n_ring][None]
                        for j in range(n_ring)]
        roll_out = {"obs": torch.empty((T, B, N * 10), dtype=torch.float32, device=dev),
                    "reward": torch.empty((T, B), dtype=torch.float32, device=dev),
                    "done": torch.empty((T, B), dtype=torch.uint8, device=dev),
                    "flags": torch.empty((T, B, N), dtype=torch.int16, device=dev)}

    def stats():
        if S == 1:
            return env.ep_return, env.ep_length
        return torch.cat([e.ep_return for e in subs]), torch.cat([e.ep_length for e in subs])

    gate = None
    if rank == 0 and not args.no_parity_gate and not stub:
        gate = parity_gate(scn, N, args.grid_cell, args.sep_nm, local, held_hint=held_launchers is not None)   # raises if results are wrong

    # Untimed: bring every rank's GPU to its working clocks and the envs into their steady episode mix before the W warm-up
    # steps the caller asked for (the driver uses a handful; the first launches after start-up are not representative).
    PREWARM = args.prewarm if args.prewarm >= 0 else (0 if stub else 6000 if N * B >= 1 << 18 else 12000)
    run(PREWARM - PREWARM % max(1, args.rollout, HOLD if args.graph else 1), 0)
    run(W, 0)
    _sync(dev)
    forced = ws == 1 and bool(collective.get("backend")) and "error" not in collective
    warm = D.all_gather_stats(*stats(), force=True)  # untimed: creates the RCCL communicator / channels
    _sync(dev)
    if forced:
        try:   # the one-rank group's collectives really ran: device tensors in, the same values out
            collective.update({"forced_single_rank": True, "gathered_shape": list(warm[0].shape), "device_tensors": bool(warm[0].is_cuda),
                               "matches_local": bool(torch.equal(warm[0][0], stats()[0])),
                               "all_reduce_max": D.max_over_ranks(1.5, dev, force=True), "all_reduce_sum": D.sum_over_ranks(2.5, dev, force=True)})
        except Exception as exc:
            collective["error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
    if (ws > 1 or forced) and "error" not in collective:
        # What the exchange costs on THIS group (untimed; the real backend: RCCL on a GPU box, one rank here or eight on the
        # driver's node), so that the line says what a blocking report would have added to a timed block and what the
        # asynchronous one does add (round-3 review, next #1).
        try:
            collective["us"] = measure_collective(D, stats, dev, force=forced,
                                                  block=(lambda: run(HOLD, 0)) if (launchers is not None or args.rollout in (0, HOLD)) and not args.graph else None)
        except Exception as exc:
            collective["us_error"] = "%s: %s" % (type(exc).__name__, str(exc)[:300])
        last_block[0] = None
    if ws == 1:
        D.shutdown()   # the one-rank group has served its purpose: no communicator (proxy thread, streams) during the timed region
        _sync(dev)
    # HIP events on the stream(s) the kernels are launched on (torch's current stream, or one per sub-batch)
    qs = streams if S > 1 else [None if stub else torch.cuda.current_stream(dev)]
    # The path's only exchange: ONE packed all-gather of the per-env episode statistics per rollout, asynchronous
    # (atc_hip.dist.StatsExchange), and OUTSIDE the step window: a timed block queues its K step launches and synchronises — the
    # window closes there, on every rank by its own clock — and only then snapshots its statistics and issues the collective from a
    # side stream, where it runs beside the closing barrier and the next block's start-up; it is waited for after the next
    # opening synchronisation.  (Round 4 first put the exchange INSIDE the window, overlapped with the step kernels: whether it
    # overlaps depends on which hardware queue the backend's stream lands on — +5..15 us per 20-step block in a stand-alone
    # process, +47 us in this one, tools/exchange_overlap.py — so the window does not depend on that luck; the measured cost of
    # the exchange is in `config.collective.us` and `config.exchange`, and `value_incl_exchange` charges it in full.)
    xch = D.StatsExchange()
    blocks = []   # per timed block, max over ranks: (step-window seconds, HIP-event ms, seconds incl. the closing barrier)
    local_windows = []   # per timed block, THIS rank's step-window seconds (gathered after the loop: a straggler must be visible)
    gathered = None
    for rep in range(max(1, args.repeats)):
        D.barrier()
        _sync(dev)
        if rep:
            gathered = xch.wait()        # the previous rollout's statistics of every rank: complete since the synchronisation above
        ev0 = [_Mark(dev) for _ in qs]
        ev1 = [_Mark(dev) for _ in qs]
        t0 = time.perf_counter()
        for e, q in zip(ev0, qs):
            e.record(q)
        run(K, W + rep * K)
        for e, q in zip(ev1, qs):
            e.record(q)
        _sync(dev)                       # every local step has completed (sub-batch streams joined)
        t1 = time.perf_counter()         # <- the step window closes here, on every rank by its own clock; MAX over ranks below
        xch.snapshot(*stats())           # this rollout's report: one packed collective, issued asynchronously from a side stream
        xch.issue()
        D.barrier()
        t2 = time.perf_counter()
        local_windows.append(t1 - t0)
        blocks.append((D.max_over_ranks(t1 - t0, dev), D.max_over_ranks(max(a.ms_until(b) for a, b in zip(ev0, ev1)), dev),
                       D.max_over_ranks(t2 - t0, dev)))
    _sync(dev)
    returns, lengths = xch.wait()        # the last rollout's report
    _sync(dev)
    rank_seeds = D.all_gather_stats(torch.tensor([D.rank_seed(0, rank) & 0x7fffffffffffffff], dtype=torch.int64, device=dev))[0]
    order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
    elapsed, kernel_ms_total, elapsed_with_barrier = blocks[order[len(order) // 2]]   # the median block
    # every rank's own step window of that block (the blocks list holds maxima over ranks: the same on every rank, so all ranks
    # pick the same block)
    rank_windows = D.all_gather_stats(torch.tensor([local_windows[order[len(order) // 2]]], dtype=torch.float64, device=dev))[0]
    rank_ms_per_step = [float(v) / K * 1e3 for v in rank_windows.reshape(-1).tolist()]
    T = args.rollout or 1
    n_launches = K // T
    launch_ms = kernel_ms_total / n_launches   # average launch duration (HIP events) of the median block

    episodes = D.sum_over_ranks(float(sum(e.episodes.sum().item() for e in subs)) - B, dev)
    value = ws * B * K / elapsed
    # the exchange's measured cost on this run's group (one packed all-gather issued and waited for at once), None if unmeasured
    exchange_us = (collective.get("us") or {}).get("packed_blocking", {}).get("wall")
    bytes_launch = algorithmic_bytes_per_env_step(N, T, min(T, HOLD) if args.rollout else 1) * (B // S) * T
    # S > 1: launch_ms is the wall duration of ONE sub-batch launch while S - 1 others are in flight
    achieved = S * bytes_launch / (launch_ms * 1e-3) / 1e9
    traffic, traffic_src = traffic_entry(B, N, args.rollout, held_launchers is not None, S)

    if rank == 0:
        line = {
            "metric": "env-steps/sec aggregate @16 aircraft/env, 64k envs" if (N == 16 and B == 65536) else "env-steps/sec",
            "value": value, "unit": "env-steps/s", "n_gpus": ws, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "stub" if stub else "synthetic",
            "config": {"workload": "%d envs x %d aircraft per GPU (%d envs total), sector %s, dt=1 s, reward shaping + "
                                   "normalisation on, continuous actions U(-1,1) re-sampled every %d steps, auto-reset, "
                                   "O(N^2) separation scan, MVA lookup grid %g nm" % (B, N, B * ws, type(scn).__name__, HOLD, args.grid_cell),
                       "envs_per_gpu": B, "aircraft_per_env": N, "launch": "rollout T=%d" % args.rollout if args.rollout
                       else ("one atc_step launch per step, %d-step blocks replayed as a captured HIP graph" % HOLD
                             if args.graph else ("%d sub-batches of %d envs on %d HIP streams, one launch per sub-batch per step "
                                                "(atc_step_multi), no join between steps" % (S, B // S, S) if S > 1 else "one atc_step launch per step")), "parallelism": "env-sharded x%d, no step-path collective, "
                       "1 all-gather of episode returns per rollout" % ws,
                       "episodes_finished": int(episodes), "positions": "32-bit fixed point (2^-25 nm grid)",
                       "timed_blocks_ms_per_step": [b[0] / K * 1e3 for b in blocks],
                       "timing": "median of %d timed blocks of %d steps; a block = barrier + synchronize | t0 | queue the %d step "
                                 "launches, synchronize | t1 | snapshot the episode statistics and issue the ONE packed asynchronous "
                                 "all-gather from a side stream, barrier | t2 (the all-gather is waited for after the next block's "
                                 "opening synchronisation).  `value` = ranks x envs x steps / MAX over ranks of (t1 - t0): the step "
                                 "window, which holds no collective — the path has none on its step path; `rank_ms_per_step` "
                                 "lists every rank's own window.  `value_between_barriers` = the same work / MAX over ranks of "
                                 "(t2 - t0): the aggregate between two barriers as SURVEY 8e defines it — it charges the exchange's "
                                 "host-side issue and the closing barrier to the steps; `value_incl_exchange` charges the exchange's "
                                 "measured blocking cost (`exchange.us_blocking`, on THIS group) in full" % (len(blocks), K, n_launches),
                       "rank_ms_per_step": rank_ms_per_step,
                       "value_between_barriers": ws * B * K / elapsed_with_barrier,
                       "ms_per_step_incl_closing_barrier": elapsed_with_barrier / K * 1e3,
                       "exchange": {"collectives_per_report": 1, "issued": xch.collectives, "reports": len(blocks),
                                    "async": True, "in_step_window": False, "us_blocking": exchange_us,
                                    "payload": "ep_return + ep_length packed as one [envs, 2] 32-bit tensor, "
                                    "%d bytes per rank" % (8 * B)},
                       "value_incl_exchange": ws * B * K / (elapsed + (exchange_us or 0.0) * 1e-6),
                       "actions_held_hint": ("launches 2..%d of every %d-step action block carry ATC_M_ACTIONS_HELD (the caller's "
                                             "promise that the block is repeated; results identical, last_action record skipped)"
                                             % (HOLD, HOLD)) if held_launchers is not None else None,
                       "prewarm_steps": PREWARM, "parity_gate": gate, "gathered_returns_shape": list(returns.shape),
                       "collective": collective, "collective_backend": collective.get("backend"),
                       "rank_seeds": [int(v) for v in rank_seeds.reshape(-1).tolist()]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_step<%d, false, %s, %s>" % (1 << max(0, (N - 1).bit_length()), "false" if args.rollout else "true",
                                                                  "true" if (N & (N - 1)) == 0 and (B // S * N) % 256 == 0 else "false"),
                         "avg_launch_ms": launch_ms, "algorithmic_bytes_per_launch": bytes_launch,
                         "concurrent_launches": S,
                         "working_set_bytes": working_set_bytes(B, N, T), "fits_infinity_cache": working_set_bytes(B, N, T) <= INFINITY_CACHE_BYTES,
                         "frac_of": "algorithmic bytes per launch / launch duration / 8 TB/s (HBM3E spec).  The step's working set "
                                    "(state + one action tensor + outputs) is re-touched every launch: where it fits the 256 MiB "
                                    "Infinity Cache (`fits_infinity_cache`) part of those bytes is served by the MALL, not DRAM — "
                                    "the rate is a rate of algorithmic bytes, not proven DRAM traffic; `config.beyond_l3` is the same "
                                    "kernel on a working set 4x larger than the cache",
                         "note": "HIP events on the launch stream around the %d timed launches (includes inter-launch "
                                 "gaps)" % n_launches},
        }
        if stub:   # what the CPU test checks the report against: StubEnv encodes (rank, env) in its statistics
            line["config"]["stub_report"] = {"first_return_of_each_rank": [float(v) for v in returns[:, 0].tolist()],
                                             "first_length_of_each_rank": [int(v) for v in lengths[:, 0].tolist()],
                                             "launches_rank0": subs[0].launches}
        if ws == 1 and not args.no_single_env and not args.rollout and S == 1 and graph is None:
            # the same envs with 20 steps fused per launch (atc_rollout_hold, the action held like in the timed loop): a
            # side record, not `value` — the headline stays one launch per step, what env.step() costs
            Tf = HOLD
            fused_gate = None if args.no_parity_gate else rollout_gate(scn, N, args.grid_cell, args.sep_nm, local)   # raises if wrong
            ro = {"obs": torch.empty((Tf, B, N * 10), dtype=torch.float32, device=dev),
                  "reward": torch.empty((Tf, B), dtype=torch.float32, device=dev),
                  "done": torch.empty((Tf, B), dtype=torch.uint8, device=dev),
                  "flags": torch.empty((Tf, B, N), dtype=torch.int16, device=dev)}
            for j in range(50):
                env.rollout(ring[j % n_ring][None], out=ro, hold=Tf)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_l = 100
            e0.record()
            for j in range(n_l):
                env.rollout(ring[j % n_ring][None], out=ro, hold=Tf)
            e1.record()
            torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) * 1e3 / (n_l * Tf)
            fb = algorithmic_bytes_per_env_step(N, Tf, Tf)
            line["config"]["fused_rollout"] = {"entry": "atc_rollout_hold", "T": Tf, "hold": Tf, "launches": n_l,
                                               "us_per_step": us, "env_steps_per_s": B / (us * 1e-6),
                                               "algorithmic_bytes_per_env_step": fb,
                                               "hbm_frac": fb * B / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                               "traffic": traffic_entry(B, N, Tf, False)[0], "parity_gate": fused_gate}
            del ro
            torch.cuda.synchronize(dev)
            so = store_only_reference(B, N, Tf)
            if so is not None:
                so["fused_launch_us"] = us * Tf
                so["fused_over_store_pattern"] = us * Tf / so["store_pattern_us_per_launch"]
                if "store_pattern_400_fma_1_gather_us_per_launch" in so:
                    # the launch against a kernel that writes the same bytes in the same pattern AND carries the step's instruction
                    # count and its one gather: what is left above 1.0 is the step's waits, scalar work and occupancy (6 of 8 waves)
                    so["fused_over_pattern_with_arithmetic"] = us * Tf / so["store_pattern_400_fma_1_gather_us_per_launch"]
                arr = so.get("arrangements_us_per_launch")
                if arr:
                    best = min(arr, key=arr.get)
                    so["best_pattern"] = {"name": best, "us_per_launch": arr[best],
                                          "hbm_frac_if_the_launch_ran_at_it": fb * B * Tf / (arr[best] * 1e-6) / 1e9 / HBM_PEAK_GBS}
                line["config"]["fused_rollout"]["store_only_reference"] = so
            if held_launchers is not None:
                # the same loop without the held-action promise (every launch reads the last-action record): a side record
                def run_plain(n):
                    for t in range(n):
                        launchers[(t // HOLD) % n_ring]()
                run_plain(400)
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run_plain(2000)
                e1.record()
                torch.cuda.synchronize(dev)
                usp = e0.elapsed_time(e1) * 1e3 / 2000
                line["config"]["without_held_hint"] = {"steps": 2000, "us_per_step": usp, "env_steps_per_s": B / (usp * 1e-6),
                                                       "hbm_frac": algorithmic_bytes_per_env_step(N) * B / (usp * 1e-6) / 1e9 / HBM_PEAK_GBS}
                last_block[0] = None
            if B % 2 == 0 and B >= 8192:
                # the same batch as two independent sub-batches of B / 2 envs on two HIP streams (atc_step_multi: one foreign call
                # per step, no join between steps, so one sub-batch's launch floor overlaps the other's body): a side record too —
                # `value` / `roofline` stay the single in-order launch whose rocprofv3 kernel duration can be compared
                from atc_hip.vec_env import make_multi_launcher
                ms_gate = None if args.no_parity_gate else multi_stream_gate(scn, N, args.grid_cell, args.sep_nm, local,
                                                                             not args.no_held_hint)   # raises if wrong
                halves = [AtcVecEnv(B // 2, N, scenario=scn, device=local, auto_reset=True, seed=7919 * (s + 1),
                                    grid_cell=args.grid_cell, sep_nm=args.sep_nm) for s in range(2)]
                qs2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
                n_r = min(n_ring, 8)
                first = [make_multi_launcher(halves, [a[s * (B // 2):(s + 1) * (B // 2)] for s in range(2)], qs2) for a in ring[:n_r]]
                rest = first if args.no_held_hint else [
                    make_multi_launcher(halves, [a[s * (B // 2):(s + 1) * (B // 2)] for s in range(2)], qs2, held=True)
                    for a in ring[:n_r]]

                def run2(n):
                    for t in range(n):
                        (rest if t % HOLD else first)[(t // HOLD) % n_r]()
                run2(2000)
                for q in qs2:
                    q.synchronize()
                n2 = 2000
                t0 = time.perf_counter()
                run2(n2)
                for q in qs2:
                    q.synchronize()
                us2 = (time.perf_counter() - t0) / n2 * 1e6
                line["config"]["two_streams"] = {"entry": "atc_step_multi", "sub_batches": 2, "envs_per_sub_batch": B // 2, "steps": n2,
                                                 "us_per_step": us2, "env_steps_per_s": B / (us2 * 1e-6),
                                                 "hbm_frac": algorithmic_bytes_per_env_step(N) * B / (us2 * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                                 "parity_gate": ms_gate}
                for e in halves:
                    e.close()
        if (ws == 1 and not args.no_baseline_configs and N == AIRCRAFT and B == ENVS_PER_GPU and not args.rollout and S == 1
                and graph is None):
            # The headline kernel on a working set beyond the 256 MiB Infinity Cache (4 x the envs: 366 MB per launch): what
            # the same code does when the bytes provably come from and go to DRAM (round-3 review, next #7).  Same kernel
            # instantiation and the same 0.25 nm grid as the gated headline; a side record, never `value`.
            Bl = 4 * ENVS_PER_GPU
            big = AtcVecEnv(Bl, N, scenario=scn, device=local, auto_reset=True, seed=5, grid_cell=args.grid_cell, sep_nm=args.sep_nm)
            gb = torch.Generator(device=dev)
            gb.manual_seed(99)
            ring_b = [torch.rand((Bl, N, 3), generator=gb, device=dev, dtype=torch.float32) * 2 - 1 for _ in range(3)]
            fb = [big.make_launcher(a) for a in ring_b]
            rb = [big.make_launcher(a, held=True) for a in ring_b] if not args.no_held_hint else fb

            def run_big(n):
                for t in range(n):
                    (rb if t % HOLD else fb)[(t // HOLD) % 3]()
            run_big(1500)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run_big(600)
            e1.record()
            torch.cuda.synchronize(dev)
            usb = e0.elapsed_time(e1) * 1e3 / 600
            line["config"]["beyond_l3"] = {"envs": Bl, "aircraft": N, "grid_cell_nm": args.grid_cell, "steps": 600, "us_per_step": usb,
                                           "env_steps_per_s": Bl / (usb * 1e-6), "working_set_bytes": working_set_bytes(Bl, N),
                                           "fits_infinity_cache": working_set_bytes(Bl, N) <= INFINITY_CACHE_BYTES,
                                           "hbm_frac": algorithmic_bytes_per_env_step(N) * Bl / (usb * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                           "traffic": traffic_entry(Bl, N, 0, not args.no_held_hint)[0]}
            big.close()
            del big, ring_b, fb, rb
        if (ws == 1 and not args.no_baseline_configs and not args.no_parity_gate and N == AIRCRAFT and B == ENVS_PER_GPU
                and not args.rollout and S == 1 and graph is None):
            # BASELINE.json's other single-GPU configurations, each gated on its own first 256 envs (side records: `value`
            # stays the headline configuration)
            hh = not args.no_held_hint
            line["config"]["baseline_configs"] = [
                side_config("C2: 65 536 envs x 1 aircraft (kinematics + MVA only)", 65536, 1, scenarios.LOWW(),
                            args.sep_nm, local, hh, cpu=not args.no_cpu_baseline, alt_grid=0.125),
                side_config("C3: 8 192 envs x 16 aircraft", 8192, 16, scenarios.LOWW(random_entrypoints=True),
                            args.sep_nm, local, hh, cpu=not args.no_cpu_baseline),
                side_config("C4: 4 096 envs x 64 aircraft, multi-polygon MVA + noise-abatement areas", 4096, 64,
                            scenarios.LOWWDense(), args.sep_nm, local, hh, cpu=not args.no_cpu_baseline)]
        if ws == 1 and not args.no_single_env:
            se = single_env_protocol()
            if not args.no_cpu_baseline:
                # the compiled-CPU counterpart on 1 host core, same protocol: a one-env step is a launch-latency path on a
                # GPU — the number that says by how much a CPU step beats it belongs beside it
                co = single_env_cpu_oracle()
                se["cpu_oracle_steps_per_s"] = co["f32"]
                se["cpu_oracle_f64_steps_per_s"] = co["f64"]
                se["cpu_oracle_note"] = ("oracle/ (C restatement of the reference step) stepped 1 env x 1 aircraft from Python, "
                                         "1 core; the GPU-backed drop-in is %.2fx of it: one env's step is two crossings of the host link and one wavefront's serial chain"
                                         % (se["steps_per_s"] / co["f32"]))
            line["config"]["single_env"] = se
        if ws == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N)
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    for e in subs:
        e.close()
    D.shutdown()


if __name__ == "__main__":
    main()
