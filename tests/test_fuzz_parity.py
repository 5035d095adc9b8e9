"""Randomised differential test of the HIP step path against the fp32 oracle: random batch shapes, aircraft counts,
sectors, lookup-grid cells, modes (dt, discrete, shaping, normalisation, spawn, timeout limit, separation minimum),
kernel variant (fast / full) and launch form (single steps / fused rollout / fused rollout with held action blocks).  ATC_FUZZ_CASES sets the number of cases
(default: a short pass), ATC_FUZZ_SEED the first seed; every case is reproducible from its seed
(tests/fuzz_debug.py <seed> replays one and prints the first deviation with its context).

Since ABI 11 the fp32 spec (include/atc_step.h: fixed-point position grid, shared heading kinematics) makes the aircraft
state of the HIP path BIT-IDENTICAL to the fp32 oracle's, so flags / done / counters agree structurally, not statistically:
the round-1 knife-edge case (seed 52960: an aircraft within one fp32 ulp of a vertical MVA border binned on different
sides because the two float64 positions differed by 4e-8 nm) cannot occur any more.  The default run does 200 cases; a
sweep of 10 000 cases is recorded in profiles/ (see profiles/README.md)."""
import os

import numpy as np
import pytest

import helpers as H
from test_hip_parity import _run_vs_oracle

pytestmark = pytest.mark.gpu


def _case(seed):
    from envs.atc import scenarios
    rng = np.random.default_rng(seed)
    N = int(rng.choice([1, 1, 2, 3, 5, 8, 9, 15, 16, 16, 17, 24, 32, 33, 48, 63, 64]))
    kind = rng.choice(["LOWW", "LOWW_random", "Simple", "Dense"])
    if N > 54:
        kind = "Dense"      # the other sectors have fewer conflict-free spawn slots
    scn = scenarios.LOWWDense() if kind == "Dense" else H.make_scenario(str(kind))
    grid_cell = [None, 0.25, 0.5, 1.0][int(rng.integers(4))]
    comp = scenarios.compile_scenario(scn, grid_cell=grid_cell)
    rollout = int(rng.choice([0, 0, 0, 4]))
    kw = dict(B=int(rng.integers(1, 400)), N=N, steps=int(rng.choice([60, 120, 200])), seed=int(seed),
              dt=float(rng.choice([1.0, 1.0, 2.0, 5.0])), discrete=bool(rng.integers(2)),
              spawn=str(rng.choice(["lattice", "random"])) if comp.n_entry > 1 else "lattice",
              hold=int(rng.choice([1, 7, 20])), grid_cell=grid_cell, use_rollout=rollout,
              timestep_limit=int(rng.choice([6000, 6000, 40])), full=bool(rng.integers(2)),
              shaping=bool(rng.integers(4) > 0), normalize=bool(rng.integers(4) > 0),
              sep_nm=float(rng.choice([3.0, 3.0, 0.0, 5.0])), keep_active=bool(rng.integers(5) == 0))
    kw["held_hint"] = bool(rng.integers(2))   # drawn last: the cases of earlier sweeps keep their configurations
    # atc_rollout_hold with hold > 1 (drawn after everything else for the same reason): a quarter of the cases
    rh = int(rng.choice([1, 1, 1, 1, 1, 1, 4, 20]))
    if rh > 1:
        rollout = rh * int(rng.choice([1, 2, 5]))
        kw.update(use_rollout=rollout, rollout_hold=rh, hold=rh * int(rng.choice([1, 2])))
    if rollout:
        kw["steps"] = max(rollout, (kw["steps"] // rollout) * rollout)
    # round 4 (drawn after everything else: earlier sweeps keep their configurations): a third of the cases replace the
    # drawn cell size by the SHIPPED defaults — 0.125 nm explicitly, or "auto" (atc_hip.vec_env.auto_grid_cell: 0.125 nm for
    # every batch this sweep draws), whose compiled sector must be the one the env builds for itself
    pick = int(rng.integers(6))
    if pick < 2:
        kw["grid_cell"] = 0.125 if pick == 0 else "auto"
        comp = scenarios.compile_scenario(scn, grid_cell=0.125)
    elif pick == 2 and kind != "Dense":   # 0.0625 nm: what `auto` picks from 4 096 aircraft slots up (bigger than this sweep's batches)
        kw["grid_cell"] = 0.0625
        comp = scenarios.compile_scenario(scn, grid_cell=0.125)
    # round 4, drawn last again: a quarter of the cases whose aircraft count is a power of two get a batch that is a whole number of
    # workgroups — the launches then run the all-valid kernel instantiations (csrc/atc_step.hip: make_ids<W, ALLV>), the others
    # the general ones
    if int(rng.integers(4)) == 0 and (N & (N - 1)) == 0:
        per = max(1, 256 // N)
        kw["B"] = per * max(1, kw["B"] // per // (4 if N == 1 else 1))
    # round 5, drawn last: a tenth of the cases carry actions OUTSIDE the action space (U(-4, 4) and beyond) — the reference
    # enforces none (atc_gym.py:128-141) and never validates or wraps a heading (model.py:104-120)
    if int(rng.integers(10)) == 0:
        kw["wild"] = float(rng.choice([0.05, 0.3, 1.0]))
    # round 6, drawn last: a third of the cases step at a timestep that is NOT a small dyadic multiple (SimParameters.timestep is
    # any float, model.py:132-145): the class of inputs the sweeps of rounds 2-5 never drew (the fp32 altitude accumulator was
    # only exact at 1 / 2 / 5 s; tests/golden/g12 pins the reference at these)
    if int(rng.integers(3)) == 0:
        kw["dt"] = float(rng.choice([0.05, 0.1, 0.15, 0.3, 0.7, 1.3, 3.7, 0.37, 2.1]))
    # round 6 (ABI 21), drawn last: one-aircraft envs stepped by multi-step launches get, every second time, a batch of whole
    # 256-env workgroups — with a lookup grid and no noise-abatement areas that launch answers the MVA lookup from the sector's
    # LDS-resident table (k_step<1, ..., LDSG>; tests/test_lds_table.py)
    if N == 1 and kw["use_rollout"] and int(rng.integers(2)) == 0:
        kw["B"] = 256 * int(rng.integers(1, 3))
    return scn, comp, kw


@pytest.mark.parametrize("seed", [int(os.environ.get("ATC_FUZZ_SEED", "1000")) + i
                                  for i in range(int(os.environ.get("ATC_FUZZ_CASES", "200")))])
def test_random_configuration_matches_oracle(seed):
    scn, comp, kw = _case(seed)
    print("fuzz case", seed, type(scn).__name__, kw)
    _run_vs_oracle(scn, comp, **kw)
