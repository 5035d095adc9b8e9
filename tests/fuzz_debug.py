"""Debugging aid for a failing case of test_fuzz_parity.py (not collected by pytest):
    python tests/fuzz_debug.py <seed>
replays the case step by step and prints the first deviation of every kind with its context."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "atc-reinforcement-learning_amd"), os.path.dirname(os.path.abspath(__file__))]
import numpy as np
import test_fuzz_parity as T
from atc_hip.vec_env import AtcVecEnv
from envs.atc import model
from oracle import oracle as O

seed = int(sys.argv[1])
scn, comp, kw = T._case(seed)
print(type(scn).__name__, kw)
B, N, full = kw["B"], kw["N"], kw["full"]
sp = model.SimParameters(kw["dt"], discrete_action_space=kw["discrete"], reward_shaping=kw["shaping"],
                         normalize_state=kw["normalize"])
env = AtcVecEnv(B, N, sim_parameters=sp, scenario=scn, auto_reset=True, spawn=kw["spawn"], seed=seed, grid_cell=kw["grid_cell"],
                want_raw_obs=True, want_ac_reward=True, want_min_sep=True, want_term_obs=True,
                timestep_limit=kw["timestep_limit"], sep_nm=kw["sep_nm"])
p = O.make_params(dt=kw["dt"], discrete=kw["discrete"], auto_reset=True, random_entry=(kw["spawn"] == "random"), seed=seed,
                  timestep_limit=kw["timestep_limit"], shaping=kw["shaping"], normalize=kw["normalize"], sep_nm=kw["sep_nm"])
orc = O.OracleEnv(comp, B, N, p, np.float32)
half_range = 0.5 * comp.norm_max.astype(np.float32)
rng = np.random.default_rng(seed)
act = None
for t in range(kw["steps"]):
    if t % kw["hold"] == 0 or act is None:
        if kw["discrete"]:
            act = np.floor(rng.uniform(0, 1, (B, N, 3)) * np.array([20, 380, 360])).astype(np.float32)
        else:
            act = rng.uniform(-1.05, 1.05, (B, N, 3)).astype(np.float32)
    o, r, d, info = env.step(act)
    orc.step(act)
    fl = info["flags"].cpu().numpy().astype(np.uint32)
    bad = False
    if not np.array_equal(fl, orc.flags):
        b, k = np.argwhere(fl != orc.flags)[0]
        print("t", t, "FLAGS differ at env", b, "aircraft", k, "hip", fl[b, k], "orc", orc.flags[b, k])
        bad = True
    else:
        on = o.cpu().numpy().reshape(B, N, 10)
        scale = np.maximum(1.0, np.abs(orc.obs))
        if not kw["normalize"]:
            scale = np.maximum(scale, half_range)
        viol = np.abs(on - orc.obs) / scale
        rr = r.cpu().numpy()
        rv = np.abs(rr - orc.reward) / (np.maximum(1.0, np.abs(orc.reward)) * max(1, N // 4))
        if viol.max() > 1e-5:
            b, k, c = np.unravel_index(viol.argmax(), viol.shape)
            print("t", t, "OBS deviation", viol.max(), "env", b, "aircraft", k, "component", c)
            bad = True
        elif rv.max() > 1e-5:
            b = int(rv.argmax())
            k = int(np.abs(info["aircraft_reward"].cpu().numpy()[b] - orc.ac_reward[b]).argmax())
            print("t", t, "REWARD deviation", rv.max(), "env", b, "hip", rr[b], "orc", orc.reward[b], "worst aircraft", k)
            print(" per-aircraft hip", info["aircraft_reward"].cpu().numpy()[b], "\n per-aircraft orc", orc.ac_reward[b])
            bad = True
    if bad:
        raw = info["original_state"].cpu().numpy().reshape(B, N, 10)
        print(" hip obs", o.cpu().numpy().reshape(B, N, 10)[b, k], "\n orc obs", orc.obs[b, k])
        print(" hip raw", raw[b, k], "\n orc raw", orc.raw_obs[b, k])
        print(" flags hip/orc", fl[b, k], orc.flags[b, k], "done", int(d[b]), int(orc.done[b]), "t_env", orc.timesteps[b])
        print(" state hip", env.get_state(int(b), int(k)), "\n state orc", [orc.x.reshape(B, N)[b, k], orc.y.reshape(B, N)[b, k],
              orc.h.reshape(B, N)[b, k]])
        break
print("done")
