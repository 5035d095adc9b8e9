"""AtcSBVecEnv — stable-baselines-shaped VecEnv over AtcVecEnv.

Counterpart of `SubprocVecEnv([make_env] * n)` + `Monitor` in the reference's trainer
(learning/atc-gym-stable-baselines.py:69-80): same method names and return conventions (numpy arrays, list of info dicts,
auto-reset with the reset observation returned and the terminal one under info["terminal_observation"],
Monitor's info["episode"] = {"r", "l", "t"} for finished episodes, `get_attr`/`set_attr`/`env_method`), but all envs are
stepped by ONE kernel launch and the Monitor statistics are accumulated on the device; only the per-step arrays cross
PCIe (B x (10 N + 2) floats).  For throughput without host copies use AtcVecEnv directly (device tensors)."""
import importlib
import time

import numpy as np

from . import layout as L
from .vec_env import AtcVecEnv


def _vecenv_base():
    """The VecEnv abstract base of whichever stable-baselines is importable — the reference pins stable-baselines 2.8.0
    (requirements.txt), whose BaseRLModel wraps anything that is NOT a `VecEnv` instance in DummyVecEnv([lambda: env]); its
    trainer hands the vector env straight to PPO2 (learning/atc-gym-stable-baselines.py:76-90).  Only that library is looked for: the
    adapter has been exercised against its VecEnv contract (tests/sb_shim), not against stable-baselines3's (gymnasium spaces,
    reset_infos / _seeds bookkeeping).  It is not installed in this image: then the adapter is a plain class with the same surface."""
    for mod in ("stable_baselines.common.vec_env",):
        try:
            return importlib.import_module(mod).VecEnv
        except Exception:   # noqa: BLE001 — not installed, or an installation that does not import here
            continue
    return object


_Base = _vecenv_base()


class AtcSBVecEnv(_Base):
    def __init__(self, num_envs, num_aircraft=1, sim_parameters=None, scenario=None, device=0, seed=0, sparse_infos=None,
                 host_mapped=None, **kw):
        try:   # gym's own space classes where gym is there (stable-baselines' policies type-check them; gym is its dependency)
            from gym.spaces import Box, MultiDiscrete
        except ImportError:
            from envs.atc._spaces import Box, MultiDiscrete
        # Small batches (the 8-16 envs stable-baselines users run) are latency-bound: their state and outputs live in pinned
        # host memory mapped into the device, so a vector step is one launch + one synchronisation with the numpy results read
        # in place — no device-to-host copy.  Large batches stay in HBM and cross PCIe once per step.
        if host_mapped is None:
            host_mapped = int(num_envs) * int(num_aircraft) <= 256
        self.vec = AtcVecEnv(num_envs, num_aircraft, sim_parameters=sim_parameters, scenario=scenario, device=device,
                             auto_reset=True, seed=seed, want_raw_obs=True, want_term_obs=True, host_mapped=host_mapped,
                             **kw)
        self.num_envs = self.vec.B
        n = self.vec.N
        sp = self.vec.sim_parameters
        if sp.discrete_action_space:  # atc_gym.py:72-74 (per aircraft)
            self.action_space = MultiDiscrete([20, 380, 360] * n)
        else:                         # atc_gym.py:81-82
            self.action_space = Box(low=-np.ones(3 * n, np.float32), high=np.ones(3 * n, np.float32))
        self.observation_space = Box(low=-1.0, high=1.0, shape=(L.OBS_DIM * n,))  # atc_gym.py:113
        if _Base is not object:   # VecEnv.__init__(num_envs, observation_space, action_space): the same three attributes
            _Base.__init__(self, self.num_envs, self.observation_space, self.action_space)
        self.reward_range = (-3000.0, 23000.0)                                    # atc_gym.py:115
        self.metadata = {'render.modes': ['human', 'rgb_array'], 'video.frames_per_second': 50}
        self._t0 = time.time()
        self._actions = None
        self._act_pinned = None
        # Building one dict per env per step is what bounds large batches.  With sparse infos only finished envs get their
        # own dict (terminal_observation + Monitor's episode record, which is all stable-baselines reads); the others share
        # one empty dict and the raw states of the whole batch are exposed as `self.original_state` [B, 10 N].
        self.sparse_infos = (self.num_envs > 512) if sparse_infos is None else bool(sparse_infos)
        self.original_state = None
        self._no_info = {}

    # -- VecEnv protocol ---------------------------------------------------------------------------------------------
    def reset(self):
        """All envs restart; returns the RAW reset observations [B, 10 N] (atc_gym.py:365 returns the raw state)."""
        return self.vec.reset().cpu().numpy().copy()

    def step_async(self, actions):
        a = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self.vec.N, L.ACT_DIM)
        if self.vec.host_mapped:   # the kernel reads the actions in place from a pinned, mapped buffer
            if self._act_pinned is None:
                self._act_pinned = self.vec.torch.zeros((self.num_envs, self.vec.N, L.ACT_DIM)).pin_memory()
                self._act_np = self._act_pinned.numpy()
            self._act_np[...] = a
            self._actions = self._act_pinned
        else:
            self._actions = a

    def step_wait(self):
        vec = self.vec
        obs, rew, done, info = vec.step(self._actions)
        torch = vec.torch
        d = vec.obs_dim
        if vec.host_mapped:   # results are host memory already (the step ended with a stream synchronisation)
            obs_h, raw_h = obs.numpy().copy(), info["original_state"].numpy().copy()
            term_h = info["terminal_observation"].numpy().copy()
            rew_h, done_h = rew.numpy().copy(), done.numpy() != 0
            # (host_mapped="io" keeps the state and the episode records in HBM: those two come over with a copy)
            ep_r, ep_l = vec.ep_return.cpu().numpy().copy(), vec.ep_length.cpu().numpy().copy()
        else:
            pack = torch.cat([obs, info["original_state"], info["terminal_observation"],
                              rew[:, None], done[:, None].to(torch.float32), vec.ep_return[:, None],
                              vec.ep_length[:, None].to(torch.float32)], dim=1).cpu().numpy()   # one device->host hop
            obs_h, raw_h, term_h = pack[:, :d], pack[:, d:2 * d], pack[:, 2 * d:3 * d]
            rew_h, done_h = pack[:, 3 * d], pack[:, 3 * d + 1] != 0
            ep_r, ep_l = pack[:, 3 * d + 2], pack[:, 3 * d + 3]
        now = round(time.time() - self._t0, 6)
        self.original_state = raw_h
        if self.sparse_infos:
            infos = [self._no_info] * self.num_envs
            for b in np.nonzero(done_h)[0]:
                infos[b] = {"original_state": raw_h[b], "terminal_observation": term_h[b],
                            "episode": {"r": float(ep_r[b]), "l": int(ep_l[b]), "t": now}}
        else:
            infos = []
            for b in range(self.num_envs):
                item = {"original_state": raw_h[b]}
                if done_h[b]:
                    item["terminal_observation"] = term_h[b]
                    item["episode"] = {"r": float(ep_r[b]), "l": int(ep_l[b]), "t": now}
                infos.append(item)
        if vec.host_mapped:
            return obs_h, rew_h, done_h, infos
        return obs_h.copy(), rew_h.copy(), done_h.copy(), infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.vec.close()

    def seed(self, seed=None):
        """VecEnv.seed (stable-baselines >= 2.10, SB3): env i is seeded with `seed + i`, one entry per env in the return value,
        each what AtcGym.seed returns (atc_gym.py:117-126: `[seed]`).  The batch has ONE counter-based generator keyed by
        (seed, env index, episode): env i's stream is its own, like seed + i gives a SubprocVecEnv worker."""
        self.vec.seed(seed)
        return [[None if seed is None else int(seed) + i] for i in range(self.num_envs)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        """SB3's VecEnv.env_is_wrapped: there are no per-env gym wrappers inside the batch."""
        n = self.num_envs if indices is None else len(self._idx(indices))
        return [False] * n

    def get_attr(self, attr_name, indices=None):
        """`actions_per_timestep`, `winning_ratio` (read by the reference's TensorBoard callback,
        learning/atc-gym-stable-baselines.py:34,36), counters, and constant attributes of the env."""
        if attr_name in ("timestep_limit",):
            vals = [self.vec.timestep_limit] * self.num_envs
            return vals if indices is None else [vals[i] for i in self._idx(indices)]
        return self.vec.get_attr(attr_name, None if indices is None else self._idx(indices))

    def set_attr(self, attr_name, value, indices=None):
        """Per-env counters that the reference exposes as plain attributes can be overwritten (e.g. `timesteps`)."""
        t = {"timesteps": self.vec.timesteps, "actions_taken": self.vec.actions_taken,
             "total_reward": self.vec.total_reward}.get(attr_name)
        if t is None:
            raise AttributeError("cannot set %r on the batched env" % attr_name)
        for i in (range(self.num_envs) if indices is None else self._idx(indices)):
            t[i] = value

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        """`reset` and `seed` per env (what SB wrappers call); other methods of the single-env class have no batched
        meaning."""
        idx = list(range(self.num_envs)) if indices is None else self._idx(indices)
        if method_name == "reset":
            mask = np.zeros(self.num_envs, np.uint8)
            mask[idx] = 1
            obs = self.vec.reset(mask=mask).cpu().numpy()
            return [obs[i].copy() for i in idx]
        if method_name == "seed":
            return [self.vec.seed(*method_args, **method_kwargs)[0] for _ in idx]
        raise AttributeError("env_method(%r) is not available on the batched env" % method_name)

    def get_images(self, *args, **kwargs):
        from . import render
        return [render.rgb_array(self.vec, env=b) for b in range(self.num_envs)]

    def render(self, mode="human", *args, **kwargs):
        if mode == "rgb_array":
            from . import render
            return render.rgb_array(self.vec, env=0)
        return None

    def _idx(self, indices):
        return [indices] if isinstance(indices, int) else list(indices)
