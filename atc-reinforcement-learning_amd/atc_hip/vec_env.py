"""AtcVecEnv — B envs x N aircraft of the AtcGym.step() path, all state in device tensors, stepped by libatcstep.so.

Build-own batched surface (the reference has only the single-env AtcGym, atc_gym.py:22-365, and reaches parallelism
through stable-baselines' SubprocVecEnv x8, learning/atc-gym-stable-baselines.py:76-78).  Method names follow the
stable-baselines VecEnv protocol (reset / step / get_attr / seed / close) so that a PPO loop can consume it directly;
tensors stay on the device.  `auto_reset=True` gives VecEnv semantics: a finished env restarts inside the step and the
returned observation is the RAW reset observation (quirk of atc_gym.py:351,365 that SubprocVecEnv workers expose too).
"""
import ctypes as C

import numpy as np

from . import layout as L
from . import lib as _lib


def auto_grid_cell(num_envs, num_aircraft):
    """Cell size [nm] of the MVA lookup grid for a batch, by its number of aircraft slots (envs x next_pow2(aircraft)):
        4 096 .. 131 072 slots   0.0625 nm   (65 536 x 1, 8 192 x 16)
        up to 262 144 slots      0.125 nm    (4 096 x 64; and the tiny batches below 4 096 slots, whose step is a launch latency)
        beyond                   0.25 nm     (65 536 x 16)
    A finer grid puts fewer aircraft into cells a border passes through — whose records are a second dependent L2 round trip
    for the wavefront holding such an aircraft, the longest chain of a step — which is what a small, latency-bound batch feels
    (round 4, split cells in place: 65 536 x 1 fused 3.04 / 3.35 / 3.71 / 4.33 us per step at 0.0625 / 0.125 / 0.25 / 0.5 nm, single
    steps 5.93 / 6.29 / 6.68 / 6.97; 8 192 x 16 fused 2.42 / 2.51 / 2.87 / 3.3); a large batch is bound by HBM or instruction issue
    and only pays for the bigger table (LOWW: 11.6 MB at 0.0625 nm — it lives in the Infinity Cache —, 3.2 MB at 0.125, 0.9 MB at
    0.25; 65 536 x 16: 17.7 us single steps at 0.125 against 17.6 at 0.25, 4 096 x 64 the same at every size).  The sector
    compiler takes 5 s for LOWW at 0.0625 nm (1.4 s at 0.125), once per process and sector.  Results do not depend on the cell
    size (the lookup is exact for any)."""
    w = 1
    while w < int(num_aircraft):
        w *= 2
    slots = int(num_envs) * w
    if 4096 <= slots <= 131072:
        return 0.0625
    return 0.125 if slots <= 262144 else 0.25


class AtcVecEnv:
    def __init__(self, num_envs, num_aircraft=1, sim_parameters=None, scenario=None, device=0, auto_reset=True,
                 spawn="auto", seed=0, grid_cell="auto", want_raw_obs=False, want_ac_reward=False, want_min_sep=False,
                 want_term_obs=False, timestep_limit=6000, sep_nm=3.0, sep_ft=1000.0, conflict_reward=-200.0,
                 host_mapped=False, keep_active=False, want_packet=False, check_held=False, lds_table=True):
        """host_mapped=True keeps state and outputs in pinned host memory mapped into the device (zero-copy): the kernels
        read / write it over the host link, every call ends with a stream synchronisation, and what is returned are CPU
        tensors.  Meant for tiny latency-bound batches (the single-env AtcGym); large batches belong in HBM.
        host_mapped="io" maps only what crosses the host link every step (actions in, results out); the state stays in HBM.
        want_packet=True (host_mapped, N == 1) adds atc_out_t.packet: the step result as self-validating 16-byte chunks that a
        host can poll in mapped memory instead of synchronising the stream (see `poll_packet`).
        keep_active=True is the reference's single-aircraft rule (ATC_M_KEEP_ACTIVE): an aircraft that reaches the corridor
        ends the episode and stays under control instead of being handed over.
        check_held=True (debugging aid) verifies the promise of step(..., held=True) — the actions equal those of the previous
        step — on every such call and raises if it is broken (costs a device comparison and a synchronisation per step)."""
        torch = _lib._torch_cuda()
        self.torch = torch
        self._check_held = bool(check_held)
        self._prev_actions = None
        self.host_mapped = bool(host_mapped)
        from envs.atc import model, scenarios
        self.sim_parameters = sim_parameters if sim_parameters is not None else model.SimParameters(1)
        self.scenario_obj = scenario if scenario is not None else scenarios.LOWW()
        if not 1 <= num_aircraft <= L.MAX_AIRCRAFT:
            raise ValueError("1 <= num_aircraft <= %d" % L.MAX_AIRCRAFT)
        self.B, self.N = int(num_envs), int(num_aircraft)
        self.num_envs = self.B
        if grid_cell == "auto":
            # the batch's preferred cell size, or the next coarser one the sector's blob can hold (a sector a few times LOWW's
            # size does not fit 2^24 words at 0.0625 nm): an explicitly requested size that does not fit raises SectorTooLarge
            from .scenario import SectorTooLarge
            want = auto_grid_cell(self.B, self.N)
            for grid_cell in [c for c in (0.0625, 0.125, 0.25, 0.5, 1.0, 2.0) if c >= want] + [None]:
                try:
                    self.compiled = scenarios.compile_scenario(self.scenario_obj, grid_cell=grid_cell)
                    break
                except SectorTooLarge:
                    if grid_cell is None:
                        raise
        else:
            self.compiled = scenarios.compile_scenario(self.scenario_obj, grid_cell=grid_cell)
        self.grid_cell = grid_cell
        # (one-aircraft envs: the sector's LDS-resident lookup table goes along — the multi-step launches of batches that fit one
        # workgroup per CU answer the MVA lookup from LDS, include/atc_step.h ABI 21; `lds_table=False` keeps it off, for A/B runs)
        # (... batches the LDSG launch can serve: whole workgroups of 256 one-aircraft envs — a single env never gets there)
        self.sector = _lib.Scenario(self.compiled, device, lds_table=(self.N == 1 and self.B % 256 == 0 and bool(lds_table)))
        self.device = self.sector.device
        n_entry = self.compiled.n_entry
        if n_entry < 1:
            raise ValueError("scenario has no entry point")
        if spawn == "auto":
            spawn = "random" if (self.N == 1 and n_entry > 1) else "lattice"
        if spawn not in ("random", "lattice"):
            raise ValueError("spawn must be 'auto', 'random' or 'lattice'")
        sp = self.sim_parameters
        self.params = _lib.make_params(dt=sp.timestep, shaping=sp.reward_shaping, normalize=sp.normalize_state,
                                       discrete=sp.discrete_action_space, auto_reset=auto_reset,
                                       random_entry=(spawn == "random"), seed=seed, timestep_limit=timestep_limit,
                                       sep_nm=sep_nm, sep_ft=sep_ft, conflict_reward=conflict_reward,
                                       keep_active=keep_active)
        self.timestep_limit = timestep_limit
        B, N, BN, dev = self.B, self.N, self.B * self.N, self.device
        if self.host_mapped:
            z = lambda shape, dt: torch.zeros(shape, dtype=dt).pin_memory()  # noqa: E731
        else:
            z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        # host_mapped="io": only what crosses the host link every step (actions in, results out) lives in mapped host memory;
        # the state stays in HBM, so the kernel's state loads and stores do not pay the link's round trip
        zs = (lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)) if host_mapped == "io" else z  # noqa: E731
        f32, i32 = torch.float32, torch.int32
        # persistent state (atc_state_t): packed records, see include/atc_step.h
        self.ac = zs((BN, L.AC_WORDS), i32)    # x, y (position-grid counts), phi (heading counts), v (speed counts, unsigned)
        self.alt = zs(BN, torch.float64)       # altitude [ft]: the reference's float64 (ABI 20)
        self.last_act = zs((BN, L.LA_WORDS), i32)   # last accepted targets: v counts, phi counts, altitude target (float64, words 2..3)
        self.env = zs((B, L.ENV_WORDS), i32)   # per-step env record
        self.stats = zs((B, L.STAT_WORDS), i32)  # per-episode env record
        self.pos_origin, self.pos_k = self.compiled.pos_origin, self.compiled.pos_k
        # named views into the records (live memory, usable for reads and in-place writes)
        self.h = self.alt
        self.phi_fix = self.ac[:, L.AC_PHI]    # heading counts (deg = 180 + phi_fix 2^-23); `phi` / `v` below are copies in units
        self.v_fix = self.ac[:, L.AC_V]        # speed counts (unsigned 32-bit in an int32 word, kt = v_fix 2^-23)
        self.timesteps = self.env[:, L.ENV_TIMESTEPS]
        self.actions_taken = self.env[:, L.ENV_ACTIONS_TAKEN]
        self.total_reward = self.env[:, L.ENV_TOTAL_REWARD:L.ENV_TOTAL_REWARD + 1].view(f32).squeeze(1)
        self.episodes = self.stats[:, L.STAT_EPISODES]
        self.ep_length = self.stats[:, L.STAT_EP_LENGTH]
        self.ep_return = self.stats[:, L.STAT_EP_RETURN:L.STAT_EP_RETURN + 1].view(f32).squeeze(1)
        self.win_bits = self.stats[:, L.STAT_WIN_BITS]
        self.ep_actions = self.stats[:, L.STAT_EP_ACTIONS]
        # per-step outputs (atc_out_t)
        self.obs = z((B, N * L.OBS_DIM), f32)
        self.raw_obs = z((B, N * L.OBS_DIM), f32) if want_raw_obs else None
        self.reward = z(B, f32)
        self.ac_reward = z((B, N), f32) if want_ac_reward else None
        self.done = z(B, torch.uint8)
        self.flags = z((B, N), torch.int16)
        self.min_sep = z(B, f32) if want_min_sep else None
        self.term_obs = z((B, N * L.OBS_DIM), f32) if want_term_obs else None
        if want_packet and not (self.host_mapped and N == 1):
            raise ValueError("want_packet needs host_mapped=True and num_aircraft=1")
        self.packet = z((B, L.PKT_CHUNKS, 4), i32) if want_packet else None
        # exact heading counts / last heading target of aircraft whose 32-bit heading fields are saturated ("WIDE", ABI 19): the
        # reference's heading is unbounded (model.py:104-120).  Untouched while headings stay inside (-76, 436) deg.
        # (allocated after everything a step streams through, so that those tensors sit where they would without it)
        self.phi_wide = zs((BN, L.PHI_WIDE_WORDS), torch.float64)
        self._state = _lib.AtcState(*[self._ptr(getattr(self, n)) for n in _lib.STATE_FIELDS])
        # twin of `params` with ATC_M_ACTIONS_HELD set: what step(held=True) and the held launchers pass.  A persistent object,
        # refreshed by seed() / refresh_params(), so that pre-bound launchers see parameter changes like plain launches do
        self._params_held = type(self.params).from_buffer_copy(self.params)
        self._bind_outputs()
        self.refresh_params()
        self._lib = _lib.load()
        self.reset(first=True)

    # ------------------------------------------------------------------------------------------------ plumbing
    def pack_outputs(self):
        """Re-homes obs / raw_obs / reward / flags / done of a small env batch in ONE contiguous byte buffer (plus a pinned
        host mirror) so that a host-side caller (AtcGym) fetches a whole step result with a single device->host copy.
        Returns (device_bytes, host_bytes, layout) with layout[name] = (offset, nbytes); in host-mapped mode the two are
        the same pinned buffer and no copy is needed at all."""
        torch = self.torch
        assert self.raw_obs is not None
        B, N = self.B, self.N
        sizes = [("obs", B * N * L.OBS_DIM * 4), ("raw_obs", B * N * L.OBS_DIM * 4), ("reward", B * 4),
                 ("flags", B * N * 2), ("done", B)]
        layout, off = {}, 0
        for name, nb in sizes:
            layout[name] = (off, nb)
            off += (nb + 15) & ~15
        host = torch.zeros(off, dtype=torch.uint8).pin_memory()
        dev = host if self.host_mapped else torch.zeros(off, dtype=torch.uint8, device=self.device)
        view = lambda name, dt, shape: dev[layout[name][0]:layout[name][0] + layout[name][1]].view(dt).view(shape)  # noqa: E731
        self.obs = view("obs", torch.float32, (B, N * L.OBS_DIM))
        self.raw_obs = view("raw_obs", torch.float32, (B, N * L.OBS_DIM))
        self.reward = view("reward", torch.float32, (B,))
        self.flags = view("flags", torch.int16, (B, N))
        self.done = view("done", torch.uint8, (B,))
        self._bind_outputs()
        return dev, host, layout

    def _ptr(self, t):
        """Device address of a tensor this env hands to the library (mapped address for pinned host tensors)."""
        if t is None:
            return None
        return t.data_ptr() if t.is_cuda else _lib.mapped_ptr(t)

    def _make_out(self, *tensors):
        return _lib.AtcOut(*[self._ptr(t) for t in tensors])

    def _bind_outputs(self):
        """(Re)builds atc_out_t for this env's own output tensors and everything step() reuses from call to call."""
        self._out = self._make_out(self.obs, self.raw_obs, self.reward, self.ac_reward, self.done, self.flags,
                                   self.min_sep, self.term_obs, getattr(self, "packet", None))
        self._out_ref = C.byref(self._out)
        self._state_ref = C.byref(self._state)
        self._params_ref = C.byref(self.params)
        self._params_held_ref = C.byref(self._params_held)
        self._n_act = self.B * self.N * L.ACT_DIM
        self._info_cache = self._info()

    def _finish(self):
        if self.host_mapped:  # results live in host memory: valid only once the stream has drained
            self.torch.cuda.current_stream(self.device).synchronize()

    def _stream(self):
        return _lib.current_stream_ptr(self.device)

    @property
    def action_dim(self):
        return self.N * L.ACT_DIM

    @property
    def obs_dim(self):
        return self.N * L.OBS_DIM

    # ------------------------------------------------------------------------------------------------ VecEnv surface
    def seed(self, seed=None):
        self.params.seed = int(seed or 0) & (2 ** 64 - 1)
        self.refresh_params()
        return [seed] * self.B

    def refresh_params(self):
        """Copies `params` into its held-action twin (the struct launches with ATC_M_ACTIONS_HELD pass).  seed() calls it; call
        it after changing a field of `env.params` directly, or launchers made with held=True keep the old value."""
        C.memmove(C.byref(self._params_held), C.byref(self.params), C.sizeof(self.params))
        self._params_held.mode |= L.M_ACTIONS_HELD

    def reset(self, mask=None, first=False):
        """AtcGym.reset (atc_gym.py:337-365) for all envs (or those with mask != 0); returns RAW obs [B, N*10]."""
        torch = self.torch
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.atc_reset(self.sector.handle, self.B, self.N, C.byref(self._state),
                                           m.data_ptr() if m is not None else None, self._ptr(self.obs),
                                           C.byref(self.params), int(first), self._stream()))
        self._finish()
        return self.obs

    def observe(self, mask=None):
        """Raw observation (mva = 0) of the current state, like the tail of AtcGym.reset (atc_gym.py:351,365)."""
        torch = self.torch
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.atc_observe(self.sector.handle, self.B, self.N, C.byref(self._state),
                                             m.data_ptr() if m is not None else None, self._ptr(self.obs),
                                             C.byref(self.params), self._stream()))
        self._finish()
        return self.obs

    def _as_actions(self, actions, lead=()):
        torch = self.torch
        a = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions, dtype=np.float32))
        if not (self.host_mapped and not a.is_cuda and a.is_pinned() and a.dtype == torch.float32):
            a = a.to(device=self.device, dtype=torch.float32)
        a = a.contiguous()
        want = int(np.prod(lead, dtype=np.int64)) * self.B * self.N * L.ACT_DIM if lead else self.B * self.N * L.ACT_DIM
        if a.numel() != want:
            raise ValueError("actions must have %d elements, got %d" % (want, a.numel()))
        return a

    def step(self, actions, held=False):
        """AtcGym.step (atc_gym.py:128-192) for every env.  actions: [B, N, 3] (or [B, N*3]) float tensor / array:
        continuous in [-1, 1] or discrete indices (atc_gym.py:318-335).  Returns (obs [B,N*10], reward [B], done [B]
        uint8, info) — device tensors that are overwritten by the next step.
        held=True is the caller's promise that `actions` holds the same values as in the previous step of these envs (a
        held action block / frame skip, learning/atc-gym-demo.py:18-19; ATC_M_ACTIONS_HELD): same results, and the kernel
        skips the last-action record."""
        if self._check_held:
            cur = self._as_actions(actions).reshape(-1).to("cpu", copy=True)
            if held and (self._prev_actions is None or not self.torch.equal(cur.view(self.torch.int32),
                                                                              self._prev_actions.view(self.torch.int32))):
                raise ValueError("step(held=True): the actions differ from the previous step's (or there was none)")
            self._prev_actions = cur
        torch = self.torch
        pref = self._params_held_ref if held else self._params_ref
        # fast path: a float32 device tensor of the right size on this env's (current) device is handed over as it is —
        # small batches are host-bound otherwise (8 192 x 16: 6.4 us on the GPU against 10.4 us of Python per call)
        if (torch.is_tensor(actions) and actions.is_cuda and actions.dtype is torch.float32 and actions.is_contiguous()
                and actions.numel() == self._n_act and actions.device == self.device
                and torch.cuda.current_device() == self.device.index and not self.host_mapped):
            rc = self._lib.atc_step(self.sector.handle, self.B, self.N, self._state_ref, actions.data_ptr(), self._out_ref,
                                    pref, torch.cuda.current_stream().cuda_stream)
            if rc:
                _lib.check(rc)
            self._keep = actions
            return self.obs, self.reward, self.done, self._info_cache
        a = self._as_actions(actions)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.atc_step(self.sector.handle, self.B, self.N, C.byref(self._state), self._ptr(a),
                                          C.byref(self._out), pref, self._stream()))
        self._keep = a
        self._finish()
        return self.obs, self.reward, self.done, self._info_cache

    def make_launcher(self, actions, stream=None, held=False):
        """Pre-bound `atc_step` call for FIXED buffers (this env's state / outputs, the given device action tensor, the
        given torch stream or the current one): returns a no-argument callable that only launches — host cost of a few
        microseconds instead of the argument handling of step().  Meant for pipelined actors that keep several
        independent sub-batches in flight on separate streams (tools/multi_stream.py, bench.py --streams): a sub-batch's
        launch ramp and tail then overlap the others' bodies.  Results are in self.obs / reward / done / flags once the
        stream has reached the launch.  held=True passes the env's held-action parameter twin (ATC_M_ACTIONS_HELD set; the live
        object, so seed() / refresh_params() reach launchers that already exist): for the launches of a held action block after
        its first (see step()).  The promise itself — same actions as the previous launch, no set_last_action() in between —
        is the caller's and is not checked here (AtcVecEnv(check_held=True) checks it in step())."""
        torch = self.torch
        a = self._as_actions(actions)
        q = C.c_void_p((stream if stream is not None else torch.cuda.current_stream(self.device)).cuda_stream)
        params = self._params_held if held else self.params
        args = (self.sector.handle, self.B, self.N, C.byref(self._state), C.c_void_p(self._ptr(a)), C.byref(self._out),
                C.byref(params), q)
        fn, check = self._lib.atc_step, _lib.check

        def launch(_keep=(a, stream, params)):
            rc = fn(*args)
            if rc:
                check(rc)
        return launch

    def step_call(self, actions, stream=None, held=False):
        """The arguments of one atc_step of this env as an `atc_step_call_t` (+ the objects that must outlive it)."""
        a = self._as_actions(actions)
        q = (stream if stream is not None else self.torch.cuda.current_stream(self.device)).cuda_stream
        params = self._params_held if held else self.params   # the live objects, see make_launcher
        call = _lib.AtcStepCall(self.sector.handle, self.B, self.N, C.pointer(self._state), self._ptr(a),
                                C.pointer(self._out), C.pointer(params), q)
        return call, (a, stream, self, params)

    def step_async(self, actions):
        self._pending = actions

    def step_wait(self):
        return self.step(self._pending)

    def _info(self):
        info = {"flags": self.flags, "ep_return": self.ep_return, "ep_length": self.ep_length}
        if self.raw_obs is not None:
            info["original_state"] = self.raw_obs
        if self.ac_reward is not None:
            info["aircraft_reward"] = self.ac_reward
        if self.min_sep is not None:
            info["min_separation"] = self.min_sep
        if self.term_obs is not None:
            info["terminal_observation"] = self.term_obs
        return info

    def rollout(self, actions, out=None, hold=1):
        """T consecutive steps in one launch (state stays in registers).  actions: [T / hold, B, N, 3]; each action block is
        applied for `hold` consecutive steps (frame skip, learning/atc-gym-demo.py:18-19), so T = hold * actions.shape[0].
        Returns a dict of [T, ...] device tensors (obs, reward, done, flags [+ optional outputs when `out` provides them])."""
        torch = self.torch
        hold = int(hold)
        n_blocks = int(actions.shape[0])
        T = n_blocks * hold
        a = self._as_actions(actions, lead=(n_blocks,))
        B, N, dev = self.B, self.N, self.device
        if out is None:
            out = {
                "obs": torch.empty((T, B, N * L.OBS_DIM), dtype=torch.float32, device=dev),
                "reward": torch.empty((T, B), dtype=torch.float32, device=dev),
                "done": torch.empty((T, B), dtype=torch.uint8, device=dev),
                "flags": torch.empty((T, B, N), dtype=torch.int16, device=dev),
            }
        o = self._make_out(out["obs"], out.get("raw_obs"), out["reward"], out.get("ac_reward"), out["done"],
                           out["flags"], out.get("min_sep"), out.get("term_obs"), None)
        with torch.cuda.device(dev):
            _lib.check(self._lib.atc_rollout_hold(self.sector.handle, B, N, T, hold, C.byref(self._state), self._ptr(a),
                                                  C.byref(o), C.byref(self.params), self._stream()))
        self._keep = a
        self._finish()
        return out

    def get_attr(self, name, indices=None):
        """VecEnv.get_attr for the attributes the reference's trainer reads (learning/atc-gym-stable-baselines.py:34,36)
        and the episode counters."""
        torch = self.torch
        if name == "actions_per_timestep":
            # atc_gym.py:197 sets it on every step and reset() leaves it alone: an env that has just been (auto-)reset still
            # reports the value of its last episode's final step
            live = self.actions_taken.to(torch.float64) / self.timesteps.clamp(min=1).to(torch.float64)
            last = self.ep_actions.to(torch.float64) / self.ep_length.clamp(min=1).to(torch.float64)
            t = torch.where(self.timesteps > 0, live, last)
        elif name == "winning_ratio":  # atc_gym.py:362-363: mean of the last 10 episode outcomes
            bits = self.win_bits.to(torch.int64)
            t = sum(((bits >> k) & 1) for k in range(10)).to(torch.float64) / 10.0
        elif name in ("timesteps", "actions_taken", "total_reward", "episodes", "ep_return", "ep_length", "ep_actions"):
            t = getattr(self, name)
        else:
            raise AttributeError(name)
        vals = t.cpu().tolist()
        if indices is not None:
            vals = [vals[i] for i in indices]
        return vals

    @property
    def active_mask(self):
        """u64 mask per env (bit k = aircraft k still under control) as an int64 tensor."""
        lo = self.env[:, L.ENV_MASK_LO].to(self.torch.int64) & 0xffffffff
        hi = self.stats[:, L.STAT_MASK_HI].to(self.torch.int64) & 0xffffffff
        return lo | (hi << 32)

    # positions live on the sector's 32-bit fixed-point grid (include/atc_step.h "Aircraft positions"):
    # nm = origin + counts * 2^-k.  `x` / `y` are float64 COPIES in nautical miles; write through set_state / set_xy.
    # Speed and heading are fixed point too (ABI 18): kt = v_fix 2^-23 (unsigned counts), deg = 180 + phi_fix 2^-23; `v` / `phi`
    # are float64 copies in knots / degrees (exact), written through set_state / set_v / set_phi.
    @property
    def x(self):
        return self.ac[:, L.AC_X].to(self.torch.float64) * 2.0 ** -self.pos_k + self.pos_origin[0]

    @property
    def y(self):
        return self.ac[:, L.AC_Y].to(self.torch.float64) * 2.0 ** -self.pos_k + self.pos_origin[1]

    @property
    def v(self):
        return (self.v_fix.to(self.torch.int64) & 0xffffffff).to(self.torch.float64) * 2.0 ** -L.V_FIX_SHIFT

    @property
    def phi_counts(self):
        """exact heading counts as float64 (integer-valued): the 32-bit field, or the side record where that is saturated"""
        f = self.phi_fix
        wide = (f == L.I32_MIN) | (f == L.I32_MAX)
        return self.torch.where(wide, self.phi_wide[:, 0], f.to(self.torch.float64))

    @property
    def phi(self):
        return self.phi_counts * 2.0 ** -L.PHI_FIX_SHIFT + L.PHI_FIX_OFFSET

    def _to_fix(self, value, axis):
        from .scenario import to_fix
        return int(to_fix(float(value), self.pos_origin[axis], self.pos_k))

    @staticmethod
    def _v_counts(v):
        """knots -> speed counts as the int32 word that holds the unsigned value"""
        c = int(min(max(round(float(v) * 2.0 ** L.V_FIX_SHIFT), 0), 2 ** 32 - 1))
        return c - (1 << 32) if c >= 1 << 31 else c

    @staticmethod
    def _phi_counts(phi):
        """degrees -> (32-bit field, exact counts): the field saturates, the exact counts are clamped to +-2^52 (include/atc_step.h)"""
        P = int(min(max(round((float(phi) - L.PHI_FIX_OFFSET) * 2.0 ** L.PHI_FIX_SHIFT), -L.PHI_LIMIT), L.PHI_LIMIT))
        return min(max(P, L.I32_MIN), L.I32_MAX), P

    def _put_phi(self, field, i, col, phi):
        f, P = self._phi_counts(phi)
        field[i] = f
        if f in (L.I32_MIN, L.I32_MAX):
            self.phi_wide[i, col] = float(P)

    def set_xy(self, i, x=None, y=None):
        if x is not None:
            self.ac[i, L.AC_X] = self._to_fix(x, 0)
        if y is not None:
            self.ac[i, L.AC_Y] = self._to_fix(y, 1)

    def set_v(self, i, v):
        # The speed's rate limit is a wrapping 32-bit difference (include/atc_step.h): exact while the speed lies within 256 kt
        # of every acceptable target [100, 300] kt, i.e. inside [44, 356] kt.  The reference's constructor refuses anything outside
        # [100, 300] (model.py:22-23); a speed poked in from outside (env._airplane.v = ...) is refused beyond the format's range.
        if not 44.0 <= float(v) <= 356.0:
            raise ValueError("invalid velocity: the device's speed format holds 44 .. 356 kt (an Airplane has 100 .. 300)")
        self.v_fix[i] = self._v_counts(v)

    def set_phi(self, i, phi):
        self._put_phi(self.phi_fix, i, 0, phi)

    def set_state(self, env, slot, x, y, h, phi, v):
        i = env * self.N + slot
        self.set_xy(i, x, y)
        self.h[i] = float(h)
        self.set_phi(i, phi)
        self.set_v(i, v)

    def get_last_action(self, env, slot):
        """AtcGym.last_action (atc_gym.py:86,311) of one aircraft: [v, h, phi] targets last accepted, in kt / ft / deg."""
        i = env * self.N + slot
        rec = self.last_act[i].cpu()
        lp = int(rec[L.LA_PHI])
        if lp in (L.I32_MIN, L.I32_MAX):
            lp = float(self.phi_wide[i, 1])
        return [float(int(rec[L.LA_V]) & 0xffffffff) * 2.0 ** -L.V_FIX_SHIFT,
                float(rec[L.LA_H:L.LA_H + 2].contiguous().view(self.torch.float64)[0]),
                float(lp) * 2.0 ** -L.PHI_FIX_SHIFT + L.PHI_FIX_OFFSET]

    def set_last_action(self, env, slot, value):
        i = env * self.N + slot
        torch = self.torch
        f, P = self._phi_counts(value[2])
        rec = torch.tensor([self._v_counts(value[0]), f, 0, 0], dtype=torch.int32)
        rec[L.LA_H:L.LA_H + 2].view(torch.float64)[0] = float(value[1])
        self.last_act[i] = rec.to(self.last_act.device)
        if f in (L.I32_MIN, L.I32_MAX):
            self.phi_wide[i, 1] = float(P)

    def get_state(self, env, slot):
        i = env * self.N + slot
        return [float(t[i]) for t in (self.x, self.y, self.h, self.phi, self.v)]

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    def close(self):
        self.sector.close()


def make_multi_launcher(envs, actions, streams, held=False):
    """One step of several INDEPENDENT sub-batch envs with a single foreign call (`atc_step_multi`): envs[i] steps with
    the device tensor actions[i] on streams[i].  Returns a no-argument callable that only launches.  With no join between
    steps the sub-batches run decoupled: the launch ramp and tail of one overlap the body of the others (bench.py
    --streams, tools/multi_stream.py)."""
    n = len(envs)
    assert n == len(actions) == len(streams) and n >= 1
    built = [e.step_call(a, q, held) for e, a, q in zip(envs, actions, streams)]
    arr = (_lib.AtcStepCall * n)(*[b[0] for b in built])
    fn, check = _lib.load().atc_step_multi, _lib.check

    def launch(_keep=(built, arr)):
        rc = fn(n, arr)
        if rc:
            check(rc)
    return launch
