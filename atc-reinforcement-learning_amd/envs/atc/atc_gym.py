"""AtcGym — single-environment gym.Env surface of the reference (envs/atc/atc_gym.py:22-365), backed by the HIP step
kernels (one env x one aircraft on the GPU; there is no CPU step path).

Everything the reference's callers touch is kept: ctor signature, seed/reset/step/render/close, action_space,
observation_space, reward_range, metadata, the metric attributes read through VecEnv.get_attr
(actions_per_timestep, winning_ratio; learning/atc-gym-stable-baselines.py:34,36) and info["original_state"]
(:101-103).  The arithmetic of step() runs in libatcstep.so; this class only does the episode bookkeeping the
reference does in Python (win ring buffer, counters), in the same order, including its quirks:
  * reset() returns the RAW state, step() the normalised one (atc_gym.py:365 vs :187-192)
  * last_action is initialised once in __init__ and never reset (atc_gym.py:86)
  * no auto-reset, stepping a finished episode keeps counting (learning/atc-gym-compute-performance.py relies on it)
  * the win buffer gets one append per terminal CAUSE (atc_gym.py:151,158,165) and reset() pops exactly one (:359-363)
"""
import random
import weakref
from time import perf_counter as _perf_counter

import numpy as np

from atc_hip import layout as L
from . import model
from . import scenarios
from ._spaces import Box, Env, MultiDiscrete


class _AirplaneView:
    """`env._airplane` of the reference (model.py:13-52) as a live view of the device state."""
    h_min, h_max, v_min, v_max = 0, 38000, 100, 300
    h_dot_min, h_dot_max, a_max, a_min, phi_dot_max, phi_dot_min = -41, 15, 5, -5, 3, -3

    def __init__(self, env, name="FLT01"):
        object.__setattr__(self, "_env", env)
        object.__setattr__(self, "name", name)
        object.__setattr__(self, "id", 0)
        object.__setattr__(self, "position_history", [])   # model.py:51,123: (x, y) before every move, new list per reset

    def __getattr__(self, key):
        if key in ("x", "y", "h", "phi", "v"):
            return float(getattr(self._env._vec, key)[0])      # (env._vec drains the stream first)
        raise AttributeError(key)

    def __setattr__(self, key, value):
        if key in ("x", "y"):   # positions live on the device's fixed-point grid (include/atc_step.h)
            self._env._vec.set_xy(0, **{key: float(value)})
            self._env._pos_now = None
        elif key == "h":
            self._env._vec.h[0] = float(value)
        elif key == "phi":      # speed and heading are fixed point on the device too (ABI 18; set_phi places any heading, also beyond the 32-bit field: ABI 19)
            self._env._vec.set_phi(0, value)
        elif key == "v":
            self._env._vec.set_v(0, value)
        else:
            object.__setattr__(self, key, value)


# Persistent step servers of this process (AtcGym._serving): every one is a resident kernel on its own HIP stream, and a process has
# only a few hardware queues to put streams on (4 by default) — one more resident kernel than queues and somebody's launches wait
# behind a server until its lease runs out.  Envs beyond the cap step by the launch path (several AtcGym in one process, e.g. a
# DummyVecEnv; the reference's own arrangement is one env per process, learning/atc-gym-stable-baselines.py:69-80).
_SERVING = {}          # id(env) -> weakref(env)
_MAX_SERVERS = 2
# A resident kernel also blocks whatever ELSE is submitted to a stream that shares its hardware queue (streams beyond the first few
# share queues) until it leaves.  So the server is a tight-loop device: it lingers for one short lease after a step, and a step is
# only served when the previous one ended less than _TIGHT_GAP_S ago (the reference's FPS loop, a rollout loop with a CPU policy);
# a caller that does other work between steps — policy inference on this GPU — steps by launches and never finds a server in its way
# for longer than a kernel launch takes.
_LEASE_US = 50
_TIGHT_GAP_S = 50e-6


def _server_slot_free():
    """True if another resident kernel fits; first forgets servers that have left by themselves (lease) or whose env is gone."""
    for key, ref in list(_SERVING.items()):
        env = ref()
        if env is None:
            del _SERVING[key]
        elif int(env._mailbox[4]) >= 2:       # left: its kernel has ended
            env._srv_stream.synchronize()
            env._serving = False
            del _SERVING[key]
    return len(_SERVING) < _MAX_SERVERS


class AtcGym(Env):
    metadata = {
        'render.modes': ['human', 'rgb_array'],
        'video.frames_per_second': 50
    }

    def __init__(self, sim_parameters=None, scenario=None, device=0, persistent=None):
        """persistent (build-own keyword; default: on, ATC_GYM_SERVER=0 turns it off): step through the library's persistent step
        server (atc_serve_*: one resident wavefront polling a mailbox in mapped host memory) instead of one kernel launch per
        step WHILE steps follow each other within 50 us (a tight stepping loop); otherwise by launches.  Same results bit for
        bit; the server is stopped before anything else touches the env's device state and leaves by itself 50 us after its last
        step."""
        # the reference evaluates its defaults once at import (atc_gym.py:28): SimParameters(1), LOWW()
        sim_parameters = sim_parameters if sim_parameters is not None else model.SimParameters(1)
        scenario = scenario if scenario is not None else scenarios.LOWW()
        self.last_reward = 0
        self.total_reward = 0
        self.actions_taken = 0
        self._episodes_run = 0
        self._actions_ignoring_resets = 0
        self._won_simulations_ignoring_resets = 0
        self._win_buffer_size = 10
        self._win_buffer = [0] * self._win_buffer_size
        self.actions_per_timestep = 0
        self.timesteps = 0
        self.timestep_limit = 6000
        self.winning_ratio = 0
        self._sim_parameters = sim_parameters
        self._scenario = scenario
        self._mvas = scenario.mvas
        self._runway = scenario.runway
        self._airspace = scenario.airspace

        self._backend = self._make_backend(sim_parameters, scenario, device)
        self._outstanding = False
        self._serving = False
        if persistent is None:
            import os
            persistent = os.environ.get("ATC_GYM_SERVER", "1") != "0"
        self._persistent = bool(persistent)
        torch = self._backend.torch
        # Zero-copy step: aircraft state, the action and everything step() returns live in pinned host memory that is
        # mapped into the device (AtcVecEnv(host_mapped=True)); one step = write 3 floats, one kernel launch, one stream
        # synchronisation, read the results in place.  No copy is launched around the step.
        _, self._host_out, self._out_layout = self._vec.pack_outputs()
        self._host_act = torch.zeros(3, dtype=torch.float32).pin_memory()
        self._act_np = self._host_act.numpy()
        self._out_np = self._host_out.numpy()
        self._pos_rec = self._vec.ac       # (device memory in "io" mode: read with a copy after a reset only)
        self._pos_inv = 2.0 ** -self._vec.pos_k
        self._pos_origin = tuple(self._vec.pos_origin)
        lay = self._out_layout
        f32 = lambda name: self._out_np[lay[name][0]:lay[name][0] + lay[name][1]].view(np.float32)  # noqa: E731
        self._obs_np, self._raw_np, self._rew_np = f32("obs"), f32("raw_obs"), f32("reward")
        self._flags_np = self._out_np[lay["flags"][0]:lay["flags"][0] + 2].view(np.uint16)
        self._done_np = self._out_np[lay["done"][0]:lay["done"][0] + 1]
        import ctypes as C
        from atc_hip import lib as _lib
        v = self._vec
        self._check = _lib.check
        self._launch = lambda stream, _f=v._lib.atc_step, _h=v.sector.handle, _s=C.byref(v._state), \
            _a=C.c_void_p(_lib.mapped_ptr(self._host_act)), _o=C.byref(v._out), _p=C.byref(v.params): \
            _f(_h, 1, 1, _s, _a, _o, _p, stream)
        self._current_stream = torch.cuda.current_stream
        # the raw handle of the current stream without building a Stream object per step (private torch API, optional)
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        dev, dev_index = v.device, v.device.index
        self._raw_stream = (lambda: raw(dev_index)) if raw is not None else (lambda: torch.cuda.current_stream(dev).cuda_stream)
        # Completion without a stream synchronisation: the kernel also writes the step result as 9 self-validating 16-byte
        # chunks (atc_out_t.packet), each ONE store tagged with this step's sequence number.  The host polls the tags in the
        # mapped buffer; a chunk whose tag is current holds this step's payload whatever order the chunks arrived in (this is
        # what makes polling safe — plain completion-word polling is not: stores to mapped memory are not ordered among
        # themselves).  Anything else that touches device-side state first drains the stream (_settle).
        self._pkt_i = self._vec.packet.numpy().reshape(L.PKT_CHUNKS, 4)
        self._pkt_tags = self._pkt_i[:, 3]
        self._pkt_f = self._pkt_i.view(np.float32)
        # launch + poll in ONE foreign call (atc_step_packet): the 27 payload words land in _payload
        self._payload_i = np.zeros(27, np.int32)     # (signed: the grid counts are)
        self._payload_f = self._payload_i.view(np.float32)
        self._step_packet = lambda stream, seq, _f=v._lib.atc_step_packet, _h=v.sector.handle, _s=C.byref(v._state), \
            _a=C.c_void_p(self._host_act.data_ptr()), _o=C.byref(v._out), _p=C.byref(v.params), \
            _k=C.c_void_p(self._vec.packet.data_ptr()), _w=C.c_void_p(self._payload_i.ctypes.data): \
            _f(_h, _s, _a, _o, _p, seq, _k, _w, 20000, stream)   # 20 ms: a first launch on an idle device can take a while
        # the persistent step server's mailbox (64 bytes of pinned mapped memory) and its three calls
        self._mailbox = torch.zeros(16, dtype=torch.int32).pin_memory()
        assert self._mailbox.data_ptr() % 64 == 0
        _mb = C.c_void_p(self._mailbox.data_ptr())
        self._serve_start = lambda stream, last, _f=v._lib.atc_serve_start, _h=v.sector.handle, _s=C.byref(v._state), _o=C.byref(v._out), \
            _p=C.byref(v.params): _f(_h, _s, _o, _p, _mb, last, _LEASE_US, stream)
        self._serve_step = lambda seq, _f=v._lib.atc_serve_step, _a=C.c_void_p(self._host_act.data_ptr()), \
            _k=C.c_void_p(self._vec.packet.data_ptr()), _w=C.c_void_p(self._payload_i.ctypes.data): _f(_mb, _a, seq, _k, _w, 2000000)
        self._serve_stop = lambda stream, _f=v._lib.atc_serve_stop: _f(_mb, stream)
        self._srv_stream = None        # the server's own (non-blocking) stream: nothing else is ever launched on it
        self._t_done = 0.0             # when the last step returned (time.perf_counter)
        self._seq = 0
        self._outstanding = False
        self._pos_now = None                   # grid position after the last step (None: read it from the state record)
        comp = self._vec.compiled
        self._faf_mva = int(comp.faf_mva)
        self._world_x_min, self._world_y_min, self._world_x_max, self._world_y_max = comp.bbox
        self._world_max_distance = comp.world_diag
        self._airplane = _AirplaneView(self)

        self.done = True
        self.reset()
        self.viewer = None

        self.normalization_action_offset = np.array([self._airplane.v_min, 0, 0])
        if sim_parameters.discrete_action_space:
            self.normalization_action_factor = np.array([10, 100, 1])
            self.action_space = MultiDiscrete([int((self._airplane.v_max - self._airplane.v_min) / 10),
                                               int(self._airplane.h_max / 100), 360])
        else:
            self.normalization_action_factor = np.array([self._airplane.v_max - self._airplane.v_min,
                                                         self._airplane.h_max, 360])
            self.action_space = Box(low=np.array([-1, -1, -1]), high=np.array([1, 1, 1]))
        self._action_discriminator = [5, 50, 0.5]
        self.normalization_state_min = comp.norm_min.copy()
        self.normalization_state_max = comp.norm_max.copy()
        self.observation_space = Box(low=-1.0, high=1.0, shape=(10,))
        self.reward_range = (-3000.0, 23000.0)  # as declared by the reference (atc_gym.py:115)

    # -- backend ---------------------------------------------------------------------------------------------------
    @property
    def _vec(self):
        """The batched backend (1 env x 1 aircraft).  Every access from outside step() first drains the stream: step()
        returns as soon as the result packet has arrived, the kernel's trailing state stores may still be in flight."""
        self._settle()
        return self._backend

    @staticmethod
    def _make_backend(sim_parameters, scenario, device):
        from atc_hip.vec_env import AtcVecEnv
        # keep_active: the reference's aircraft is never handed over — after a win it keeps flying (and can win again) if
        # the caller steps on without reset (atc_gym.py:128-192 has no inactive state)
        return AtcVecEnv(1, 1, sim_parameters=sim_parameters, scenario=scenario, device=device, auto_reset=False,
                         spawn="lattice", want_raw_obs=True, host_mapped="io", keep_active=True, want_packet=True)

    @property
    def last_action(self):
        """atc_gym.py:86,311 — lives on the device next to the aircraft state."""
        return self._vec.get_last_action(0, 0)

    @last_action.setter
    def last_action(self, value):
        self._vec.set_last_action(0, 0, value)

    # -- gym.Env ----------------------------------------------------------------------------------------------------
    def seed(self, seed=None):
        """atc_gym.py:117-126"""
        if seed is None:
            seed = int(np.random.SeedSequence().entropy % (2 ** 31))
        self.np_random = np.random.RandomState(seed % (2 ** 32))
        random.seed(seed)
        return [seed]

    def step(self, action):
        """atc_gym.py:128-192 — one launch of the HIP step kernel + the reference's Python-side bookkeeping."""
        try:
            self._act_np[:] = action               # the common case: 3 values, converted to fp32 on assignment
        except (ValueError, TypeError):
            self._act_np[:] = np.asarray(action, dtype=np.float32).reshape(3)
        # model.py:123: Airplane.step first remembers where the aircraft IS (read by render() only), then moves it
        if self._pos_now is None:
            self._settle()
            px_py = self._pos_rec[0, :2].tolist()
            self._pos_now = (int(px_py[0]), int(px_py[1]))
        px, py = self._pos_now
        self._airplane.position_history.append((self._pos_origin[0] + px * self._pos_inv,
                                                self._pos_origin[1] + py * self._pos_inv))
        state_out, raw, rew, dn, flags, self.timesteps, self.actions_taken = self._launch_and_fetch()
        self.done = False
        # one append per terminal cause, in the reference's order (atc_gym.py:151,158,165)
        if flags & (L.F_BELOW_MVA | L.F_OUTSIDE):
            self._win_buffer.append(0)
            self.done = True
        if flags & L.F_WON:
            self._win_buffer.append(1)
            self.done = True
        if flags & L.F_TIMEOUT:
            self.done = True
        if flags & (L.F_INVALID_V | L.F_INVALID_H):  # atc_gym.py:314
            for bit, idx in ((L.F_INVALID_V, 0), (L.F_INVALID_H, 1)):
                if flags & bit:
                    print("Warning invalid action: %d for index: %d" % (self._denormalized_action(self._act_np[idx], idx), idx))
        assert self.done == bool(dn)
        self.state = raw
        self._update_metrics(rew)
        return state_out, rew, self.done, {"original_state": self.state}

    def _launch_and_fetch(self):
        """One launch of the step kernel on host-mapped buffers, one stream synchronisation, results read in place."""
        if self._seq >= 0x7ffffff0:   # sequence numbers start over (every 2^31 steps): through a stopped server
            self._settle()
            self._seq = 0
        self._seq = seq = self._seq + 1
        tight = _perf_counter() - self._t_done < _TIGHT_GAP_S
        if self._persistent and (tight or self._serving):
            # the persistent step server: write {action, seq} into the mailbox, poll the result packet — no launch per step
            rc = -5
            for attempt in range(2):
                if not self._serving:
                    if not tight or not _server_slot_free():
                        break                      # enough resident kernels in this process: this step goes by a launch
                    torch = self._backend.torch
                    if self._srv_stream is None:
                        self._srv_stream = torch.cuda.Stream(device=self._backend.device)
                    # whatever was queued for this env before (reset kernels, placed state) comes first
                    self._srv_stream.wait_stream(self._current_stream(self._backend.device))
                    self._check(self._serve_start(self._srv_stream.cuda_stream, seq - 1))
                    self._serving = True
                    _SERVING[id(self)] = weakref.ref(self)
                rc = self._serve_step(seq)
                if rc != -4:
                    break
                # the server had left (its lease ran out) before it saw this command: its kernel has ended; start it again if
                # the steps still follow each other closely, else take this one by a launch
                self._srv_stream.synchronize()
                self._serving = False
                _SERVING.pop(id(self), None)
            if rc == 0:
                w = self._payload_f.copy()
                iw = self._payload_i
                fd = int(iw[21])
                self._pos_now = (int(iw[24]), int(iw[25]))
                self._t_done = _perf_counter()
                return (w[0:10], w[10:20], float(w[20]), bool(fd >> 16), fd & 0xffff, int(iw[22]), int(iw[23]))
            if rc == -4:
                rc = -5
            if rc == -3:
                # no answer in 2 s (never expected): stop serving for good and take this step by a launch; the command was not
                # executed if the server's last sequence number is still the previous one
                self._persistent = False
                self._settle()
                if int(self._mailbox[5]) == seq:
                    raise RuntimeError("AtcGym: the step server executed step %d but its result packet never arrived" % seq)
            elif rc != -5:
                self._check(rc)
        self._outstanding = True
        rc = self._step_packet(self._raw_stream(), seq)   # ~10 us of kernel + host link, polled inside the library
        if rc == -3:                           # never expected: fall back to the blocking wait
            self._current_stream(self._backend.device).synchronize()
            self._outstanding = False
            assert (self._pkt_tags == seq).all()
            self._payload_i[:] = self._pkt_i[:, :3].reshape(27)
        elif rc:
            self._check(rc)
        w = self._payload_f.copy()
        iw = self._payload_i
        fd = int(iw[21])
        self._pos_now = (int(iw[24]), int(iw[25]))
        self._t_done = _perf_counter()
        return (w[0:10], w[10:20], float(w[20]), bool(fd >> 16), fd & 0xffff, int(iw[22]), int(iw[23]))

    def _settle(self):
        """Drains the stream before anything but step() looks at (or writes) memory the last kernel may still be writing:
        the result packet arrives before the kernel's trailing state stores."""
        if self._serving:
            # quit + stream synchronisation: the env's state is back in memory, the stream free for other launches
            self._check(self._serve_stop(self._srv_stream.cuda_stream))
            self._serving = False
            _SERVING.pop(id(self), None)
            self._outstanding = False
        if self._outstanding:
            self._current_stream(self._backend.device).synchronize()
            self._outstanding = False

    def _update_metrics(self, reward):
        """atc_gym.py:194-197"""
        self.last_reward = reward
        self.total_reward += reward
        self.actions_per_timestep = self.actions_taken / self.timesteps

    def _denormalized_action(self, action, index):
        """atc_gym.py:318-335 (host copy used only for the warning text; the kernel applies its own)."""
        f, o = self.normalization_action_factor, self.normalization_action_offset
        if self._sim_parameters.discrete_action_space:
            return action * f[index] + o[index]
        return action * f[index] / 2 + f[index] / 2 + o[index]

    def reset(self):
        """atc_gym.py:337-365.  Draws the entry point with Python's `random` in the reference's order (choice, choice,
        randint: atc_gym.py:346-348, model.py:52) so a seeded run picks the same entries, then places the aircraft on the
        device."""
        self.done = False
        vec = self._vec
        entry_point = random.choice(self._scenario.entrypoints)
        level = random.choice(entry_point.levels)
        plane_id = random.randint(0, 32767)
        vec.reset()
        vec.set_state(0, 0, entry_point.x, entry_point.y, level * 100, entry_point.phi, 250)
        self._airplane.id = plane_id
        self._airplane.position_history = []
        self._pos_now = None
        self.state = vec.observe().reshape(-1).numpy().astype(np.float32)  # host-mapped: synchronised, copied here
        self.total_reward = 0
        self.last_reward = 0
        self._actions_ignoring_resets += self.actions_taken
        self.actions_taken = 0
        self.timesteps = 0
        self._episodes_run += 1
        if len(self._win_buffer) < self._win_buffer_size:
            self._win_buffer.append(0)
        self.winning_ratio = self.winning_ratio + 1 / self._win_buffer_size * \
            (self._win_buffer[-1] - self._win_buffer.pop(0))
        return self.state

    def render(self, mode='human'):
        """The pyglet window of the reference (atc_gym.py:367-552) is out of scope; `rgb_array` is served by
        the headless numpy renderer (atc_hip/render.py: the reference's geometry — pinned on tests/golden/g10 — with the label
        text in a built-in bitmap font) so recorders keep working, `human` is a no-op."""
        if mode == 'rgb_array':
            from atc_hip import render
            return render.rgb_array(self._vec, env=0, history=self._airplane.position_history,
                                    total_reward=self.total_reward, last_reward=self.last_reward)
        return None

    def close(self):
        if getattr(self, "_backend", None) is not None:
            self._settle()
            self._backend.close()
            self._backend = None
