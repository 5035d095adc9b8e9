"""ctypes wrapper of the CPU parity oracle (oracle/liboracle_atc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; it is the checker, never
the thing shipped or measured as the product.  See oracle/atc_oracle_impl.h for what it restates and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle_atc.so")

M_REWARD_SHAPING, M_NORMALIZE, M_DISCRETE, M_AUTO_RESET, M_RANDOM_ENTRY, M_KEEP_ACTIVE = 1, 2, 4, 8, 16, 32
F_PHI_LIMIT = 1 << 9


class Params(C.Structure):
    """Mirror of atc_params_t (include/atc_step.h)."""
    _fields_ = [("dt", C.c_double), ("timestep_limit", C.c_int32), ("mode", C.c_uint32), ("seed", C.c_uint64),
                ("sep_nm", C.c_float), ("sep_ft", C.c_float), ("conflict_reward", C.c_float), ("reserved0", C.c_uint32),
                ("reserved1", C.c_float), ("reserved2", C.c_uint32)]


def make_params(dt=1.0, shaping=True, normalize=True, discrete=False, auto_reset=False, random_entry=False, seed=0,
                timestep_limit=6000, sep_nm=3.0, sep_ft=1000.0, conflict_reward=-200.0, keep_active=False):
    mode = (M_REWARD_SHAPING if shaping else 0) | (M_NORMALIZE if normalize else 0) | (M_DISCRETE if discrete else 0) | \
           (M_AUTO_RESET if auto_reset else 0) | (M_RANDOM_ENTRY if random_entry else 0) | \
           (M_KEEP_ACTIVE if keep_active else 0)
    return Params(float(dt), timestep_limit, mode, seed, sep_nm, sep_ft, conflict_reward, 0, 0.0, 0)


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB_PATH)
            for f in ("atc_oracle.c", "atc_oracle_impl.h", "../include/atc_step.h")):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.atc_oracle_set_threads(1)  # default: scalar single-thread restatement
    return _lib


def set_threads(n):
    """OpenMP threads used by atc_oracle_step_* (1 = the scalar single-core port)."""
    lib().atc_oracle_set_threads(int(n))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    """B envs x N aircraft on host arrays, stepping through atc_oracle_step_{f64,f32}."""

    def __init__(self, compiled, B=1, N=1, params=None, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.sfx = "_f64" if self.dtype == np.float64 else "_f32"
        self.blob = np.ascontiguousarray(compiled.blob64 if self.dtype == np.float64 else compiled.blob32)
        self.B, self.N = B, N
        self.params = params or make_params()
        r = self.dtype
        BN = B * N
        # positions: float64 in the reference-faithful instantiation, 32-bit fixed point on the sector's position grid in
        # the fp32 one (include/atc_step.h "Aircraft positions"); `.x` / `.y` give nautical miles either way
        self.fixed = self.dtype == np.float32
        self.pos_origin, self.pos_k = compiled.pos_origin, compiled.pos_k
        pdt = np.int32 if self.fixed else np.float64
        self.px, self.py = np.zeros(BN, pdt), np.zeros(BN, pdt)
        # speed / heading: float64 (reference) | 32-bit fixed point (fp32 spec, include/atc_step.h ABI 18: kt = v_fix 2^-23 (unsigned),
        # deg = 180 + phi_fix 2^-23); `.phi` / `.v` give degrees / knots either way, `.phi_fix` / `.v_fix` the stored counts.
        # The altitude is float64 in BOTH instantiations (ABI 20: the reference's own arithmetic).
        # last_act = the last accepted targets: reference instantiation [3, BN] float64 (v, h, phi); fp32 spec [BN, 4] int32 words in
        # the device's record layout (v_fix, phi_fix, the altitude target's float64 in words 2..3) — compares directly with
        # AtcVecEnv.last_act
        self.h = np.zeros(BN, np.float64)
        self._phi, self._v = (np.zeros(BN, np.int32 if self.fixed else r) for _ in range(2))
        self.last_act = np.zeros((BN, 4), np.int32) if self.fixed else np.zeros((3, BN), r)
        self.timesteps = np.zeros(B, np.int32)
        self.actions_taken = np.zeros(B, np.int32)
        self.total_reward = np.zeros(B, r)
        self.active_mask = np.zeros(B, np.uint64)
        self.win_bits = np.zeros(B, np.uint32)
        self.episodes = np.zeros(B, np.int32)
        self.ep_return = np.zeros(B, r)
        self.ep_length = np.zeros(B, np.int32)
        self.ep_actions = np.zeros(B, np.int32)
        self.obs = np.zeros((B, N, 10), np.float32)
        self.raw_obs = np.zeros((B, N, 10), np.float32)
        self.reward = np.zeros(B, r)
        self.ac_reward = np.zeros((B, N), r)
        self.done = np.zeros(B, np.uint8)
        self.flags = np.zeros((B, N), np.uint16)
        self.min_sep = np.zeros(B, r)
        self.term_obs = np.zeros((B, N, 10), np.float32)
        self.mva = np.zeros((B, N), np.int32)
        # exact 64-bit heading counts / last heading target of WIDE aircraft (include/atc_step.h, atc_state_t.phi_wide); the
        # float64 instantiation (the reference as it is) never touches it
        self.phi_wide = np.zeros((BN, 2), np.int64)
        self._st = (C.c_void_p * 16)(*[_ptr(a) for a in (
            self.px, self.py, self.h, self._phi, self._v, self.last_act, self.timesteps, self.actions_taken,
            self.total_reward, self.active_mask, self.win_bits, self.episodes, self.ep_return, self.ep_length,
            self.ep_actions, self.phi_wide)])
        self._out = (C.c_void_p * 9)(*[_ptr(a) for a in (
            self.obs, self.raw_obs, self.reward, self.ac_reward, self.done, self.flags, self.min_sep, self.term_obs,
            self.mva)])
        self.reset(first=True)

    def reset(self, mask=None, first=False):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        fn = getattr(lib(), "atc_oracle_reset" + self.sfx)
        rc = fn(_ptr(self.blob), self.B, self.N, self._st, _ptr(m), _ptr(self.obs), C.byref(self.params), int(first))
        assert rc == 0
        return self.obs.copy()

    def _nm(self, p, axis):
        if not self.fixed:
            return p
        return p.astype(np.float64) * 2.0 ** -self.pos_k + self.pos_origin[axis]

    @property
    def x(self):
        """positions in nautical miles (float64 view of whatever the instantiation stores)"""
        return self._nm(self.px, 0)

    @property
    def y(self):
        return self._nm(self.py, 1)

    def _to_pos(self, v, axis):
        if not self.fixed:
            return v
        c = np.rint((float(v) - self.pos_origin[axis]) * 2.0 ** self.pos_k)
        return np.int32(min(max(c, -2.0 ** 31), 2.0 ** 31 - 1))

    V_OFFSET, V_Q, PHI_OFFSET, PHI_Q = 0.0, 2.0 ** 23, 180.0, 2.0 ** 23   # include/atc_step.h: ATC_V_FIX_* / ATC_PHI_FIX_*

    @staticmethod
    def _fix(value, offset, q, unsigned=False):
        c = np.rint((float(value) - offset) * q)
        if unsigned:   # the speed's counts are unsigned 32-bit, stored in the same int32 words
            return np.uint32(min(max(c, 0.0), 2.0 ** 32 - 1)).astype(np.int32)
        return np.int32(min(max(c, -2.0 ** 31), 2.0 ** 31 - 1))

    I32_MIN, I32_MAX = -2 ** 31, 2 ** 31 - 1

    @property
    def phi_counts(self):
        """exact 64-bit heading counts (the 32-bit field, or phi_wide where that is saturated: include/atc_step.h, ABI 19)"""
        assert self.fixed
        p = self._phi.astype(np.int64)
        wide = (self._phi == self.I32_MIN) | (self._phi == self.I32_MAX)
        return np.where(wide, self.phi_wide[:, 0], p)

    @property
    def phi(self):
        """headings in degrees (float64: exact for the fixed-point instantiation)"""
        return self.phi_counts.astype(np.float64) / self.PHI_Q + self.PHI_OFFSET if self.fixed else self._phi

    def _put_phi(self, field, i, col, deg):
        """stores a heading [deg] as (sat32 counts, exact counts in phi_wide[:, col] when saturated)"""
        P = int(np.rint((float(deg) - self.PHI_OFFSET) * self.PHI_Q))
        P = min(max(P, -2 ** 52), 2 ** 52)
        field[i] = min(max(P, self.I32_MIN), self.I32_MAX)
        if field[i] in (self.I32_MIN, self.I32_MAX):
            self.phi_wide[i, col] = P

    @property
    def v(self):
        return self._v.view(np.uint32).astype(np.float64) / self.V_Q if self.fixed else self._v

    @property
    def phi_fix(self):
        assert self.fixed
        return self._phi

    @property
    def v_fix(self):
        assert self.fixed
        return self._v

    def set_state(self, e, k, x, y, h, phi, v):
        i = e * self.N + k
        self.px[i], self.py[i] = self._to_pos(x, 0), self._to_pos(y, 1)
        self.h[i] = h
        if self.fixed:
            self._put_phi(self._phi, i, 0, phi)
            self._v[i] = self._fix(v, self.V_OFFSET, self.V_Q, True)
        else:
            self._phi[i], self._v[i] = phi, v

    def set_last_action(self, e, k, value):
        """AtcGym.last_action (atc_gym.py:86,311) of one aircraft from [v, h, phi] in knots / feet / degrees."""
        i = e * self.N + k
        if self.fixed:
            self.last_act[i, 0] = self._fix(value[0], self.V_OFFSET, self.V_Q, True)
            self._put_phi(self.last_act[:, 1], i, 1, value[2])
            self.last_act[i, 2:4] = np.array([value[1]], np.float64).view(np.int32)
        else:
            self.last_act[:, i] = value

    def get_last_action(self, e, k):
        i = e * self.N + k
        if not self.fixed:
            return [float(c) for c in self.last_act[:, i]]
        lp = int(self.last_act[i, 1])
        if lp in (self.I32_MIN, self.I32_MAX):
            lp = int(self.phi_wide[i, 1])
        return [float(np.uint32(self.last_act[i, 0])) / self.V_Q, float(self.last_act[i, 2:4].copy().view(np.float64)[0]),
                float(lp) / self.PHI_Q + self.PHI_OFFSET]

    def step(self, actions):
        a = np.ascontiguousarray(np.asarray(actions, dtype=self.dtype).reshape(self.B * self.N * 3))
        fn = getattr(lib(), "atc_oracle_step" + self.sfx)
        rc = fn(_ptr(self.blob), self.B, self.N, self._st, _ptr(a), self._out, C.byref(self.params))
        assert rc == 0
        return self.obs, self.reward, self.done, self.flags


class OracleQueries:
    def __init__(self, compiled, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.sfx = "_f64" if self.dtype == np.float64 else "_f32"
        self.blob = np.ascontiguousarray(compiled.blob64 if self.dtype == np.float64 else compiled.blob32)

    def _a(self, v):
        return np.ascontiguousarray(np.asarray(v, dtype=self.dtype).ravel())

    def mva(self, x, y):
        x, y = self._a(x), self._a(y)
        out = np.zeros(len(x), np.int32)
        getattr(lib(), "atc_oracle_query_mva" + self.sfx)(_ptr(self.blob), len(x), _ptr(x), _ptr(y), _ptr(out))
        return out

    def corridor(self, x, y, h, phi, angle_only=False):
        x, y, h, phi = self._a(x), self._a(y), self._a(h), self._a(phi)
        out = np.zeros(len(x), np.uint8)
        getattr(lib(), "atc_oracle_query_corridor" + self.sfx)(_ptr(self.blob), len(x), _ptr(x), _ptr(y), _ptr(h),
                                                                _ptr(phi), int(angle_only), _ptr(out))
        return out

    def shaping(self, d_faf, phi_rel_faf, phi_plane, h, on_gp):
        args = [self._a(v) for v in (d_faf, phi_rel_faf, phi_plane, h, on_gp)]
        out = np.zeros((len(args[0]), 3), self.dtype)
        getattr(lib(), "atc_oracle_query_shaping" + self.sfx)(_ptr(self.blob), len(args[0]), *[_ptr(a) for a in args],
                                                               _ptr(out))
        return out

    def relative_angle(self, a1, a2):
        fn = getattr(lib(), "atc_oracle_relative_angle" + self.sfx)
        ct = C.c_double if self.dtype == np.float64 else C.c_float
        fn.restype = ct
        fn.argtypes = [ct, ct]
        return np.array([fn(float(a), float(b)) for a, b in zip(np.ravel(a1), np.ravel(a2))])

    def sigmoid(self, d, dmax):
        fn = getattr(lib(), "atc_oracle_sigmoid" + self.sfx)
        ct = C.c_double if self.dtype == np.float64 else C.c_float
        fn.restype = ct
        fn.argtypes = [ct, ct]
        return np.array([fn(float(a), float(dmax)) for a in np.ravel(d)])
