"""GPU edge cases of the C-ABI / AtcVecEnv: ragged shapes, degenerate sizes, hostile inputs, argument errors."""
import ctypes as C

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("B,N", [(1, 1), (1, 2), (3, 3), (5, 7), (257, 1), (63, 16), (65, 16), (2, 33), (1, 64), (3, 64)])
def test_ragged_shapes_match_oracle(B, N):
    """Batch sizes that do not fill a wavefront / workgroup, aircraft counts that are not powers of two (idle lanes in a
    group), single env, maximum N: every output must match the fp32 oracle."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    from oracle import oracle as O
    scn = scenarios.LOWWDense() if N > 54 else scenarios.LOWW(random_entrypoints=True)
    comp = scenarios.compile_scenario(scn, grid_cell=0.5)
    env = AtcVecEnv(B, N, scenario=scn, auto_reset=True, spawn="lattice", want_raw_obs=True, want_min_sep=True)
    orc = O.OracleEnv(comp, B, N, O.make_params(auto_reset=True), np.float32)
    rng = np.random.default_rng(B * 100 + N)
    half = 0.5 * comp.norm_max.astype(np.float64)
    for t in range(60):
        if t % 10 == 0:
            a = rng.uniform(-1.05, 1.05, (B, N, 3)).astype(np.float32)
        obs, rew, done, info = env.step(a)
        orc.step(a)
        assert np.array_equal(info["flags"].cpu().numpy().astype(np.uint32), orc.flags), t
        assert np.array_equal(done.cpu().numpy(), orc.done), t
        o = obs.cpu().numpy().reshape(B, N, 10)
        assert np.all(np.abs(o - orc.obs) <= 1e-5 * np.maximum(1.0, np.abs(orc.obs))), t
        assert np.all(np.abs(info["original_state"].cpu().numpy().reshape(B, N, 10) - orc.raw_obs) <= 1e-5 * half), t
        assert np.all(np.abs(rew.cpu().numpy() - orc.reward) <= 1e-5 * np.maximum(1.0, np.abs(orc.reward)) + 6e-8 * N * np.abs(orc.ac_reward).sum(1)), t
        ms = info["min_separation"].cpu().numpy()
        assert np.all(np.abs(ms - orc.min_sep) <= 1e-5 * np.maximum(1.0, np.abs(orc.min_sep))), t
    assert np.array_equal(env.timesteps.cpu().numpy(), orc.timesteps)
    assert np.array_equal(env.active_mask.cpu().numpy().astype(np.uint64), orc.active_mask)
    env.close()


def test_hostile_actions_do_not_break_the_batch():
    """NaN / inf / huge action components poison at most their own env: the launch succeeds, other envs are unaffected and
    match a run without the hostile rows."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWW(random_entrypoints=True)
    B, N = 256, 4
    env = AtcVecEnv(B, N, scenario=scn, auto_reset=True)
    ref = AtcVecEnv(B, N, scenario=scn, auto_reset=True)
    rng = np.random.default_rng(0)
    bad_rows = [3, 77, 200]
    for t in range(40):
        a = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
        b = a.copy()
        b[3, 1, :] = np.nan
        b[77, 0, 2] = np.inf
        b[200, 2, :] = [1e30, -1e30, 1e38]
        obs, rew, done, info = env.step(b)
        o2, r2, d2, i2 = ref.step(a)
        good = np.ones(B, bool)
        good[bad_rows] = False
        g = torch.as_tensor(good).cuda()
        assert torch.equal(obs[g], o2[g]) and torch.equal(rew[g], r2[g]) and torch.equal(info["flags"][g], i2["flags"][g])
    torch.cuda.synchronize()
    # invalid (out of range / NaN) speed and altitude targets are absorbed as flags, never raised (atc_gym.py:312-315)
    fl = info["flags"].cpu().numpy()
    assert fl[200, 2] & (H.F_INVALID_V | H.F_INVALID_H)
    env.close()
    ref.close()


def test_step_server_c_abi_directly():
    """atc_serve_start / atc_serve_step / atc_serve_stop through ctypes, no AtcGym in between: argument errors are codes, a served
    step equals atc_step on a twin env bit for bit, the lease ends the server (state 2) and a command it never saw is -4, quit ends
    it at once (state 3) and leaves the state in memory."""
    import ctypes as C
    import time
    torch = _torch()
    from atc_hip import lib
    from atc_hip.vec_env import AtcVecEnv
    Lb = lib.load()
    mk = lambda: AtcVecEnv(1, 1, auto_reset=False, spawn="lattice", want_raw_obs=True, host_mapped="io", keep_active=True, want_packet=True)  # noqa: E731
    env, twin = mk(), mk()
    mb = torch.zeros(16, dtype=torch.int32).pin_memory()
    act = torch.tensor([0.2, -0.4, 0.6], dtype=torch.float32).pin_memory()
    payload = np.zeros(27, np.int32)
    q = torch.cuda.Stream()
    args = lambda m=mb: (env.sector.handle, C.byref(env._state), C.byref(env._out), C.byref(env.params), m.data_ptr())  # noqa: E731
    assert Lb.atc_serve_start(*args()[:4], mb.data_ptr() + 4, 0, 50000, q.cuda_stream) == -1 and b"64-byte" in Lb.atc_last_error()
    no_pkt = lib.AtcOut(*[getattr(env._out, n) if n != "packet" else None for n in lib.OUT_FIELDS])
    assert Lb.atc_serve_start(env.sector.handle, C.byref(env._state), C.byref(no_pkt), C.byref(env.params), mb.data_ptr(), 0, 50000, q.cuda_stream) == -1
    assert Lb.atc_serve_step(mb.data_ptr(), act.data_ptr(), 0xffffffff, env.packet.data_ptr(), payload.ctypes.data, 1000) == -1
    torch.cuda.synchronize()
    # (While a server is resident nothing else is submitted to the GPU here: a stream that happens to share the server's hardware
    # queue would wait for the whole lease — include/atc_step.h "CHOOSING THE LEASE".  The twin steps afterwards.)
    def twin_steps(n):
        out = []
        for _ in range(n):
            o, r, d, info = twin.step(act.numpy().reshape(1, 1, 3))
            out.append((o.numpy().reshape(-1).copy(), float(r[0])))
        return out

    def same_state():
        return (torch.equal(env.ac.cpu(), twin.ac.cpu()) and torch.equal(env.alt.cpu(), twin.alt.cpu()) and torch.equal(env.env.cpu(), twin.env.cpu())
                and torch.equal(env.last_act.cpu(), twin.last_act.cpu()))
    assert Lb.atc_serve_start(*args(), 0, 2000000, q.cuda_stream) == 0
    served = []
    for seq in range(1, 41):
        assert Lb.atc_serve_step(mb.data_ptr(), act.data_ptr(), seq, env.packet.data_ptr(), payload.ctypes.data, 2000000) == 0
        assert payload[22] == seq
        served.append((payload[:10].view(np.float32).copy(), float(payload[20:21].view(np.float32)[0])))
    assert int(mb[4]) == 1 and int(mb[3]) == 40
    assert Lb.atc_serve_stop(mb.data_ptr(), q.cuda_stream) == 0             # quit: at once, the state is in memory
    assert int(mb[4]) == 3 and int(mb[5]) == 40
    for (so, sr), (to, tr) in zip(served, twin_steps(40)):
        assert np.array_equal(so, to) and sr == tr
    assert same_state()
    assert Lb.atc_serve_start(*args(), 40, 50000, q.cuda_stream) == 0        # a 50 ms lease this time
    assert Lb.atc_serve_step(mb.data_ptr(), act.data_ptr(), 41, env.packet.data_ptr(), payload.ctypes.data, 2000000) == 0
    time.sleep(0.25)
    assert int(mb[4]) == 2 and int(mb[5]) == 41                             # left by itself
    q.synchronize()
    twin_steps(1)
    assert same_state()
    assert Lb.atc_serve_step(mb.data_ptr(), act.data_ptr(), 42, env.packet.data_ptr(), payload.ctypes.data, 2000000) == -4   # nobody there
    assert Lb.atc_serve_start(*args(), 41, 2000000, q.cuda_stream) == 0
    assert Lb.atc_serve_step(mb.data_ptr(), act.data_ptr(), 42, env.packet.data_ptr(), payload.ctypes.data, 2000000) == 0
    assert Lb.atc_serve_stop(mb.data_ptr(), q.cuda_stream) == 0
    assert int(mb[4]) == 3 and int(mb[5]) == 42
    twin_steps(1)
    assert same_state()
    env.close()
    twin.close()


def test_argument_errors_are_reported_not_raised_from_c():
    torch = _torch()
    from atc_hip import lib
    from atc_hip.vec_env import AtcVecEnv
    env = AtcVecEnv(4, 2)
    L = lib.load()
    a = torch.zeros(4 * 2 * 3, device="cuda")
    st, out, p = env._state, env._out, env.params
    stream = env._stream()
    assert L.atc_step(env.sector.handle, 0, 2, C.byref(st), a.data_ptr(), C.byref(out), C.byref(p), stream) == -1
    assert b"B >= 1" in L.atc_last_error()
    assert L.atc_step(env.sector.handle, 4, 65, C.byref(st), a.data_ptr(), C.byref(out), C.byref(p), stream) == -1
    assert L.atc_step(None, 4, 2, C.byref(st), a.data_ptr(), C.byref(out), C.byref(p), stream) == -1
    assert L.atc_step(env.sector.handle, 4, 2, C.byref(st), None, C.byref(out), C.byref(p), stream) == -1
    assert L.atc_rollout(env.sector.handle, 4, 2, 0, C.byref(st), a.data_ptr(), C.byref(out), C.byref(p), stream) == -1
    # a partial last action block (T not a multiple of hold) is refused, not read (round-3 review, weak #6)
    assert L.atc_rollout_hold(env.sector.handle, 4, 2, 10, 4, C.byref(st), a.data_ptr(), C.byref(out), C.byref(p), stream) == -1
    assert b"multiple of hold" in L.atc_last_error()
    assert L.atc_rollout_hold(env.sector.handle, 4, 2, 3, 0, C.byref(st), a.data_ptr(), C.byref(out), C.byref(p), stream) == -1
    bad = lib.AtcState(st.ac, None, st.last_act, st.env, st.stats, st.phi_wide)
    assert L.atc_step(env.sector.handle, 4, 2, C.byref(bad), a.data_ptr(), C.byref(out), C.byref(p), stream) == -1
    assert b"null" in L.atc_last_error()
    too_big = 2 ** 27  # 2^27 x 64 aircraft x 40 B >> 4 GiB
    assert L.atc_step(env.sector.handle, too_big, 64, C.byref(st), a.data_ptr(), C.byref(out), C.byref(p), stream) == -1
    assert b"split the batch" in L.atc_last_error()
    # a blob of the wrong version is refused
    blob = env.compiled.blob32.copy()
    blob[0] = 1.0
    h = C.c_void_p()
    assert L.atc_scenario_create(blob.ctypes.data_as(C.c_void_p), blob.size, 0, C.byref(h)) == -1
    with pytest.raises(ValueError):
        env.step(np.zeros((4, 2, 2), np.float32))
    with pytest.raises(ValueError):
        AtcVecEnv(4, 65)
    # still healthy afterwards
    obs, rew, done, info = env.step(np.zeros((4, 2, 3), np.float32))
    assert bool(torch.isfinite(obs).all())
    env.close()


def test_masked_reset_only_touches_selected_envs():
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    env = AtcVecEnv(8, 2, auto_reset=False)
    a = np.random.default_rng(0).uniform(-1, 1, (8, 2, 3)).astype(np.float32)
    for _ in range(5):
        env.step(a)
    before = (env.ac.clone(), env.last_act.clone(), env.env.clone(), env.alt.clone())
    mask = np.array([0, 1, 0, 0, 1, 0, 0, 1], np.uint8)
    env.reset(mask=mask)
    m = torch.as_tensor(mask.astype(bool)).cuda()
    assert torch.equal(env.env[~m], before[2][~m])
    assert bool((env.timesteps[m] == 0).all()) and bool((env.timesteps[~m] == 5).all())
    mm = m.repeat_interleave(2)
    assert torch.equal(env.ac[~mm], before[0][~mm]) and not torch.equal(env.ac[mm], before[0][mm]) and torch.equal(env.alt[~mm], before[3][~mm])
    assert bool((env.episodes[m] == 2).all()) and bool((env.episodes[~m] == 1).all())
    # last_action survives a reset (atc_gym.py:86 is only executed in __init__)
    assert torch.equal(env.last_act, before[1]) and bool((env.last_act[:, [0, 1, 3]] != 0).all())
    env.close()


def test_streams_and_graph_capture():
    """The C-ABI launches on the caller's stream: works on a side stream and inside a captured HIP graph."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    env = AtcVecEnv(1024, 16, auto_reset=True)
    ref = AtcVecEnv(1024, 16, auto_reset=True)
    a = (torch.rand((1024, 16, 3), device="cuda") * 2 - 1).contiguous()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            env.step(a)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        ref.step(a)
    assert torch.equal(env.obs, ref.obs) and torch.equal(env.ac, ref.ac) and torch.equal(env.alt, ref.alt)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            env.step(a)
    for _ in range(2):
        g.replay()
    for _ in range(12):   # capture itself does not execute; 4 warm launches are not part of it
        ref.step(a)
    # graph: capture (0 executed) + 2 replays x 4 = 8 steps; bring ref to the same count
    env2 = AtcVecEnv(1024, 16, auto_reset=True)
    for _ in range(3 + 8):
        env2.step(a)
    torch.cuda.synchronize()
    assert torch.equal(env.ac, env2.ac) and torch.equal(env.alt, env2.alt) and torch.equal(env.obs, env2.obs)
    env.close(); ref.close(); env2.close()


@pytest.mark.parametrize("N", [1, 4, 16, 64])
def test_fast_and_full_kernel_variants_agree(N):
    """k_step<W, false> (obs/reward/done/flags only) and k_step<W, true> (all optional outputs) must produce identical
    results; a rollout with optional outputs must equal single steps."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWWDense() if N == 64 else scenarios.LOWW(random_entrypoints=True)
    B = 300
    fast = AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=5, spawn="random")
    full = AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=5, spawn="random", want_raw_obs=True, want_ac_reward=True,
                     want_min_sep=True, want_term_obs=True)
    roll = AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=5, spawn="random")
    g = torch.Generator(device="cpu").manual_seed(N)
    T = 12
    for block in range(6):
        acts = (torch.rand((T, B, N, 3), generator=g) * 2.1 - 1.05).cuda()
        out = {k: torch.empty((T,) + tuple(shape), dtype=dt, device="cuda") for k, shape, dt in (
            ("obs", (B, N * 10), torch.float32), ("reward", (B,), torch.float32), ("done", (B,), torch.uint8),
            ("flags", (B, N), torch.int16), ("raw_obs", (B, N * 10), torch.float32), ("ac_reward", (B, N), torch.float32),
            ("min_sep", (B,), torch.float32), ("term_obs", (B, N * 10), torch.float32))}
        roll.rollout(acts, out=out)
        for t in range(T):
            o1, r1, d1, i1 = fast.step(acts[t])
            o2, r2, d2, i2 = full.step(acts[t])
            assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(i1["flags"], i2["flags"])
            assert torch.equal(out["obs"][t], o2) and torch.equal(out["reward"][t], r2) and torch.equal(out["flags"][t], i2["flags"])
            assert torch.equal(out["raw_obs"][t], i2["original_state"]) and torch.equal(out["ac_reward"][t], i2["aircraft_reward"])
            assert torch.equal(out["min_sep"][t], i2["min_separation"])
            dn = d2 != 0
            if bool(dn.any()):
                assert torch.equal(out["term_obs"][t][dn], i2["terminal_observation"][dn])
    for name in ("ac", "alt", "last_act", "env", "stats", "phi_wide"):
        assert torch.equal(getattr(fast, name), getattr(full, name)) and torch.equal(getattr(roll, name), getattr(full, name)), name
    for e in (fast, full, roll):
        e.close()


@pytest.mark.parametrize("N,T,hold", [(16, 20, 20), (16, 20, 5), (1, 12, 4), (64, 10, 10), (5, 9, 3)])
def test_rollout_with_held_actions_equals_single_steps(N, T, hold):
    """atc_rollout_hold (frame skip): T steps in one launch, every action block held for `hold` steps == the same steps
    launched one by one with the action repeated; outputs and the whole persistent state bit-identical."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWWDense() if N > 16 else scenarios.LOWW(random_entrypoints=True)
    B = 300
    one = AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=5)
    roll = AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=5)
    g = torch.Generator(device="cpu").manual_seed(N * 100 + T)
    for launch in range(4):
        blocks = (torch.rand((T // hold, B, N, 3), generator=g) * 2.1 - 1.05).cuda()
        out = roll.rollout(blocks, hold=hold)
        for t in range(T):
            o, r, d, info = one.step(blocks[t // hold])
            assert torch.equal(out["obs"][t], o) and torch.equal(out["reward"][t], r), (launch, t)
            assert torch.equal(out["done"][t], d) and torch.equal(out["flags"][t], info["flags"]), (launch, t)
    for name in ("ac", "alt", "last_act", "env", "stats", "phi_wide"):
        assert torch.equal(getattr(one, name), getattr(roll, name)), name
    assert N < 16 or int(one.episodes.sum()) > B
    one.close()
    roll.close()


def test_sub_batches_on_streams_equal_one_batch():
    """atc_step_multi: a batch stepped as 3 independent sub-batches on 3 streams (one foreign call per step, no join
    between steps) gives exactly the results of the same envs stepped as one batch; launch errors are reported."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv, make_multi_launcher
    from atc_hip import lib as binding
    from envs.atc import scenarios
    scn = scenarios.LOWW(random_entrypoints=True)
    N, sizes = 16, [700, 300, 1048]
    B = sum(sizes)
    kw = dict(scenario=scn, auto_reset=True, spawn="lattice")
    whole = AtcVecEnv(B, N, **kw)
    subs = [AtcVecEnv(b, N, **kw) for b in sizes]
    streams = [torch.cuda.Stream() for _ in sizes]
    g = torch.Generator(device="cpu").manual_seed(5)
    ring = [(torch.rand((B, N, 3), generator=g) * 2.1 - 1.05).cuda() for _ in range(3)]
    lo = [0, sizes[0], sizes[0] + sizes[1], B]
    calls = [make_multi_launcher(subs, [a[lo[i]:lo[i + 1]] for i in range(3)], streams) for a in ring]
    torch.cuda.synchronize()
    for t in range(120):
        calls[(t // 7) % 3]()          # runs ahead on the three streams, never joined inside the loop
    for t in range(120):
        whole.step(ring[(t // 7) % 3])
    torch.cuda.synchronize()
    for i, e in enumerate(subs):
        sl = slice(lo[i], lo[i + 1])
        assert torch.equal(e.obs, whole.obs[sl]) and torch.equal(e.reward, whole.reward[sl])
        assert torch.equal(e.done, whole.done[sl]) and torch.equal(e.flags, whole.flags[sl])
        assert torch.equal(e.env, whole.env[sl]) and torch.equal(e.stats, whole.stats[sl])
        asl = slice(lo[i] * N, lo[i + 1] * N)
        assert torch.equal(e.ac, whole.ac[asl]) and torch.equal(e.alt, whole.alt[asl])
        assert torch.equal(e.last_act, whole.last_act[asl])
    assert int(whole.episodes.sum()) > B
    # a bad call in the list is reported (second call has B = 0), the ones before it were issued
    bad = (binding.AtcStepCall * 2)(subs[0].step_call(ring[0][:sizes[0]])[0], subs[1].step_call(ring[0][lo[1]:lo[2]])[0])
    bad[1].B = 0
    assert binding.load().atc_step_multi(2, bad) == -1
    assert b"B >= 1" in binding.load().atc_last_error()
    torch.cuda.synchronize()
    for e in subs + [whole]:
        e.close()


@pytest.mark.parametrize("B,N", [(1, 1), (5, 3), (70, 16)])
def test_host_mapped_buffers_match_device_buffers(B, N):
    """Zero-copy mode (state, actions and outputs in pinned host memory mapped into the device, atc_host_mapped_ptr):
    bit-identical to the same env held in HBM, and the returned tensors are host tensors that are valid on return."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWW(random_entrypoints=True)
    kw = dict(scenario=scn, auto_reset=True, seed=11, spawn="random", want_raw_obs=True, want_term_obs=True)
    dev = AtcVecEnv(B, N, **kw)
    hst = AtcVecEnv(B, N, host_mapped=True, **kw)
    assert not hst.obs.is_cuda and hst.obs.is_pinned() and dev.obs.is_cuda
    assert torch.equal(dev.reset().cpu(), hst.reset())
    g = torch.Generator(device="cpu").manual_seed(B * 64 + N)
    pinned = torch.zeros((B, N, 3), dtype=torch.float32).pin_memory()
    for t in range(150):
        if t % 10 == 0:
            acts = torch.rand((B, N, 3), generator=g) * 2.1 - 1.05
        pinned.copy_(acts)
        o1, r1, d1, i1 = dev.step(acts.cuda())
        o2, r2, d2, i2 = hst.step(pinned if t % 2 else acts)  # pinned actions are read in place, others are uploaded
        assert torch.equal(o1.cpu(), o2) and torch.equal(r1.cpu(), r2) and torch.equal(d1.cpu(), d2)
        assert torch.equal(i1["flags"].cpu(), i2["flags"]) and torch.equal(i1["original_state"].cpu(), i2["original_state"])
    for name in ("ac", "alt", "last_act", "env", "stats", "phi_wide"):
        assert torch.equal(getattr(dev, name).cpu(), getattr(hst, name)), name
    assert int(hst.episodes.sum()) > 0
    # a pageable host pointer is rejected by the library, not dereferenced
    from atc_hip import lib as binding
    with pytest.raises(RuntimeError):
        binding.mapped_ptr(torch.zeros(4))
    dev.close()
    hst.close()


def test_huge_batch_uses_correct_offsets():
    """B*N*40 B just below the 4 GiB limit of the 32-bit per-lane byte offsets (1 048 576 envs x 64 aircraft = 2.7 GB of
    observations): the LAST envs of the huge batch (largest offsets) must equal the same envs run as a small batch — envs
    are independent and the slot lattice does not depend on the env index."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWWDense()
    B, N, tail = 1 << 20, 64, 96
    big = AtcVecEnv(B, N, scenario=scn, auto_reset=True, spawn="lattice")
    small = AtcVecEnv(tail, N, scenario=scn, auto_reset=True, spawn="lattice")
    g = torch.Generator(device="cuda").manual_seed(1)
    a_small = torch.rand((tail, N, 3), generator=g, device="cuda") * 2 - 1
    a_big = torch.zeros((B, N, 3), device="cuda")
    a_big[-tail:] = a_small
    for t in range(6):
        ob, rb, db, ib = big.step(a_big)
        os_, rs, ds, is_ = small.step(a_small)
        assert torch.equal(ob[-tail:], os_) and torch.equal(rb[-tail:], rs) and torch.equal(db[-tail:], ds)
        assert torch.equal(ib["flags"][-tail:], is_["flags"])
    assert torch.equal(big.ac[-tail * N:], small.ac) and torch.equal(big.alt[-tail * N:], small.alt) and torch.equal(big.env[-tail:], small.env)
    # and the very first envs are untouched by anything the tail did
    assert bool(torch.isfinite(ob[:4]).all())
    big.close()
    small.close()


@pytest.mark.parametrize("persistent", [True, False])
def test_atcgym_packet_polling_equals_synchronised_reads(persistent, monkeypatch):
    """AtcGym.step returns as soon as the self-validating result packet (atc_out_t.packet) has arrived in mapped memory —
    from the persistent step server (round 6, the default) or from a launch of its own (atc_step_packet) before the stream is
    drained.  30 000 steps with resets: every value it returned equals what the ordinary output buffers hold once the server has
    been stopped / the stream HAS been drained, and an env that does so after every step sees the same trajectory."""
    _torch()
    from envs.atc import atc_gym, scenarios
    import random
    monkeypatch.setattr(atc_gym, "_TIGHT_GAP_S", 1.0)   # serve every step (AtcGym serves only steps that follow each other within 50 us)
    random.seed(3)
    a_env = atc_gym.AtcGym(scenario=scenarios.LOWW(random_entrypoints=True), persistent=persistent)
    random.seed(3)
    b_env = atc_gym.AtcGym(scenario=scenarios.LOWW(random_entrypoints=True), persistent=persistent)
    rng = np.random.default_rng(8)
    n_done = 0
    fallbacks = 0
    for t in range(30000):
        if t % 20 == 0:
            act = rng.uniform(-1.02, 1.02, 3).astype(np.float32)
        oa, ra, da, ia = a_env.step(act)
        fallbacks += not (a_env._serving if persistent else a_env._outstanding)   # returned on the packet, not on a synchronisation
        ob, rb, db, ib = b_env.step(act)
        b_env._settle()
        assert np.array_equal(oa, ob) and ra == rb and da == db and np.array_equal(ia["original_state"], ib["original_state"])
        if t % 97 == 0:                                # the packet against the ordinary (drained) output buffers
            v = a_env._vec                             # (property: drains the stream)
            assert not a_env._outstanding and not a_env._serving
            assert np.array_equal(oa, v.obs.numpy().reshape(-1)) and ra == float(v.reward[0]) and da == bool(v.done[0])
            assert a_env.timesteps == int(v.timesteps[0]) and a_env.actions_taken == int(v.actions_taken[0])
            assert a_env._pos_now == (int(v.ac[0, 0]), int(v.ac[0, 1]))
        if da:
            n_done += 1
            random.seed(1000 + n_done)                 # both envs draw their entry point from Python's global RNG
            ra0 = a_env.reset()
            random.seed(1000 + n_done)
            assert np.array_equal(ra0, b_env.reset())
    assert n_done > 20
    assert fallbacks <= 3, fallbacks                   # (a stalled host thread may hit the time limit; the results are the same)
    a_env.close()
    b_env.close()


def _winning_state(scn):
    """an (x, y, h, phi) well inside the final-approach corridor and above its MVA (found with the query kernels)"""
    import numpy as np
    c = scn.runway.corridor
    fx, fy = float(np.ravel(c.faf)[0]), float(np.ravel(c.faf)[1])
    ix, iy = float(np.ravel(c.iaf)[0]), float(np.ravel(c.iaf)[1])
    for s in np.linspace(0.2, 0.8, 7):
        for off in (0.15, -0.15, 0.3, -0.3):
            x, y = fx + s * (ix - fx) + off, fy + s * (iy - fy)
            for dphi in (8.0, -8.0, 15.0, -15.0):
                phi = scn.runway.phi_to_runway + dphi
                h = float(scn.airspace.get_mva_heights([x], [y])[0]) + 150.0
                if all(scn.runway.inside_corridor(x + dx, y + dy, h, phi + dp)
                       for dx in (-0.1, 0.0, 0.1) for dy in (-0.1, 0.0, 0.1) for dp in (-3.0, 0.0, 3.0)):
                    return x, y, h, phi
    raise AssertionError("no robust winning state found")


@pytest.mark.parametrize("N,hold", [(16, 7), (1, 5), (64, 10), (5, 20)])
def test_held_actions_hint_changes_nothing(N, hold):
    """ATC_M_ACTIONS_HELD (step(..., held=True) / make_launcher(held=True)): a promise that the action block is repeated —
    outputs and the whole persistent state (incl. last_action and the action counters) are bit-identical to plain steps,
    also for envs that are auto-reset in the middle of a block while carrying aircraft handed over in an EARLIER block
    (their last_action record is older than the block: the one case in which a held step can count an action)."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWWDense() if N > 16 else scenarios.LOWW(random_entrypoints=N > 1)
    B = 400
    plain = AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=11)
    hint = AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=11)
    if N > 1:   # slot 0 of every second env starts inside the corridor: handed over in the first step of the first block
        x, y, h, phi = _winning_state(scn)
        for env in (plain, hint):
            for e in range(0, B, 2):
                env.set_state(e, 0, x, y, h, phi, 180.0)
    g = torch.Generator(device="cpu").manual_seed(N * 1000 + hold)
    stale_resets = 0
    handed_over = 0
    t = 0
    for block in range(8):
        a = (torch.rand((B, N, 3), generator=g) * 2.1 - 1.05).cuda()
        launch_first, launch_held = hint.make_launcher(a), hint.make_launcher(a, held=True)
        for k in range(hold):
            slot0_inactive = (plain.active_mask & 1) == 0
            o, r, d, info = plain.step(a)
            if block % 2:   # both ways of passing the promise
                o2, r2, d2, i2 = hint.step(a, held=k > 0)
            else:
                (launch_held if k > 0 else launch_first)()
                o2, r2, d2, i2 = hint.obs, hint.reward, hint.done, {"flags": hint.flags}
            assert torch.equal(o, o2) and torch.equal(r, r2) and torch.equal(d, d2), (block, k)
            assert torch.equal(info["flags"], i2["flags"]), (block, k)
            for name in ("ac", "alt", "last_act", "env", "stats", "phi_wide"):
                assert torch.equal(getattr(plain, name), getattr(hint, name)), (name, block, k)
            handed_over += int(slot0_inactive.sum()) if t == 1 else 0
            if k == 0:
                stale = slot0_inactive.clone()   # slot 0 handed over before this block began: its record is older than the block
            if k < hold - 1:                     # reset with a held step to follow
                stale_resets += int((d.bool() & stale).sum())
            stale &= ~d.bool()
            t += 1
    assert not (hint.params.mode & 64)          # step(held=True) leaves the parameters as they were
    if N > 1:
        assert handed_over >= B // 4, handed_over
        assert stale_resets > 0, "the stale-record case did not occur"
    plain.close()
    hint.close()


def test_check_held_catches_a_broken_promise():
    """AtcVecEnv(check_held=True): the debugging aid for callers adopting step(..., held=True)."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    env = AtcVecEnv(8, 4, check_held=True)
    a = torch.rand((8, 4, 3), device="cuda") * 2 - 1
    with pytest.raises(ValueError):
        env.step(a, held=True)            # nothing to repeat yet
    env.step(a)
    env.step(a.clone(), held=True)        # same values, another tensor: fine
    b = a.clone()
    b[3, 2, 1] += 0.25
    with pytest.raises(ValueError):
        env.step(b, held=True)
    env.step(b)
    env.step(b.cpu().numpy(), held=True)  # arrays count by value too
    env.close()


def test_held_hint_after_a_multi_step_launch():
    """The promise also holds across launch forms: a single step that repeats the LAST action block of the preceding
    atc_rollout_hold launch may carry ATC_M_ACTIONS_HELD."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWW(random_entrypoints=True)
    B, N = 300, 16
    a, b = (AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=9) for _ in range(2))
    g = torch.Generator(device="cpu").manual_seed(77)
    for rnd in range(6):
        blocks = (torch.rand((3, B, N, 3), generator=g) * 2.1 - 1.05).cuda()
        a.rollout(blocks, hold=4)
        b.rollout(blocks, hold=4)
        for k in range(3):
            ra = a.step(blocks[2])
            rb = b.step(blocks[2], held=True)
            assert all(torch.equal(x, y) for x, y in zip(ra[:3], rb[:3])), (rnd, k)
        for name in ("ac", "alt", "last_act", "env", "stats", "phi_wide"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (name, rnd)
    a.close()
    b.close()


def test_envs_with_different_parameters_interleaved_on_one_thread():
    """The library caches the host-evaluated step constants per thread, keyed by sector handle and the parameters they
    depend on: envs that differ in time step / action space / separation minima / time-step limit — over the SAME sector,
    stepped alternately from one thread, single steps and fused launches — must each keep getting their own constants."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import model, scenarios
    from oracle import oracle as O
    scn = scenarios.LOWW(random_entrypoints=True)
    comp = scenarios.compile_scenario(scn, grid_cell=0.25)
    B, N = 96, 16
    cfgs = [dict(dt=1.0, discrete=False, sep_nm=3.0, limit=6000), dict(dt=2.0, discrete=True, sep_nm=5.0, limit=6000),
            dict(dt=5.0, discrete=False, sep_nm=3.0, limit=30)]
    envs, orcs = [], []
    for c in cfgs:
        sp = model.SimParameters(c["dt"], discrete_action_space=c["discrete"])
        envs.append(AtcVecEnv(B, N, sim_parameters=sp, scenario=scn, auto_reset=True, spawn="lattice", seed=3, grid_cell=0.25,
                              sep_nm=c["sep_nm"], timestep_limit=c["limit"]))
        orcs.append(O.OracleEnv(comp, B, N, O.make_params(dt=c["dt"], discrete=c["discrete"], auto_reset=True, seed=3,
                                                           sep_nm=c["sep_nm"], timestep_limit=c["limit"]), np.float32))
    rng = np.random.default_rng(77)

    def draw(c):
        if c["discrete"]:
            return np.floor(rng.uniform(0, 1, (B, N, 3)) * np.array([20, 380, 360])).astype(np.float32)
        return rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    for rnd in range(12):
        for c, env, orc in zip(cfgs, envs, orcs):   # one call per env, round robin: every call meets another env's cache entry
            a = draw(c)
            if rnd % 3 == 2:
                out = env.rollout(torch.as_tensor(a[None]), hold=4)
                res = [(out["flags"][t], out["done"][t], out["reward"][t]) for t in range(4)]
            else:
                _, rew, done, info = env.step(a)
                res = [(info["flags"], done, rew)]
            for fl, done, rew in res:
                orc.step(a)
                assert np.array_equal(fl.cpu().numpy().astype(np.uint32), orc.flags), (c, rnd)
                assert np.array_equal(done.cpu().numpy(), orc.done), (c, rnd)
                assert np.all(np.abs(rew.cpu().numpy() - orc.reward) <= 1e-5 * np.maximum(1.0, np.abs(orc.reward)) + 6e-8 * N * np.abs(orc.ac_reward).sum(1)), (c, rnd)
    for env, orc in zip(envs, orcs):
        assert np.array_equal(env.ac[:, 0].cpu().numpy(), orc.px) and np.array_equal(env.h.cpu().numpy(), orc.h)
        assert np.array_equal(env.timesteps.cpu().numpy(), orc.timesteps)
        env.close()


@pytest.mark.parametrize("N", [1, 16, 64])
def test_all_valid_instantiation_equals_general_kernels(N):
    """A batch that is a whole number of workgroups of aircraft slots runs the ALL-VALID kernel instantiation (every validity flag a
    compile-time constant: csrc/atc_step.hip make_ids<W, ALLV>); one env more and the launch takes the general kernels.  The envs
    the two batches share must come out bit-identical — single steps (with the held-action hint) and atc_rollout_hold — and a
    batch with the optional outputs (FULL kernels, never all-valid) must agree with both."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWWDense() if N == 64 else scenarios.LOWW(random_entrypoints=N > 1)
    B = 2 * max(1, 256 // N) * (4 if N == 1 else 1)
    envs = [AtcVecEnv(B, N, scenario=scn, seed=9), AtcVecEnv(B + 1, N, scenario=scn, seed=9),
            AtcVecEnv(B, N, scenario=scn, seed=9, want_raw_obs=True)]
    g = torch.Generator(device="cpu").manual_seed(N)
    for t in range(120):
        if t % 20 == 0:
            a = (torch.rand((B + 1, N, 3), generator=g) * 2 - 1).cuda()
        res = [e.step(a[:e.B].contiguous(), held=t % 20 != 0) for e in envs]
        for o, r, d, info in res[1:]:
            assert torch.equal(res[0][0], o[:B]) and torch.equal(res[0][1], r[:B]) and torch.equal(res[0][2], d[:B])
            assert torch.equal(res[0][3]["flags"], info["flags"][:B])
    for j in range(3):
        blocks = (torch.rand((1, B + 1, N, 3), generator=g) * 2 - 1).cuda()
        outs = [e.rollout(blocks[:, :e.B].contiguous(), hold=20) for e in envs[:2]]
        for k in ("obs", "reward", "done", "flags"):
            assert torch.equal(outs[0][k], outs[1][k][:, :B]), k
    for name in ("ac", "alt", "last_act"):
        assert torch.equal(getattr(envs[0], name), getattr(envs[1], name)[:B * N]), name
    assert torch.equal(envs[0].env, envs[1].env[:B]) and torch.equal(envs[0].stats, envs[1].stats[:B])
    assert int(envs[0].episodes.sum()) > B      # episodes ended and restarted inside the comparison
    for e in envs:
        e.close()


@pytest.mark.parametrize("N,big", [(1, 131072 + 256), (16, 8192 + 256), (64, 2048 + 4)])
def test_latency_bound_instantiation_equals_the_throughput_kernels(N, big):
    """Multi-step launches of at most two wavefronts per SIMD run the LATENCY-BOUND instantiation (uniform floating-point terms in
    vector registers, csrc/atc_step.hip: LAT), bigger all-valid batches the throughput instantiation, other shapes the general
    kernels.  The envs three such batches share (same env indices, same seed: same spawn draws) must come out bit-identical over
    several atc_rollout_hold launches with resets inside."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWWDense() if N == 64 else scenarios.LOWW(random_entrypoints=N > 1)
    small = 2 * max(1, 256 // N) * (4 if N == 1 else 1)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    W = 1 << max(0, (N - 1).bit_length())
    assert (big * W + 63) // 64 > 2 * 4 * n_cu >= (small * W + 63) // 64, "batch sizes no longer straddle the LAT threshold"
    envs = [AtcVecEnv(small, N, scenario=scn, seed=5), AtcVecEnv(big, N, scenario=scn, seed=5), AtcVecEnv(big + 1, N, scenario=scn, seed=5)]
    g = torch.Generator(device="cpu").manual_seed(N)
    # every 7th aircraft of the shared envs starts at 430 deg and is told to turn on to 720 (a_phi = 3) for the whole run: its heading
    # leaves the 32-bit field in the second step and stays WIDE (ABI 19) — in all three kernel families
    turner = torch.arange(0, small * N, 7)
    saw_wide = False
    for e in envs:
        e.ac[turner.to(e.device), 2] = int(round((430.0 - 180.0) * 2 ** 23))
    for j in range(6):
        blocks = torch.rand((2, big + 1, N, 3), generator=g) * 2.1 - 1.05
        # a tenth of the heading actions far outside the action space: saturated heading targets everywhere
        blocks[..., 2] *= torch.where(torch.rand((2, big + 1, N), generator=g) < 0.1, 6.0, 1.0)
        blocks.reshape(2, -1, 3)[:, turner, 2] = 3.0
        blocks = blocks.cuda()
        outs = [e.rollout(blocks[:, :e.B].contiguous(), hold=10) for e in envs]
        for k in ("obs", "reward", "done", "flags"):
            assert torch.equal(outs[0][k], outs[1][k][:, :small]), (k, "latency-bound vs throughput")
            assert torch.equal(outs[1][k], outs[2][k][:, :big]), (k, "throughput vs general")
        # (observation word 3 of a turner beyond 436 deg: normalised (phi - 180) / 180 > 1.4223 — only a WIDE heading gets there)
        # (steps that ended an episode return the RAW reset observation instead: masked out)
        o3 = outs[1]["obs"][:, :small].reshape(-1, small, N, 10)[..., 3]
        saw_wide = saw_wide or bool(((o3 > 1.45) & (outs[1]["done"][:, :small] == 0)[..., None]).any())
    for name in ("ac", "alt", "last_act"):
        assert torch.equal(getattr(envs[0], name), getattr(envs[1], name)[:small * N]), name
        assert torch.equal(getattr(envs[1], name), getattr(envs[2], name)[:big * N]), name
    assert torch.equal(envs[0].phi_counts, envs[1].phi_counts[:small * N]) and torch.equal(envs[1].phi_counts, envs[2].phi_counts[:big * N])
    assert saw_wide, "no heading left the 32-bit field: the WIDE path was not exercised"
    assert torch.equal(envs[0].env, envs[1].env[:small]) and torch.equal(envs[0].stats, envs[1].stats[:small])
    assert int(envs[0].episodes.sum()) > small
    for e in envs:
        e.close()


def test_wide_headings_in_host_mapped_state():
    """Small batches of the SB adapter / the single-env AtcGym keep their state in pinned host memory mapped into the device: the
    side record of WIDE headings (ABI 19) is read and written over the host link there.  Same results as the same envs in HBM."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from envs.atc import scenarios
    scn = scenarios.LOWW(random_entrypoints=True)
    dev_env = AtcVecEnv(8, 4, scenario=scn, seed=2, auto_reset=False, keep_active=True)
    map_env = AtcVecEnv(8, 4, scenario=scn, seed=2, auto_reset=False, keep_active=True, host_mapped=True)
    g = torch.Generator(device="cpu").manual_seed(4)
    for t in range(300):
        if t % 25 == 0:
            a = torch.rand((8, 4, 3), generator=g) * 2 - 1
            a[..., 2] *= 5.0
        o1, r1, d1, i1 = dev_env.step(a.cuda())
        o2, r2, d2, i2 = map_env.step(a.pin_memory())
        assert torch.equal(o1.cpu(), o2) and torch.equal(r1.cpu(), r2) and torch.equal(i1["flags"].cpu(), i2["flags"]), t
    assert torch.equal(dev_env.phi_counts.cpu(), map_env.phi_counts)
    assert int(((map_env.phi_fix == -2 ** 31) | (map_env.phi_fix == 2 ** 31 - 1)).sum()) > 0
    assert float(dev_env.phi.abs().max()) > 700.0
    dev_env.close()
    map_env.close()


def closing_course_formations(comp, N, phases=16, pairs=True):
    """States and (held) actions for test_scan_horizon_on_the_fastest_closing_courses: [B][N] x (x, y, h, phi, v), [B][N][3].
    B = 4 kinds of pair x `phases` gaps x the envs of one wavefront (each wavefront holds one kind at one gap)."""
    from oracle import oracle as O
    G = max(1, 64 // N)
    B = 4 * phases * G
    pitch = 7.0 if N <= 16 else 5.0
    bx0, by0, bx1, by1 = comp.meta["bbox"]
    gx, gy = np.meshgrid(np.arange(bx0, bx1, pitch), np.arange(by0, by1, pitch))
    gx, gy = gx.ravel(), gy.ravel()
    q = O.OracleQueries(comp, np.float32)
    inside = lambda x, y: q.mva(x, y) >= 0
    ok = np.ones(len(gx), bool)   # a point with everything within 4.5 nm of it inside the airspace: nobody leaves it during the run
    for dx, dy in ((0, 0), (4.5, 0), (-4.5, 0), (0, 4.5), (0, -4.5), (3.2, 3.2), (-3.2, 3.2), (3.2, -3.2), (-3.2, -3.2)):
        ok &= inside(gx + dx, gy + dy)
    # the anchor of the pair: its row's neighbour to the east is a formation point too (the head-on partner starts near it)
    idx = np.flatnonzero(ok)
    anchor = next(i for i in idx if i + 1 in idx and gy[i + 1] == gy[i])
    rest = [i for i in idx if i not in (anchor, anchor + 1)][:N - 2]
    assert len(rest) == N - 2, "sector too small for the formation"
    px, py = np.r_[gx[anchor], gx[anchor + 1], gx[rest]], np.r_[gy[anchor], gy[anchor + 1], gy[rest]]
    ch, cv = 2 * 300.0 / 3600.0, 56.0
    hold_h = float(np.float32(30000.0 / 19000.0 - 1.0))
    st = np.zeros((B, N, 5))
    act = np.zeros((B, N, 3), np.float32)
    st[:, :, 0], st[:, :, 1], st[:, :, 2], st[:, :, 3], st[:, :, 4] = px, py, 30000.0, 0.0, 250.0   # the formation flies north at 250 kt
    act[:, :, 0], act[:, :, 1], act[:, :, 2] = 0.5, hold_h, -1.0
    for e in range(B if pairs else 0):
        kind, gap = (e // G) % 4, (e // (4 * G)) % phases
        if kind in (0, 1):      # head-on along the x axis, at 300 kt / from 356 kt (the speed format's limit)
            v = 300.0 if kind == 0 else 356.0
            st[e, 0] = (px[0], py[0], 30000.0, 90.0, v)
            st[e, 1] = (px[0] + 3.0 + ch * gap + 0.03, py[0], 30000.0, 270.0, v)
            act[e, 0], act[e, 1] = (1.0, hold_h, -0.5), (1.0, hold_h, 0.5)
        elif kind == 2:         # stacked over one point: the lower one climbs, the upper one descends
            st[e, 0] = (px[0], py[0], 20000.0, 0.0, 250.0)
            st[e, 1] = (px[0], py[0], 21000.0 + cv * gap + 5.0, 0.0, 250.0)
            act[e, 0], act[e, 1] = (0.5, 1.0, -1.0), (0.5, -0.5, -1.0)
        else:                   # crossing at right angles, meeting at one point
            d = (3.0 + ch * gap) / np.sqrt(2.0) + 0.03
            st[e, 0] = (px[0] - d, py[0], 30000.0, 90.0, 300.0)
            st[e, 1] = (px[0], py[0] - d, 30000.0, 0.0, 300.0)
            act[e, 0], act[e, 1] = (1.0, hold_h, -0.5), (1.0, hold_h, -1.0)
    return st, act, G


@pytest.mark.parametrize("N", [16, 64])
def test_scan_horizon_on_the_fastest_closing_courses(N):
    """Multi-step launches of the fast variant leave the separation scan out for a few steps after one that found every pair of the
    wavefront beyond the horizon thresholds (csrc/atc_step.hip: scan_horizon_limits; include/atc_step.h "Separation scan horizon").
    Here the pairs close as fast as the step allows — head-on at 300 kt each, head-on from 356 kt (the speed format's limit: the
    horizon's premise fails and nothing may be skipped), stacked over one point at 15 ft/s up against 41 ft/s down, crossing at right
    angles — from gaps swept so that the loss of separation falls on every phase of a horizon, in formations whose other aircraft
    keep their distance (the wavefronts DO skip).  Every step must equal the single-step launches, which scan in every step, and the
    oracle's flags."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from atc_hip import layout as L
    from envs.atc import scenarios
    from oracle import oracle as O
    scn = scenarios.LOWWDense() if N == 64 else scenarios.LOWW(random_entrypoints=True)
    comp = scenarios.compile_scenario(scn)
    st, act, G = closing_course_formations(comp, N)
    B = st.shape[0]
    one = AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=3)
    roll = AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=3)
    orc = O.OracleEnv(comp, B, N, O.make_params(auto_reset=True, seed=3), np.float32)
    for e in range(B):
        for k in range(N):
            orc.set_state(e, k, *st[e, k])
    for env in (one, roll):   # (the oracle's words: one copy instead of B N x 5 scalar writes)
        env.ac[:, 0] = torch.as_tensor(orc.px).to(env.device)
        env.ac[:, 1] = torch.as_tensor(orc.py).to(env.device)
        env.h[:] = torch.as_tensor(orc.h).to(env.device)
        env.phi_fix[:] = torch.as_tensor(orc.phi_fix).to(env.device)
        env.v_fix[:] = torch.as_tensor(orc.v_fix).to(env.device)
    blocks = torch.as_tensor(act).cuda()[None]
    kinds = (np.arange(B) // G) % 4
    conflict_steps = set()
    for launch in range(2):
        out = roll.rollout(blocks, hold=20)
        for t in range(20):
            o, r, d, info = one.step(blocks[0])
            orc.step(act)
            assert torch.equal(out["flags"][t], info["flags"]) and torch.equal(out["done"][t], d), (launch, t)
            assert torch.equal(out["obs"][t], o) and torch.equal(out["reward"][t], r), (launch, t)
            assert np.array_equal(info["flags"].cpu().numpy().astype(np.uint32).reshape(B, N), orc.flags), (launch, t)
            if launch == 0:
                hit = (orc.flags[:, :2] & L.F_CONFLICT) != 0
                conflict_steps |= {(kind, t) for kind in range(4) if hit[kinds == kind].any()}
    for name in ("ac", "alt", "last_act", "env", "stats"):
        assert torch.equal(getattr(one, name), getattr(roll, name)), name
    # every kind of pair lost its separation at many different steps of the launch (= phases of the horizon)
    for kind in range(4):
        assert len([1 for k, _ in conflict_steps if k == kind]) >= 10, (kind, sorted(conflict_steps))
    one.close()
    roll.close()


@pytest.mark.parametrize("N", [64, 32, 40])
def test_state_written_between_launches_of_large_envs(N):
    """Nothing about the separation scan may be carried from launch to launch behind the caller's back (the scan horizon lives inside
    a multi-step launch only; a variant that kept it in the state between single-step launches was measured and not shipped:
    profiles/r05_experiments.txt, ab_s19): an aircraft placed next to another one between two steps, a multi-step launch in between,
    a separation minimum changed in the parameters — every step gives the oracle's flags."""
    torch = _torch()
    from atc_hip.vec_env import AtcVecEnv
    from atc_hip import layout as L
    from envs.atc import scenarios
    from oracle import oracle as O
    scn = scenarios.LOWWDense()
    comp = scenarios.compile_scenario(scn)
    # every env starts as the formation alone: everybody on a lattice point, north-bound at 250 kt
    st, act, G = closing_course_formations(comp, 64 if N == 40 else N, phases=4, pairs=False)
    st, act = st[:, :N], act[:, :N]
    B = st.shape[0]
    env = AtcVecEnv(B, N, scenario=scn, auto_reset=True, seed=3)
    orc = O.OracleEnv(comp, B, N, O.make_params(auto_reset=True, seed=3), np.float32)

    def place(e, k, *state):
        env.set_state(e, k, *state)
        orc.set_state(e, k, *state)

    for e in range(B):
        for k in range(N):
            place(e, k, *st[e, k])
    a = torch.as_tensor(act).cuda()
    seen = {"conflict": 0}

    def step(n=1, rollout=False):
        for _ in range(n):
            if rollout:
                out = env.rollout(a[None], hold=1)
                fl = out["flags"][0]
            else:
                fl = env.step(a)[3]["flags"]
            orc.step(act)
            f = fl.cpu().numpy().astype(np.uint32).reshape(B, N)
            assert np.array_equal(f, orc.flags), np.argwhere(f != orc.flags)[:5]
            seen["conflict"] += int(((orc.flags & L.F_CONFLICT) != 0).sum())

    step(3)                                   # (nobody is near anybody)
    assert seen["conflict"] == 0 and not (orc.flags != 0).any(), "the formation is not clear of the sector's own flags"
    x, y = orc.x.reshape(B, N), orc.y.reshape(B, N)
    place(0, 5, x[0, 9] + 0.5, y[0, 9], 30000.0, 0.0, 250.0)            # an aircraft appears half a mile from another one
    step(1)
    assert seen["conflict"] >= 2
    c0 = seen["conflict"]
    step(2)
    x, y = orc.x.reshape(B, N), orc.y.reshape(B, N)
    place(1, 7, x[1, 20] + 2.9, y[1, 20], 30000.0, 0.0, 250.0)
    step(1)
    assert seen["conflict"] >= c0 + 2
    c0 = seen["conflict"]
    step(3)
    # multi-step launches in between, then an aircraft placed again
    step(2, rollout=True)
    x, y = orc.x.reshape(B, N), orc.y.reshape(B, N)
    place(2, 3, x[2, 30] - 1.0, y[2, 30] + 1.0, 30000.0, 0.0, 250.0)
    step(1)
    assert seen["conflict"] >= c0 + 2
    c0 = seen["conflict"]
    step(3)
    # the separation minimum grows beyond the lattice pitch: every env is in conflict at once
    for p in (env.params, orc.params):
        p.sep_nm = 5.5
    env.refresh_params()
    step(1)
    assert seen["conflict"] > c0 + B
    env.close()
