"""Worker side of tests/test_dropin_boundary.py: what stable-baselines' SubprocVecEnv worker + Monitor do around ONE env
(learning/atc-gym-stable-baselines.py:69-80 of the reference: make_env -> gym.make('AtcEnv-v0') -> Monitor -> seed), driven
over a pipe: ('step', action) -> (obs, reward, done, info) with reset-on-done and Monitor's episode record,
('get_attr', name), ('reset',), ('close',).  Imports gym from tests/oracle_shims (a stand-in without reference code)."""
import os
import sys
import time


def _setup_path():
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    for p in (os.path.join(root, "atc-reinforcement-learning_amd"), os.path.join(here, "oracle_shims"), root):
        if p not in sys.path:
            sys.path.insert(0, p)


def worker(conn, rank, seed):
    _setup_path()
    import gym
    import envs.atc.atc_gym  # noqa: F401  (side effect: registers 'AtcEnv-v0', like the reference's envs/__init__.py:3-5)
    env = gym.make('AtcEnv-v0')
    assert isinstance(env, gym.Env)
    env.seed(seed + rank)
    import numpy as np
    rng = np.random.default_rng(seed + rank)   # the policy's stand-in: actions ~ U(-1, 1) like action_space.sample()
    ep_ret, ep_len, t0 = 0.0, 0, time.time()
    try:
        while True:
            cmd, data = conn.recv()
            if cmd == 'step':
                obs, rew, done, info = env.step(data)
                ep_ret += rew
                ep_len += 1
                if done:  # Monitor's record + SubprocVecEnv's reset-on-done
                    info = dict(info, episode={"r": ep_ret, "l": ep_len, "t": round(time.time() - t0, 6)})
                    obs = env.reset()
                    ep_ret, ep_len = 0.0, 0
                conn.send((obs, rew, done, info))
            elif cmd == 'reset':
                conn.send(env.reset())
            elif cmd == 'sample':
                conn.send(rng.uniform(-1.0, 1.0, 3).astype(np.float32))
            elif cmd == 'get_attr':
                conn.send(getattr(env, data))
            elif cmd == 'close':
                env.close()
                conn.send(True)
                break
    finally:
        conn.close()
