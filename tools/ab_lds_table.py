"""Developer A/B (GPU box): us per step of atc_rollout_hold (T = 20) at 65 536 x 1 with the LDS-resident lookup table attached / not.
  python tools/ab_lds_table.py [rounds] [lds|grid|both] [launches]"""
import sys, torch, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'atc-reinforcement-learning_amd'))
from atc_hip import lib as _binding
if os.environ.get('ATC_AB_LIB'): _binding.use_library(os.environ['ATC_AB_LIB'])
from atc_hip.vec_env import AtcVecEnv
from envs.atc import scenarios
B, T = 65536, 20
g = torch.Generator(device="cpu").manual_seed(1)
ring = [(torch.rand((1, B, 1, 3), generator=g) * 2 - 1).cuda() for _ in range(8)]
res = {}
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
WHICH = sys.argv[2] if len(sys.argv) > 2 else 'both'
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
for rnd in range(ROUNDS):
  for lds in [v for v in (True, False) if WHICH == 'both' or (WHICH == 'lds') == v]:
    env = AtcVecEnv(B, 1, scenario=scenarios.LOWW(), auto_reset=True, seed=11, lds_table=lds)
    ro = {"obs": torch.empty((T, B, 10), dtype=torch.float32, device='cuda'), "reward": torch.empty((T, B), dtype=torch.float32, device='cuda'),
          "done": torch.empty((T, B), dtype=torch.uint8, device='cuda'), "flags": torch.empty((T, B, 1), dtype=torch.int16, device='cuda')}
    for j in range(60): env.rollout(ring[j % 8], out=ro, hold=T)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for j in range(N): env.rollout(ring[j % 8], out=ro, hold=T)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (N * T))
    print('lds_table', lds, env.sector.has_lds_table, 'us/step', ['%.3f' % t for t in ts], flush=True)
    env.close()
