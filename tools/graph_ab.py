"""Developer A/B (GPU box): 20 single-step launches of the headline workload replayed as ONE hipGraph (stream capture of the
same atc_step calls) against the 20 individual launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "atc-reinforcement-learning_amd")]
import torch
from atc_hip.vec_env import AtcVecEnv
from envs.atc import scenarios
B, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 16, 20
dev = torch.device("cuda", 0)
env = AtcVecEnv(B, N, scenario=scenarios.LOWW(random_entrypoints=N > 1), auto_reset=True, seed=11)
g = torch.Generator(device=dev); g.manual_seed(1)
a0 = torch.rand((B, N, 3), generator=g, device=dev) * 2 - 1
s = torch.cuda.Stream(dev)
first, rest = env.make_launcher(a0, stream=s), env.make_launcher(a0, stream=s, held=True)
def block():
    first()
    for _ in range(K - 1):
        rest()
with torch.cuda.stream(s):
    for _ in range(50):
        block()
s.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s):
    block()
def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record(s)
        for _ in range(reps):
            fn()
        e1.record(s)
    s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * K)
for rnd in range(3):
    print("launches %.3f us/step   graph %.3f us/step" % (timed(block, 100), timed(graph.replay, 100)), flush=True)
# the driver's protocol: ONE block between two synchronisations, wall clock
for name, fn in (("launches", block), ("graph", graph.replay)):
    ws = []
    for rep in range(40):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            fn()
        torch.cuda.synchronize(dev)
        ws.append((time.perf_counter() - t0) / K * 1e6)
    ws.sort()
    print("one block, wall: %-9s median %.3f min %.3f us/step" % (name, ws[20], ws[0]), flush=True)
env.close()
