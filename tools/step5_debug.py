import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
from gym_trading_env_amd.batched import BatchedTradingEnv
wl = bench.WORKLOADS["c3"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
feat, close = bench.synthetic_dataset(0, 100000, wl["n_static"])
kw = dict(bench.env_kwargs(wl))
a = BatchedTradingEnv((feat, close), num_envs=N, seed=5, output="torch", debug_flags=64, **kw)
a.reset()
acts = torch.randint(0, 3, (64, N), dtype=torch.int32, device='cuda')
bench.desynchronise(a, acts, 500)
g = torch.Generator(device="cuda").manual_seed(1)
for k in range(3):
    act = torch.randint(0, 3, (N,), dtype=torch.int32, device="cuda", generator=g)
    o = a.step(act)[0]
    m = (o[:, 0, 3] == -12345.0)
    print("step", k, "mismatching envs", int(m.sum()), "of", N, a.launch_info())
    if m.any():
        e = m.nonzero()[:5, 0]
        for i in e.tolist():
            print(" env", i, "src diff", float(o[i,0,0]), "real meta", int(o[i,0,1]), "pred meta", int(o[i,0,2]), "idx/step/start", a.state("idx")[i], a.state("step")[i], a.state("start_idx")[i])
a.close()
