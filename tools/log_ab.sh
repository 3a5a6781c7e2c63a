# step time with a trajectory log: the row written by the step kernel (2048) / by a separate launch (1024) / the library's choice (0)
for r in 1 2; do for v in 0 1024 2048; do echo -n "c3 log_steps 2 variant $v: "; python3 - $v <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import bench, torch
from gym_trading_env_amd.batched import BatchedTradingEnv
v = int(sys.argv[1])
wl = bench.WORKLOADS["c3"]; N = wl["envs"]
feat, close = bench.synthetic_dataset(0, wl["T"], wl["n_static"])
acts = torch.randint(0, 3, (64, N), dtype=torch.int32, device="cuda")
env = BatchedTradingEnv((feat, close), num_envs=N, seed=1, output="torch", log_steps=2, kernel_variant=v, verbose=0, **bench.env_kwargs(wl))
env.reset(); bench.desynchronise(env, acts, wl["max_episode_duration"])
for i in range(100): env.step(acts[i % 64])
env.timer_start()
for i in range(600): env.step(acts[i % 64])
print(round(env.timer_stop() * 1e3 / 600, 2), "us/step")
env.close()
PY
done; done
