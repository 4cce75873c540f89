"""Is the update phase GPU-bound or launch-bound?  Host time to ENQUEUE one update() (no sync inside) against the GPU time it takes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.runner import Runner
r = Runner(cfg=load_cfg("T1", {"terrain.type": "plane", "basic.seed": 42}))
obs, infos = r.env.reset()
r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
for _ in range(3):
    r.iteration()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); r.rollout(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    r.update(); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"rollout: host enqueue {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms | update: host enqueue {1e3*(t3-t2):.2f} ms, until GPU done {1e3*(t4-t2):.2f} ms", flush=True)
