import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.runner import Runner
cfg = load_cfg("T1", {"env.num_envs": 4096, "terrain.type": "plane"})
r = Runner(cfg=cfg)
obs, infos = r.env.reset()
r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
for _ in range(3): r.iteration()
torch.cuda.synchronize()
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); r.rollout(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    r.update(); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    r.buffer.roll()
    print(f"dp.active={r.dp.active}: rollout enqueue {1e3*(t1-t0):6.2f} ms, done {1e3*(t2-t0):6.2f} ms | update enqueue {1e3*(t3-t2):6.2f} ms, done {1e3*(t4-t2):6.2f} ms", flush=True)
