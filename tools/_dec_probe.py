import os, sys
sys.path.insert(0, "/root/repo")
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.envs import T1
for dec in (1, 2, 5, 10, 20):
    env = T1(load_cfg("T1", {"env.num_envs": 4096, "terrain.type": "plane", "control.decimation": dec}))
    env.reset()
    act = torch.zeros(4096, 12, device=env.device)
    for _ in range(10): env.step(act)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): env.step(act)
    e1.record(); torch.cuda.synchronize()
    print(f"decimation {dec:3d}: {e0.elapsed_time(e1)/30*1e3:8.1f} us per env.step")
    del env
