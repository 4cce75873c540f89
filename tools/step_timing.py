"""Time the bare env-step / forward-dynamics launches (HIP events on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.envs import T1

def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    for n, terrain in [(4096, "plane"), (4096, "trimesh"), (16384, "plane"), (16384, "trimesh"), (65536, "plane"), (262144, "plane")]:
        env = T1(load_cfg("T1", {"env.num_envs": n, "terrain.type": terrain, "sim.state_dtype": dtype}))
        env.reset()
        act = torch.zeros(n, 12, device=env.device)
        for _ in range(20): env.step(act)   # settle onto the ground
        K = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K): env.step(act)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / K * 1e3
        print(f"N={n:7d} {terrain:8s} env.step {us:9.1f} us  {n/us:8.2f} M env-steps/s  {n*1966/us/1e3:8.1f} GB/s algorithmic ({n*1966/us/1e3/8000*100:.2f}% of 8 TB/s)")
        del env
main()
