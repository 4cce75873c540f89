"""Small driver for rocprofv3: N env steps (default 4096 envs, plane) + a few forward-dynamics launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.envs import T1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = T1(load_cfg("T1", {"env.num_envs": n, "terrain.type": sys.argv[2] if len(sys.argv) > 2 else "plane"}))
env.reset()
act = torch.zeros(n, 12, device=env.device)
for _ in range(60): env.step(act)
root, q, qd = env.root_states, env.dof_pos, env.dof_vel
for _ in range(20): env.forward_dynamics(root, q, qd, torch.zeros(n, 12, device=env.device))
torch.cuda.synchronize()
