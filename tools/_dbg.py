import os, sys
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.envs import T1
from gpu_smoke import actor_from_npz
cfg = load_cfg("T1", {"env.num_envs": 4096, "terrain.type": "plane"})
env = T1(cfg)
pi = actor_from_npz(env.device)
obs, ex = env.reset()
torch.cuda.synchronize()
for s in range(3):
    act = pi(obs)
    obs, rew, done, ex = env.step(act)
    a2 = act.clone()
    root = env.root_states
    bad = ~torch.isfinite(root).all(dim=1)
    print("step", s, "nan root", bad.sum().item(), "done", done.sum().item(), "act finite", torch.isfinite(a2).all().item(), a2.abs().max().item())
    if bad.any():
        idx = torch.nonzero(bad).flatten()[:10].cpu().numpy(); print("bad idx", idx, "count by block", np.bincount(torch.nonzero(bad).flatten().cpu().numpy()//32)[:20])
        for k in ["root_states","dof_pos","dof_vel","actions","last_dof_targets","torques","feet_contact_forces","delay_steps"]:
            print(k, env.get_field(k)[idx[0]].cpu().numpy())
        break
