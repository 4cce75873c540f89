import os, sys, subprocess
code = '''
import sys; sys.path.insert(0, "/root/repo")
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.envs import T1
env = T1(load_cfg("T1", {"env.num_envs": 4096, "terrain.type": "plane"}))
env.reset(); act = torch.zeros(4096, 12, device=env.device)
for _ in range(20): env.step(act)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): env.step(act)
e1.record(); torch.cuda.synchronize()
print("env.step us:", e0.elapsed_time(e1)/50*1e3)
'''
for lib in [None, "tools/probe/variant4.bin", "tools/probe/variant5.bin"]:
    env = dict(os.environ)
    if lib: env["BG_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
