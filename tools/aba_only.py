"""The ABA launch alone (forward dynamics, 1 M envs, standing pose + 0.1 rad joint noise), for rocprofv3 --kernel-trace --stats / --pmc runs.
    python tools/aba_only.py [num_envs] [noise] [launches=12] [packed=0]   (1200 launches: the clock transient of the first ~160 is 13 % of the average;
    packed=1: bg_env_forward_dynamics_packed, the one-env-per-lane kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.envs import T1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
env = T1(load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane"}))
dev = env.device
g = torch.Generator(device="cpu").manual_seed(1234)
root = torch.zeros(n, 13); root[:, 2] = 0.66; root[:, 6] = 1.0; root[:, 7:13] = torch.randn(n, 6, generator=g) * 0.3
q = torch.tensor([-0.2, 0, 0, 0.4, -0.25, 0] * 2).repeat(n, 1) + torch.randn(n, 12, generator=g) * noise
qd = torch.randn(n, 12, generator=g); tau = (torch.rand(n, 12, generator=g) * 2 - 1) * 20
root, q, qd, tau = (t.to(dev).contiguous() for t in (root, q, qd, tau))
qacc = torch.empty(n, 18, device=dev)
lib = _lib.load()
entry = lib.bg_env_forward_dynamics_packed if len(sys.argv) > 4 and sys.argv[4] == "1" else lib.bg_env_forward_dynamics
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 12):
    _lib.check(entry(env._env, _lib.ptr(root), _lib.ptr(q), _lib.ptr(qd), _lib.ptr(tau), None, _lib.ptr(qacc), _lib.current_stream_ptr()))
torch.cuda.synchronize()
print("done", float(qacc.abs().mean()))
