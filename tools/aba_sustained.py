"""Sustained time of the ABA launch at 1 M envs (bench.py's regime: 300 untimed launches, then the median of five series of 50; HIP events on the launch
stream), both bench states, for same-box A/B of library builds (BG_LIB) or of the two kernel forms.   python tools/aba_sustained.py [tag] [packed=0]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
from booster_gym_amd.envs import T1
from booster_gym_amd.utils.config import load_cfg
tag = sys.argv[1] if len(sys.argv) > 1 else "current"
n = 1 << 20
env = T1(load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane"}))
dev, m, lib = env.device, env.model, _lib.load()
entry = lib.bg_env_forward_dynamics_packed if len(sys.argv) > 2 and sys.argv[2] == "1" else lib.bg_env_forward_dynamics
qacc = torch.empty(n, 18, device=dev)
g = torch.Generator(device="cpu").manual_seed(1234)


def sustained(root, q, qd, tau):
    root, q, qd, tau = (t.to(dev).contiguous() for t in (root, q, qd, tau))
    call = lambda: _lib.check(entry(env._env, _lib.ptr(root), _lib.ptr(q), _lib.ptr(qd), _lib.ptr(tau), None, _lib.ptr(qacc), _lib.current_stream_ptr()))
    for _ in range(300):
        call()
    reps = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record(); torch.cuda.synchronize()
        reps.append(e0.elapsed_time(e1) / 50 * 1e3)
    return sorted(reps)[2]


root = torch.zeros(n, 13); root[:, 2] = 0.66; root[:, 6] = 1.0; root[:, 7:13] = torch.randn(n, 6, generator=g) * 0.3
q = torch.tensor([-0.2, 0, 0, 0.4, -0.25, 0] * 2).repeat(n, 1) + torch.randn(n, 12, generator=g) * 0.1
qd = torch.randn(n, 12, generator=g); tau = (torch.rand(n, 12, generator=g) * 2 - 1) * 20
us = sustained(root, q, qd, tau)
lo, hi, eff = (torch.tensor(a, dtype=torch.float32) for a in (m.dof_lower, m.dof_upper, m.dof_effort))
root2 = torch.zeros(n, 13); root2[:, 2] = 0.72
ax = torch.randn(n, 3, generator=g); ax = ax / ax.norm(dim=1, keepdim=True)
ang = torch.rand(n, generator=g) * 0.3
root2[:, 3:6] = ax * torch.sin(ang / 2)[:, None]; root2[:, 6] = torch.cos(ang / 2)
us2 = sustained(root2, lo + (hi - lo) * torch.rand(n, 12, generator=g), torch.randn(n, 12, generator=g), (torch.rand(n, 12, generator=g) * 2 - 1) * eff)
print(json.dumps({"tag": tag, "lib": os.path.basename(_lib.LIB_PATH), "standing_us": round(us, 2), "survey_8d_us": round(us2, 2),
                  "frac_548B": round(n * 548 / us / 1e3 / 8000.0, 4), "survey_frac_548B": round(n * 548 / us2 / 1e3 / 8000.0, 4)}), flush=True)
