"""Per-launch time of the ABA launch (bg_env_forward_dynamics, 1 M envs, the bench's standing state) over 400 back-to-back launches, one HIP event
pair each: the clock under this kernel is a transient for the first ~40 ms (226 us for launches 1-4 from idle, up to 270 around launch 12, then a
steady decline to the sustained rate from launch ~160 on).  ABA_TREE: another checkout's package (e.g. an older round's) to time with the same script.
    python tools/aba_series.py -> stdout (kept: profiles/r04_aba_series.txt)"""
import sys, os
sys.path.insert(0, os.environ.get("ABA_TREE", os.getcwd()))
import torch, bench
from booster_gym_amd import _lib
from booster_gym_amd.envs import T1
from booster_gym_amd.utils.config import load_cfg
n = 1 << 20
env = T1(load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane"}))
dev = env.device
g = torch.Generator(device="cpu").manual_seed(1234)
root = torch.zeros(n, 13); root[:, 2] = 0.66; root[:, 6] = 1.0; root[:, 7:13] = torch.randn(n, 6, generator=g) * 0.3
q = torch.tensor([-0.2, 0, 0, 0.4, -0.25, 0] * 2).repeat(n, 1) + torch.randn(n, 12, generator=g) * 0.1
qd = torch.randn(n, 12, generator=g); tau = (torch.rand(n, 12, generator=g) * 2 - 1) * 20
root, q, qd, tau = (t.to(dev).contiguous() for t in (root, q, qd, tau))
qacc = torch.empty(n, 18, device=dev)
lib = _lib.load()
call = lambda: _lib.check(lib.bg_env_forward_dynamics(env._env, _lib.ptr(root), _lib.ptr(q), _lib.ptr(qd), _lib.ptr(tau), None, _lib.ptr(qacc), _lib.current_stream_ptr()))
torch.cuda.synchronize()
K = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
ev[0].record()
for k in range(K):
    call(); ev[k + 1].record()
torch.cuda.synchronize()
t = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(K)]
print("per-launch us, launches 0-39:", [round(x) for x in t[:40]])
for a in range(40, K, 40):
    print(f"launches {a}-{a+39}: mean {sum(t[a:a+40])/40:.1f} min {min(t[a:a+40]):.1f} max {max(t[a:a+40]):.1f}")
