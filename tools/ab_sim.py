"""Same-box A/B of the simulator kernels: env step (4096 envs, plane / trimesh, 10 substeps + task logic) and the ABA substep kernel
(forward dynamics, 1 M envs), HIP events on the launch stream.  BG_LIB selects the library build (default: the in-tree one).
    python tools/ab_sim.py [tag] -> one JSON line"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd import _lib
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.envs import T1

tag = sys.argv[1] if len(sys.argv) > 1 else "current"
SC = int(os.environ.get("AB_SELF_MASK", "0"))  # asset.self_collisions filter mask: 0 = legs collide
out = {"tag": tag, "self_mask": SC, "lib": os.path.basename(_lib.LIB_PATH)}
for terrain in ("plane", "trimesh"):
    n = 4096
    env = T1(load_cfg("T1", {"env.num_envs": n, "terrain.type": terrain, "asset.self_collisions": SC}))
    env.reset()
    g = torch.Generator(device="cpu").manual_seed(0)
    acts = [(torch.rand(n, 12, generator=g) * 0.6 - 0.3).to(env.device) for _ in range(8)]
    for k in range(40): env.step(acts[k % 8])
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(48): env.step(acts[k % 8])
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 48 * 1e3)
    out[f"env_step_us_{terrain}"] = round(best, 2)
    del env
n = 1048576
env = T1(load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane", "asset.self_collisions": SC}))
dev = env.device
g = torch.Generator(device="cpu").manual_seed(1234)
root = torch.zeros(n, 13); root[:, 2] = 0.66; root[:, 6] = 1.0; root[:, 7:13] = torch.randn(n, 6, generator=g) * 0.3
q = torch.tensor([-0.2, 0, 0, 0.4, -0.25, 0] * 2).repeat(n, 1) + torch.randn(n, 12, generator=g) * 0.1
qd = torch.randn(n, 12, generator=g); tau = (torch.rand(n, 12, generator=g) * 2 - 1) * 20
root, q, qd, tau = (t.to(dev).contiguous() for t in (root, q, qd, tau))
qacc = torch.empty(n, 18, device=dev)
lib = _lib.load()
call = lambda: _lib.check(lib.bg_env_forward_dynamics(env._env, _lib.ptr(root), _lib.ptr(q), _lib.ptr(qd), _lib.ptr(tau), None, _lib.ptr(qacc), _lib.current_stream_ptr()))
for _ in range(5): call()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
out["aba_1M_us"] = round(best, 2)
out["aba_frac_548B"] = round(n * 548 / best / 1e3 / 8000.0, 4)
print(json.dumps(out), flush=True)
