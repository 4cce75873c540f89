"""K1 roofline curve: HIP forward dynamics (one ABA substep per launch, 548 algorithmic bytes per env) over N, HIP events on the launch stream."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from booster_gym_amd import _lib
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.envs import T1
BYTES = 4 * (13 + 12 + 12 + 12 + 6 + 18 + 6 + 58)
rows = []
for n in (4096, 16384, 65536, 262144, 1048576):
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
    for _ in range(max(5, min(400, int(80e3 / max(1.0, n * 2e-4))))): call()  # ~80 ms of untimed launches: past the clock transient (tools/aba_series.py)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K = 50
    for _ in range(K): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / K * 1e3
    gbs = n * BYTES / us / 1e3
    rows.append({"num_envs": n, "avg_launch_us": us, "substeps_per_s": n / us * 1e6, "achieved_GBps": gbs, "frac_of_8TBps": gbs / 8000.0})
    print(json.dumps(rows[-1]), flush=True)
    del env
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"kernel": "forward_dynamics_kernel", "algorithmic_bytes_per_env": BYTES, "rows": rows}, open("gpurun_out/dyn_roofline.json", "w"), indent=1)
