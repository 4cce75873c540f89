"""Child process of tests/test_gpu_rccl.py: every RCCL call of booster_gym_amd/utils/parallel.py executed in a world of ONE rank on the one GPU
of the box (backend "nccl" = RCCL), in the stream pattern Runner.update() uses, then one whole PPO iteration through that path.  Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.parallel import DataParallel
from booster_gym_amd.utils.runner import Runner

out = {}
if os.environ.get("BG_DIST_FORCE", "0") != "1":
    # the same seeded iteration WITHOUT a process group (single-process two-launch tail): the yardstick for the collective path's arithmetic
    r = Runner(cfg=load_cfg("T1", {"env.num_envs": 64, "terrain.type": "plane", "runner.mini_epochs": 2, "commands.curriculum": True}))
    assert not r.dp.active
    obs, infos = r.env.reset()
    r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
    stats = r.iteration()
    torch.cuda.synchronize()
    print("RCCL_WORLD1 " + json.dumps({"params": r.optimizer.flat.double().cpu().tolist()[::97], "stats": stats.cpu().tolist(), "lr": float(r.optimizer.lr)}), flush=True)
    sys.exit(0)
dp = DataParallel()
assert dp.active and dp.world_size == 1 and dp.backend == "nccl" and dist.is_initialized() and dist.get_backend() == "nccl"
dev = torch.device(f"cuda:{dp.device_index}")
dp.barrier()                                   # barrier(device_ids=[...])
main, side = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
a64 = torch.arange(5, dtype=torch.float64, device=dev) + 0.25
g32 = torch.randn(177945, device=dev)          # the flat gradient bucket
g0 = g32.clone()
side.wait_stream(main)
with torch.cuda.stream(side):                  # collectives issued on the side stream, as exchanges (1) and (3) of Runner.update()
    dp.sum_(a64)
    dp.average_(g32)
main.wait_stream(side)
dep = g32 * 2.0 + a64.sum().float()            # a dependent kernel on the main stream
torch.cuda.synchronize()
own = os.environ.get("BG_OWN_RCCL", "1") != "0"
assert (dp.comm is not None) == own, "the per-mini-epoch exchanges go through the own communicator (utils/rccl.py) unless BG_OWN_RCCL=0"
out["own_rccl"] = dp.comm is not None
s64, l64 = torch.arange(5, dtype=torch.float64, device=dev) + 0.5, torch.arange(12, dtype=torch.float64, device=dev) - 3.0
dp.exchange_tail_(g32, s64, l64)               # bucket (mean) + loss sums (sum) + log-std gradient (mean): one grouped launch on the current stream
torch.cuda.synchronize()
out["group_exact"] = bool(torch.equal(g32, g0) and torch.equal(s64, torch.arange(5, dtype=torch.float64, device=dev) + 0.5)
                          and torch.equal(l64, torch.arange(12, dtype=torch.float64, device=dev) - 3.0))
out["sum_fp64_exact"] = bool(torch.equal(a64, torch.arange(5, dtype=torch.float64, device=dev) + 0.25))
out["avg_fp32_exact"] = bool(torch.equal(g32, g0))
out["dependent_ok"] = bool(torch.allclose(dep, g0 * 2.0 + float(a64.sum())))
out["max"] = float(dp.max_(torch.tensor([3.5], device=dev))[0])
out["broadcast_int"] = dp.broadcast_int(1234)
cur, last = torch.full((21, 21), 0.3, device=dev), torch.full((21, 21), 0.2, device=dev)
out["sync_grid"] = float(dp.sync_grid(cur, last).mean())
# one PPO iteration (24 env steps + 2 mini-epochs, 64 envs) with every exchange of the update issued through RCCL
r = Runner(cfg=load_cfg("T1", {"env.num_envs": 64, "terrain.type": "plane", "runner.mini_epochs": 2, "commands.curriculum": True}))
assert r.dp.active and r.dp.backend == "nccl"
obs, infos = r.env.reset()
r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
p0 = r.optimizer.flat.clone()
stats = r.iteration()
r._sync_curriculum()
torch.cuda.synchronize()
out["iteration_finite"] = bool(torch.isfinite(stats).all() and torch.isfinite(r.optimizer.flat).all())
out["parameters_moved"] = bool((r.optimizer.flat - p0).abs().max() > 0)
out["params"], out["stats"], out["lr"] = r.optimizer.flat.double().cpu().tolist()[::97], stats.cpu().tolist(), float(r.optimizer.lr)
dp.barrier()
dp.shutdown()
out["shutdown"] = not dist.is_initialized()
print("RCCL_WORLD1 " + json.dumps(out), flush=True)
