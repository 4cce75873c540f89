import json, os, sys, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.runner import Runner
from booster_gym_amd.utils.recorder import Recorder
N = int(sys.argv[1]); over = {"env.num_envs": N, "terrain.type": "plane"}
for kv in sys.argv[2:]:
    k, v = kv.split("=", 1)
    try:
        v = json.loads(v)
    except ValueError:
        pass
    over[k] = v
cfg = load_cfg("T1", over); cfg["runner"]["save_interval"] = 10 ** 9
r = Runner(cfg=cfg); r.begin_training(Recorder(cfg, root=tempfile.mkdtemp(prefix="bg_loop_"), rank=0))
it = 0
for _ in range(4):
    r.train_iteration(it); it += 1
torch.cuda.synchronize()
for rep in range(3):
    tr = tu = 0.0
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r.rollout(); torch.cuda.synchronize(); t1 = time.perf_counter(); r.update(); torch.cuda.synchronize(); t2 = time.perf_counter(); r.buffer.roll()
        tr += t1 - t0; tu += t2 - t1
    print(f"{N} envs {sys.argv[2:]}: rollout {tr/5*1e3:.2f} ms, update {tu/5*1e3:.2f} ms, one_tail? chain wgs {r._critic_tr.chain_workgroups}/{r._actor_tr.chain_workgroups}", flush=True)
