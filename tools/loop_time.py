"""The training loop with NO instrumentation (bench.py's timed region carries ~250 HIP timing events per iteration for its roofline entries):
wall-clock ms per iteration over K iterations behind W warm-up iterations.   python tools/loop_time.py [K=20] [W=5] [reps=3] [num_envs=4096]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.runner import Runner
K, W, reps, N = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 20), (2, 5), (3, 3), (4, 4096)))
import tempfile
from booster_gym_amd.utils.recorder import Recorder
cfg = load_cfg("T1", {"env.num_envs": N, "terrain.type": "plane"})
cfg["runner"]["save_interval"] = 10 ** 9  # as bench.py: no checkpoint inside the timed region
r = Runner(cfg=cfg)
r.begin_training(Recorder(cfg, root=tempfile.mkdtemp(prefix="bg_loop_"), rank=0))
it = 0
for _ in range(W):
    r.train_iteration(it); it += 1
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        r.train_iteration(it); it += 1
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
    print(f"no instrumentation, {N} envs: {ms:.3f} ms per iteration = {N * 24 / ms * 1e3 / 1e6:.3f} M env-steps/s", flush=True)
