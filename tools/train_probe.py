"""Short real training run on the GPU: does PPO learn to walk in this simulator?  Prints reward / episode length every 100 iterations."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.runner import Runner
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
terrain = sys.argv[2] if len(sys.argv) > 2 else "trimesh"
over = {"terrain.type": terrain, "basic.max_iterations": iters}
for kv in sys.argv[3:]:  # extra overrides, e.g. env.num_envs=16384 sim.state_dtype=fp16
    k, v = kv.split("=", 1)
    try:
        v = json.loads(v)
    except ValueError:
        pass
    over[k] = v
cfg = load_cfg("T1", over)
r = Runner(cfg=cfg)
obs, infos = r.env.reset()
r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
t0 = time.time()
hist = []
for it in range(iters):
    stats = r.iteration()
    if (it + 1) % 100 == 0:
        s = r._summarize(stats)
        es = r.env.episode_stats(reset=True).cpu().tolist()
        n = max(es[0], 1.0)
        row = {"it": it + 1, "t": round(time.time() - t0, 1), "ep_len": round(es[1] / n, 1), "ep_rew": round(es[2] / n, 3), "episodes": int(es[0]),
               "track_x": round(es[3 + 1] / n, 3), "v_loss": round(s["value_loss"], 4), "kl": round(s["kl_mean"], 4), "lr": s["lr"], "entropy": round(s["entropy"], 2),
               "nonfinite": es[-1]}
        hist.append(row); print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
torch.save(r.checkpoint_dict(), "gpurun_out/train_probe.pth")
json.dump(hist, open("gpurun_out/train_probe.json", "w"))
