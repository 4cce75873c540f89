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
# command curriculum (commands.curriculum=true): what the grid does -- levels of the envs, cells that can be drawn, and how many of the episodes that
# ended were long enough to count as a success at all (reference envs/t1.py:391-396: episode length > ceil(episode_length_s / dt) * (1 -
# episode_length_toler), THEN the three tracking tolerances); episode lengths are followed on the host side from the rollout's done flags
curr = bool(cfg["commands"].get("curriculum", False))
T, N = cfg["runner"]["horizon_length"], r.env.num_envs
import math
need = math.ceil(cfg["rewards"]["episode_length_s"] / r.env.dt) * (1.0 - cfg["commands"]["episode_length_toler"])
cur_len = torch.zeros(N, device=r.device)
ended = long_enough = 0
for it in range(iters):
    stats = r.iteration()
    if curr:
        d = r.buffer["dones"]
        for t in range(T):
            cur_len += 1
            e = d[t]
            ended += int(e.sum()); long_enough += int((cur_len[e] > need).sum())
            cur_len[e] = 0
    if (it + 1) % 100 == 0:
        s = r._summarize(stats)
        es = r.env.episode_stats(reset=True).cpu().tolist()
        n = max(es[0], 1.0)
        row = {"it": it + 1, "t": round(time.time() - t0, 1), "ep_len": round(es[1] / n, 1), "ep_rew": round(es[2] / n, 3), "episodes": int(es[0]),
               "track_x": round(es[3 + 1] / n, 3), "v_loss": round(s["value_loss"], 4), "kl": round(s["kl_mean"], 4), "lr": s["lr"], "entropy": round(s["entropy"], 2),
               "nonfinite": es[-1]}
        if curr:
            lin, ang = r.env.get_field("env_curriculum_level_lin").abs().float(), r.env.get_field("env_curriculum_level_ang").abs().float()
            g = r.env.curriculum_prob
            row.update({"curriculum/mean_lin_vel_level": round(float(lin.mean()), 3), "curriculum/max_lin_vel_level": int(lin.max()),
                        "curriculum/mean_ang_vel_level": round(float(ang.mean()), 3), "curriculum/max_ang_vel_level": int(ang.max()),
                        "grid_cells_positive": int((g > 0).sum()), "grid_sum": round(float(g.sum()), 2),
                        "episodes_ended": ended, "share_long_enough_for_success": round(long_enough / max(ended, 1), 4)})
            ended = long_enough = 0
        hist.append(row); print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
torch.save(r.checkpoint_dict(), "gpurun_out/train_probe.pth")
json.dump(hist, open("gpurun_out/train_probe.json", "w"))
