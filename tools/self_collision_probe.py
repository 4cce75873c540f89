"""How often do the legs of the T1 interpenetrate in this build, which does not model self-collision (reference: envs/T1.yaml:69
`self_collisions: 0` = enabled in PhysX, envs/t1.py:128)?  Trains with the shipped configuration and, on every 10th iteration, checks after
every env step the collision primitives of the left leg against those of the right leg (URDF <collision>: hip-yaw and shank cylinders, foot boxes;
each covered by spheres: 3 per cylinder, 15 per foot box) for all envs.  Reports the fraction of env-steps with any overlap deeper than
0 / 5 / 20 mm, by link pair, over the course of training.
    python tools/self_collision_probe.py [iterations=10000] -> gpurun_out/self_collision_probe.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.runner import Runner

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = load_cfg("T1", {"basic.max_iterations": iters, "basic.seed": 42})
r = Runner(cfg=cfg)
env, m, dev = r.env, r.env.model, r.env.device
pos = torch.tensor(m.body_pos, dtype=torch.float32, device=dev)
axis = [int(a) for a in m.joint_axis]


def rot(ax, q):
    c, s, o, z = torch.cos(q), torch.sin(q), torch.ones_like(q), torch.zeros_like(q)
    rows = {1: [o, z, z, z, c, -s, z, s, c], 2: [c, z, s, z, o, z, -s, z, c], 3: [c, -s, z, s, c, z, z, z, o]}[ax]
    return torch.stack(rows, dim=-1).view(-1, 3, 3)


def quat_to_mat(q):  # xyzw
    x, y, z, w = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1).view(-1, 3, 3)


# covering spheres per leg link (local frame): (link offset in the leg 0..5, centre, radius)
def leg_spheres():
    out = []
    for z in (-0.03, 0.0, 0.03):
        out.append((2, (0.0, 0.0, z), 0.05, "hip_yaw"))
    for z in (-0.145, -0.12, -0.095):
        out.append((3, (0.0, 0.0, z), 0.05, "shank"))
    for x in (-0.0865, -0.038, 0.01, 0.058, 0.1065):
        for y in (-0.035, 0.0, 0.035):
            out.append((5, (x, y, -0.015), 0.015, "foot"))
    return out


SPH = leg_spheres()
kinds = ["hip_yaw", "shank", "foot"]
kind_of = torch.tensor([kinds.index(s[3]) for s in SPH], device=dev)
cen = torch.tensor([s[1] for s in SPH], dtype=torch.float32, device=dev)
rad = torch.tensor([s[2] for s in SPH], dtype=torch.float32, device=dev)
link_of = [s[0] for s in SPH]


def leg_sphere_world(root, q, leg):
    R, p = quat_to_mat(root[:, 3:7]), root[:, 0:3]
    Rs, ps = [], []
    for i in range(6):
        b = 1 + leg * 6 + i
        p = p + (R @ pos[b].view(1, 3, 1)).squeeze(-1)
        R = R @ rot(axis[b], q[:, leg * 6 + i])
        Rs.append(R); ps.append(p)
    Rl = torch.stack([Rs[k] for k in link_of], dim=1)  # [N, S, 3, 3]
    pl = torch.stack([ps[k] for k in link_of], dim=1)
    return pl + (Rl @ cen.view(1, -1, 3, 1)).squeeze(-1)


pair_kind = kind_of.view(-1, 1) * 3 + kind_of.view(1, -1)  # [S, S]
acc = {}
def sample(it):
    root, q = env.root_states, env.dof_pos
    a, b = leg_sphere_world(root, q, 0), leg_sphere_world(root, q, 1)
    d = torch.cdist(a, b)  # [N, S, S]
    depth = (rad.view(1, -1, 1) + rad.view(1, 1, -1)) - d
    worst = depth.flatten(1).max(dim=1)
    bucket = acc.setdefault(it // 1000, {"env_steps": 0, "gt0": 0, "gt5mm": 0, "gt20mm": 0, "max_depth": 0.0, "pairs_gt5mm": [0] * 9})
    bucket["env_steps"] += root.shape[0]
    bucket["gt0"] += int((worst.values > 0).sum()); bucket["gt5mm"] += int((worst.values > 0.005).sum()); bucket["gt20mm"] += int((worst.values > 0.02).sum())
    bucket["max_depth"] = max(bucket["max_depth"], float(worst.values.max()))
    hit = worst.values > 0.005
    if hit.any():
        pk = pair_kind.flatten()[worst.indices[hit]]
        for k in range(9):
            bucket["pairs_gt5mm"][k] += int((pk == k).sum())


obs, infos = env.reset()
r.buffer["obses"][0].copy_(obs); r.buffer["privileged_obses"][0].copy_(infos["privileged_obs"])
orig = env.step_to
state = {"it": 0}
def hooked(*a, **k):
    orig(*a, **k)
    if state["it"] % 10 == 0:
        sample(state["it"])
env.step_to = hooked
t0 = time.time()
for it in range(iters):
    state["it"] = it
    r.iteration()
    if (it + 1) % 1000 == 0:
        es = env.episode_stats(reset=True).cpu().tolist()
        b = acc[it // 1000]
        print(json.dumps({"it": it + 1, "t": round(time.time() - t0, 1), "ep_len": round(es[1] / max(es[0], 1), 1), "frac_gt0": b["gt0"] / b["env_steps"],
                          "frac_gt5mm": b["gt5mm"] / b["env_steps"], "frac_gt20mm": b["gt20mm"] / b["env_steps"], "max_depth_m": b["max_depth"]}), flush=True)
names = [f"{kinds[i]}(L)-{kinds[j]}(R)" for i in range(3) for j in range(3)]
rows = []
for k in sorted(acc):
    b = acc[k]
    rows.append({"iterations": f"{k * 1000}-{k * 1000 + 999}", "env_steps_checked": b["env_steps"], "frac_any_overlap": b["gt0"] / b["env_steps"],
                 "frac_deeper_5mm": b["gt5mm"] / b["env_steps"], "frac_deeper_20mm": b["gt20mm"] / b["env_steps"], "max_depth_m": b["max_depth"],
                 "deepest_pair_when_gt5mm": {names[i]: b["pairs_gt5mm"][i] for i in range(9) if b["pairs_gt5mm"][i]}})
json.dump({"what": __doc__, "config": "shipped T1.yaml (rough terrain, all randomisation), 4096 envs, seed 42", "iterations": iters, "rows": rows},
          open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "self_collision_probe.json"), "w"), indent=1)
