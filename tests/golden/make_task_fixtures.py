"""Generate golden vectors for the task logic by calling the REFERENCE's own methods (envs/t1.py, utils/terrain.py,
utils/utils.py, utils/model.py) on seeded hand-made inputs.  Run in the build container only:

    python tests/golden/make_task_fixtures.py

/root/reference is imported read-only; `isaacgym` is satisfied by tests/golden/_stub (see its docstring).  Outputs are plain
.npz files of inputs and expected outputs; no reference source text is stored.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, "_stub"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from envs.t1 import T1  # noqa: E402  (reference)
from utils.terrain import Terrain  # noqa: E402  (reference)
from utils import utils as ref_utils  # noqa: E402  (reference)
from utils.model import ActorCritic  # noqa: E402  (reference)

torch.manual_seed(1234)
np.random.seed(1234)
N = 64
cfg = yaml.load(open("/root/reference/envs/T1.yaml"), Loader=yaml.FullLoader)
for k in list(cfg["noise"].keys()):
    cfg["noise"][k] = None  # deterministic observations (utils/utils.py:6-7)
# exercise every reward term, including the three the shipped yaml drops
for k in ("dof_vel_limits", "torque_limits", "feet_vel_z"):
    cfg["rewards"]["scales"][k] = -0.5
cfg["rewards"]["soft_dof_pos_limit"] = 0.9
cfg["rewards"]["soft_dof_vel_limit"] = 0.8
cfg["rewards"]["soft_torque_limit"] = 0.7

# ---- terrain: reference Terrain.terrain_heights on a random int16 grid
ter = object.__new__(Terrain)
ter.type = "trimesh"
ter.device = "cpu"
ter.horizontal_scale, ter.vertical_scale, ter.border_pixels = 0.1, 0.005, 50
ter.height_field_raw = np.random.randint(-20, 20, size=(300, 200)).astype(np.int16)
qxy = torch.rand(500, 3) * torch.tensor([19.0, 9.0, 1.0])
np.savez_compressed(os.path.join(HERE, "terrain_heights.npz"), height_field_raw=ter.height_field_raw, hscale=0.1, vscale=0.005, border_px=50,
                    xy=qxy.numpy(), heights=ter.terrain_heights(qxy).numpy())

# ---- a T1 instance without Isaac Gym: hand-set every attribute the pure-torch methods read
env = object.__new__(T1)
env.cfg, env.device, env.num_envs, env.terrain = cfg, "cpu", N, ter
env.num_dofs, env.num_bodies, env.num_actions = 12, 13, 12
env.dt = cfg["control"]["decimation"] * cfg["sim"]["dt"]
lo = torch.tensor([-1.8, -0.3, -1, 0, -0.87, -0.44, -1.8, -1.57, -1, 0, -0.87, -0.44])
hi = torch.tensor([1.57, 1.57, 1, 2.34, 0.35, 0.44, 1.57, 0.3, 1, 2.34, 0.35, 0.44])
env.dof_pos_limits = torch.stack([lo, hi], dim=1)
env.dof_vel_limits = torch.tensor([12.5, 10.9, 10.9, 11.7, 18.8, 12.4] * 2)
env.torque_limits = torch.tensor([45.0, 30, 30, 60, 24, 15] * 2)
env.default_dof_pos = torch.tensor([[-0.2, 0, 0, 0.4, -0.25, 0] * 2])
env.feet_indices = torch.tensor([6, 12])
env.penalized_contact_indices = torch.tensor([0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11])
env.termination_contact_indices = torch.zeros(0, dtype=torch.long)
env.base_indice = 0


def rnd_quat(n, max_angle):
    ax = torch.randn(n, 3); ax = ax / ax.norm(dim=1, keepdim=True)
    ang = torch.rand(n) * max_angle
    return torch.cat([ax * torch.sin(ang / 2).unsqueeze(1), torch.cos(ang / 2).unsqueeze(1)], dim=1)


env.root_states = torch.zeros(N, 13)
env.root_states[:, 0] = torch.rand(N) * 18; env.root_states[:, 1] = torch.rand(N) * 8
env.root_states[:, 2] = 0.55 + 0.3 * torch.rand(N)
env.root_states[:, 3:7] = rnd_quat(N, 0.6)
env.root_states[:, 7:13] = torch.randn(N, 6) * torch.tensor([1.0, 1, 1, 2, 2, 2])
env.root_states[:4, 7:13] *= 4  # a few envs above terminate_vel
env.base_pos, env.base_quat = env.root_states[:, 0:3], env.root_states[:, 3:7]
env.body_states = torch.zeros(N, 13, 13)
env.body_states[:, :, 0:3] = env.root_states[:, None, 0:3] + torch.randn(N, 13, 3) * 0.2
env.body_states[:, [6, 12], 2] = torch.rand(N, 2) * 0.12 - 0.02  # feet near the ground (terrain ~ +-0.1)
env.body_states[:, :, 3:7] = rnd_quat(N * 13, 0.5).reshape(N, 13, 4)
env.feet_pos, env.feet_quat = env.body_states[:, env.feet_indices, 0:3], env.body_states[:, env.feet_indices, 3:7]
env.dof_pos = lo + (hi - lo) * (torch.rand(N, 12) * 1.2 - 0.1)
env.dof_vel = torch.randn(N, 12) * 6
env.contact_forces = torch.randn(N, 13, 3) * 1.0
env.commands = torch.rand(N, 3) * 2 - 1
env.gait_frequency = torch.rand(N) + 1.0; env.gait_frequency[::5] = 0.0
env.gait_process = torch.rand(N)
env.filtered_lin_vel, env.filtered_ang_vel = torch.randn(N, 3) * 0.5, torch.randn(N, 3) * 0.5
env.base_lin_vel, env.base_ang_vel = torch.randn(N, 3) * 0.5, torch.randn(N, 3)
env.projected_gravity = torch.randn(N, 3); env.projected_gravity /= env.projected_gravity.norm(dim=1, keepdim=True)
env.torques = torch.randn(N, 12) * env.torque_limits * 0.6
env.actions, env.last_actions = torch.rand(N, 12) * 2 - 1, torch.rand(N, 12) * 2 - 1
env.last_dof_vel, env.last_root_vel = torch.randn(N, 12) * 6, torch.randn(N, 6)
env.last_feet_pos = env.feet_pos + torch.randn(N, 2, 3) * 0.01
env.feet_roll, env.feet_yaw = torch.zeros(N, 2), torch.zeros(N, 2)
env.feet_contact = torch.zeros(N, 2, dtype=torch.bool)
env.episode_length_buf = torch.randint(0, 1600, (N,)); env.episode_length_buf[:3] = torch.tensor([0, 1, 2])
env.cmd_resample_time = torch.randint(400, 600, (N,)); env.cmd_resample_time[5:8] = env.episode_length_buf[5:8]
env.base_mass_scaled = torch.rand(N, 4)
env.pushing_forces, env.pushing_torques = torch.zeros(N, 13, 3), torch.zeros(N, 13, 3)
env.pushing_forces[:, 0], env.pushing_torques[:, 0] = torch.randn(N, 3) * 10, torch.randn(N, 3) * 2
env.rew_buf = torch.zeros(N)
env.extras = {"rew_terms": {}}

inputs = {k: getattr(env, k).clone().numpy() for k in (
    "root_states", "body_states", "dof_pos", "dof_vel", "contact_forces", "commands", "gait_frequency", "gait_process", "filtered_lin_vel",
    "filtered_ang_vel", "base_lin_vel", "base_ang_vel", "projected_gravity", "torques", "actions", "last_actions", "last_dof_vel",
    "last_root_vel", "last_feet_pos", "episode_length_buf", "cmd_resample_time", "base_mass_scaled")}
inputs["push_force"], inputs["push_torque"] = env.pushing_forces[:, 0].numpy(), env.pushing_torques[:, 0].numpy()

out = {}
env._refresh_feet_state()                                           # t1.py:529-549
out["feet_roll"], out["feet_yaw"], out["feet_contact"] = env.feet_roll.numpy().copy(), env.feet_yaw.numpy().copy(), env.feet_contact.numpy().copy()
env._check_termination()                                            # t1.py:551-558
out["reset_buf"], out["time_out_buf"] = env.reset_buf.numpy().copy(), env.time_out_buf.numpy().copy()
env._prepare_reward_function()                                      # t1.py:274-292
env._compute_reward()                                               # t1.py:560-572
out["rew_buf"] = env.rew_buf.numpy().copy()
for name in env.reward_names:
    out["term_" + name] = env.extras["rew_terms"][name].numpy().copy()
    out["raw_" + name] = getattr(env, "_reward_" + name)().float().numpy().copy()
out["reward_names"] = np.array(env.reward_names)
out["reward_scales"] = np.array([env.reward_scales[n] for n in env.reward_names])
env._compute_observations()                                         # t1.py:574-603
out["obs_buf"], out["privileged_obs_buf"] = env.obs_buf.numpy().copy(), env.privileged_obs_buf.numpy().copy()
# PD + friction + clip, t1.py:446-448 restated by calling the same tensor expressions through a tiny shim of the loop body
kp = torch.tensor([200.0, 200, 200, 200, 50, 50] * 2) * (0.95 + 0.1 * torch.rand(N, 12))
kd = torch.tensor([5.0, 5, 5, 5, 1, 1] * 2) * (0.95 + 0.1 * torch.rand(N, 12))
fr = torch.rand(N, 12) * 2
tg = env.default_dof_pos + (torch.rand(N, 12) * 2 - 1)
tq = kp * (tg - env.dof_pos) - kd * env.dof_vel
fric = torch.min(fr, tq.abs()) * torch.sign(tq)
tq = torch.clip(tq - fric, min=-env.torque_limits, max=env.torque_limits)
inputs.update(pd_kp=kp.numpy(), pd_kd=kd.numpy(), pd_fric=fr.numpy(), pd_target=tg.numpy())
out["pd_torque"] = tq.numpy()
np.savez_compressed(os.path.join(HERE, "task_logic.npz"), **{"in_" + k: v for k, v in inputs.items()}, **{"out_" + k: v for k, v in out.items()},
                    cfg_soft=np.array([0.9, 0.8, 0.7]))
print("task_logic.npz:", len(inputs), "inputs,", len(out), "outputs; rewards:", list(env.reward_names))

# ---- apply_randomization (utils/utils.py:5-30): deterministic structure check via fixed torch seed
x = torch.randn(16, 5)
cases = {}
for dist in ("gaussian", "uniform"):
    for op in ("additive", "scaling"):
        torch.manual_seed(7)
        y, noise = ref_utils.apply_randomization(x, {"distribution": dist, "operation": op, "range": [0.3, 0.7]}, return_noise=True)
        cases[f"{dist}_{op}_y"], cases[f"{dist}_{op}_noise"] = y.numpy(), noise.numpy()
np.savez_compressed(os.path.join(HERE, "apply_randomization.npz"), x=x.numpy(), **cases)

# ---- command curriculum (t1.py:391-435): deterministic part via the reference's own methods; the two RNG sources inside
# `_resample_curriculum_commands` (torch.multinomial, torch_rand_float) are replaced by recorded draws
import envs.t1 as ref_t1  # noqa: E402

cfg["commands"]["curriculum"] = True
lv, av = cfg["commands"]["lin_vel_levels"], cfg["commands"]["ang_vel_levels"]
env.env_curriculum_level = torch.stack([torch.randint(-lv, lv + 1, (N,)), torch.randint(-av, av + 1, (N,))], dim=1)
env.env_curriculum_level[0] = torch.tensor([-lv, av]); env.env_curriculum_level[1] = torch.tensor([lv, -av])  # grid corners
env.curriculum_prob = torch.rand(2 * lv + 1, 2 * av + 1) * 1.05
env.episode_length_buf = torch.randint(1200, 1600, (N,))
env.filtered_lin_vel = env.commands + torch.randn(N, 3) * 0.25
env.filtered_ang_vel = env.commands[:, [2, 2, 2]] + torch.randn(N, 3) * 0.15
ids = torch.arange(0, N, 2)
cur_in = dict(curr_levels=env.env_curriculum_level.numpy().copy(), curr_prob=env.curriculum_prob.numpy().copy(), curr_ep_len=env.episode_length_buf.numpy().copy(),
              curr_filt_lin=env.filtered_lin_vel.numpy().copy(), curr_filt_ang=env.filtered_ang_vel.numpy().copy(), curr_cmd=env.commands.numpy().copy(),
              curr_ids=ids.numpy().copy())
env._update_curriculum(ids)
cur_out = dict(curr_prob_after=env.curriculum_prob.numpy().copy())
grid_idx = torch.randint(0, (2 * lv + 1) * (2 * av + 1), (len(ids),))
draws = [torch.rand(len(ids), 1) - 0.5, torch.rand(len(ids), 1) * 2 - 1, torch.rand(len(ids), 1) - 0.5]
it = iter(draws)
real_multinomial, real_rand = torch.multinomial, ref_t1.torch_rand_float
torch.multinomial = lambda p, n, replacement=True: grid_idx
ref_t1.torch_rand_float = lambda lo, hi, shape, device: next(it)
try:
    env._resample_curriculum_commands(ids)
finally:
    torch.multinomial, ref_t1.torch_rand_float = real_multinomial, real_rand
cur_in.update(curr_grid_idx=grid_idx.numpy(), curr_ux=draws[0][:, 0].numpy(), curr_uy=draws[1][:, 0].numpy(), curr_uyaw=draws[2][:, 0].numpy())
cur_out.update(curr_commands=env.commands.numpy().copy(), curr_levels_after=env.env_curriculum_level.numpy().copy(),
               curr_level_stats=np.array([float(env.mean_lin_vel_level), float(env.mean_ang_vel_level), float(env.max_lin_vel_level), float(env.max_ang_vel_level)]))
np.savez_compressed(os.path.join(HERE, "curriculum.npz"), **cur_in, **cur_out)
print("curriculum.npz written; successes:", int((cur_out["curr_prob_after"] != np.minimum(cur_in["curr_prob"], 1.0)).sum()), "cells changed")
