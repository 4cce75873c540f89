import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from booster_gym_amd.envs import T1
from booster_gym_amd.envs.gym_calls import GymCalls, gymapi, gymtorch
from booster_gym_amd.utils.config import load_cfg
from booster_gym_amd.utils.urdf import load_model
from oracle.dyn_ref import DynRef
n = 512
TH = float(os.environ.get("DBG_TH", "0.05")); SC = int(os.environ.get("DBG_SC", "0"))
cfg = load_cfg("T1", {"env.num_envs": n, "terrain.type": "plane", "rewards.terminate_height": TH, "asset.self_collisions": SC})
env = T1(cfg); gym = GymCalls(env)
flat_model = env.model
env.reset(); dev = env.device
g = torch.Generator(device="cpu").manual_seed(5)
for _ in range(15): env.step((0.2 * torch.randn(n, 12, generator=g)).to(dev))
env.common_step_counter = 7
push = torch.randn(n, 6, generator=g).to(dev) * torch.tensor([10, 10, 10, 2, 2, 2.0], device=dev)
env.set_field("pushing", push)
root0, q0, qd0 = env.root_states.clone(), env.dof_pos.clone(), env.dof_vel.clone()
kp, kd, fric = env.get_field("dof_stiffness"), env.get_field("dof_damping"), env.get_field("dof_friction")
delay = env.get_field("delay_steps").view(n, 1)
last_tgt = env.get_field("last_dof_targets").clone()
limit = torch.tensor(flat_model.dof_effort, dtype=torch.float32, device=dev)
default = env.default_dof_pos.view(1, 12)
actions = (0.5 * torch.randn(n, 12, generator=g)).to(dev)
sim = gym.sim
root_t = gymtorch.wrap_tensor(gym.acquire_actor_root_state_tensor(sim)); dof_t = gymtorch.wrap_tensor(gym.acquire_dof_state_tensor(sim)).view(n, 12, 2)
root_t.copy_(root0); dof_t[..., 0] = q0; dof_t[..., 1] = qd0
clip = cfg["normalization"]["clip_actions"]
acts = torch.clip(actions, -clip, clip)
dof_targets = default + cfg["control"]["action_scale"] * acts
pf = torch.zeros(n, 13, 3, device=dev); pt = torch.zeros(n, 13, 3, device=dev)
pf[:, 0] = push[:, :3]; pt[:, 0] = push[:, 3:]
gym.apply_rigid_body_force_tensors(sim, gymtorch.unwrap_tensor(pf), gymtorch.unwrap_tensor(pt), gymapi.LOCAL_SPACE)
ref = DynRef(flat_model, feet_edge_pos=cfg["asset"]["feet_edge_pos"], phys={"self_collisions": 1 - SC})
E = int(sys.argv[1]) if len(sys.argv) > 1 else 389
f64 = lambda t: t.cpu().numpy().astype(np.float64)
r32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
par = dict(mass_scale=r32(env._mass_scale[E]), com_off=r32(env._com_off[E]), foot_mat=r32(env._foot_mat[E]).reshape(6))
ro, qo, vo = f64(root0[E]).copy(), f64(q0[E]).copy(), f64(qd0[E]).copy()
lt_o = f64(last_tgt[E]).copy()
com0 = np.array(ref.model.com[0][:]) + par["com_off"][0]
wr = f64(push[E]).copy(); wr[3:] += np.cross(com0, wr[:3])
for i in range(10):
    last_tgt = torch.where(delay == i, dof_targets, last_tgt)
    tq = kp * (last_tgt - dof_t[..., 0]) - kd * dof_t[..., 1]
    fr = torch.min(fric, tq.abs()) * torch.sign(tq)
    tq = torch.clip(tq - fr, min=-limit, max=limit)
    gym.set_dof_actuation_force_tensor(sim, gymtorch.unwrap_tensor(tq))
    # oracle substep from ITS OWN state with ITS OWN torque
    if int(delay[E]) == i: lt_o = f64(dof_targets[E]).copy()
    to = f64(kp[E]) * (lt_o - qo) - f64(kd[E]) * vo
    fo = np.minimum(f64(fric[E]), np.abs(to)) * np.sign(to)
    to = np.clip(to - fo, -f64(limit), f64(limit))
    sf = ref.self_contact_forces(ro, qo, vo)
    cfo = ref.step(ro, qo, vo, to, base_wrench=wr if i == 0 else None, **par)
    gym.simulate(sim)
    if os.environ.get("DBG_V"): print(i, "granular-oracle dq", np.abs(f64(dof_t[E, :, 0]) - qo).max(), "self |F|", np.abs(sf).max(), "foot cf", np.round(cfo[[6, 12]], 1).tolist(), "z", ro[2])
_, _, done, _ = env.step(actions)
print("TH", TH, "SC", SC, "fused-oracle dq", np.abs(f64(env.dof_pos[E]) - qo).max(), "fused-granular", np.abs(f64(env.dof_pos[E]) - f64(dof_t[E, :, 0])).max(), "done", bool(done[E]))
print("dq per joint", (f64(env.dof_pos[E]) - qo).round(6).tolist())
print("dqd per joint", (f64(env.dof_vel[E]) - vo).round(4).tolist())
print("droot", (f64(env.root_states[E]) - ro).round(6).tolist())
print("delay", int(delay[E]), "ep_len", int(env.get_field("episode_length_buf")[E]), "fused mean torque", f64(env.get_field("torques")[E]).round(3).tolist())
print("feet contact forces fused", f64(env.get_field("feet_contact_forces")[E]).round(1).tolist())
err = (env.dof_pos - dof_t[..., 0]).abs().max(dim=1).values
print("envs with fused-granular dq > 1e-4:", torch.nonzero((err > 1e-4) & ~done.bool()).flatten().tolist())
