"""CPU tests: the oracle (oracle/) against the golden vectors produced by the reference's own code (tests/golden/*.npz),
and the build-owned known-answer tests for the dynamics, which the reference pins nothing for."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import task_ref as tr

HERE = os.path.dirname(os.path.abspath(__file__))
G = lambda name: np.load(os.path.join(HERE, "golden", name), allow_pickle=False)


# ------------------------------------------------------------------ task logic vs reference outputs
@pytest.fixture(scope="module")
def task():
    d = G("task_logic.npz")
    i = {k[3:]: d[k] for k in d.files if k.startswith("in_")}
    o = {k[4:]: d[k] for k in d.files if k.startswith("out_")}
    t = G("terrain_heights.npz")
    terrain = dict(height_field_raw=t["height_field_raw"], hscale=float(t["hscale"]), vscale=float(t["vscale"]), border_px=int(t["border_px"]))
    return i, o, terrain


def test_terrain_heights_matches_reference():
    t = G("terrain_heights.npz")
    terrain = dict(height_field_raw=t["height_field_raw"], hscale=float(t["hscale"]), vscale=float(t["vscale"]), border_px=int(t["border_px"]))
    h = tr.terrain_heights(terrain, t["xy"])
    assert np.allclose(h, t["heights"], atol=5e-6)  # the reference divides float32 positions by hscale in float32


def test_feet_state_matches_reference(task):
    i, o, terrain = task
    feet = i["body_states"][:, [6, 12]]
    edges = [[0.1215, 0.05, -0.03], [0.1215, -0.05, -0.03], [-0.1015, 0.05, -0.03], [-0.1015, -0.05, -0.03]]
    roll, yaw, contact = tr.feet_state(feet[:, :, 0:3].astype(np.float64), feet[:, :, 3:7].astype(np.float64), edges, terrain)
    assert np.allclose(roll, o["feet_roll"], atol=2e-6) and np.allclose(yaw, o["feet_yaw"], atol=2e-6)
    assert (contact == o["feet_contact"]).all()


REW_CFG = dict(tracking_sigma=0.25, base_height_target=0.68, soft_dof_pos_limit=0.9, soft_dof_vel_limit=0.8, soft_torque_limit=0.7, swing_period=0.2,
               feet_distance_ref=0.2, episode_length_s=30.0, terminate_height=0.45, terminate_vel=50.0, only_positive_rewards=True)
LIMITS = dict(dof_pos_limits=np.stack([np.array([-1.8, -0.3, -1, 0, -0.87, -0.44, -1.8, -1.57, -1, 0, -0.87, -0.44]),
                                       np.array([1.57, 1.57, 1, 2.34, 0.35, 0.44, 1.57, 0.3, 1, 2.34, 0.35, 0.44])], axis=1),
              dof_vel_limits=np.array([12.5, 10.9, 10.9, 11.7, 18.8, 12.4] * 2), torque_limits=np.array([45.0, 30, 30, 60, 24, 15] * 2))


def test_termination_matches_reference(task):
    i, o, terrain = task
    reset, tout = tr.check_termination(i["root_states"].astype(np.float64), i["episode_length_buf"], i["cmd_resample_time"], terrain, REW_CFG, 0.02)
    assert (reset == o["reset_buf"]).all() and (tout == o["time_out_buf"]).all()
    assert reset.any() and not reset.all()


def test_all_26_reward_terms_match_reference(task):
    i, o, terrain = task
    feet = i["body_states"][:, [6, 12]].astype(np.float64)
    s = {k: v.astype(np.float64) if v.dtype.kind == "f" else v for k, v in i.items()}
    s.update(feet_pos=feet[:, :, 0:3], feet_roll=o["feet_roll"].astype(np.float64), feet_yaw=o["feet_yaw"].astype(np.float64),
             feet_contact=o["feet_contact"], penalized_contact_indices=[0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11])
    terms = tr.reward_terms(s, REW_CFG, 0.02, LIMITS, terrain)
    names = [str(x) for x in o["reward_names"]]
    assert sorted(names) == sorted(tr.REWARD_NAMES) and len(names) == 26
    for name in names:
        ref = o["raw_" + name].astype(np.float64)
        assert np.allclose(terms[name], ref, rtol=2e-5, atol=2e-5 * max(1.0, np.abs(ref).max())), name
    scales = dict(zip(names, o["reward_scales"].astype(np.float64)))
    tot, scaled = tr.total_reward(terms, scales, True)
    assert np.allclose(tot, o["rew_buf"], rtol=1e-4, atol=1e-4)
    for name in names:
        assert np.allclose(scaled[name], o["term_" + name], rtol=1e-4, atol=1e-5 * max(1.0, np.abs(o["term_" + name]).max()))


def test_observations_match_reference(task):
    i, o, terrain = task
    s = {k: v.astype(np.float64) if v.dtype.kind == "f" else v for k, v in i.items()}
    norm = dict(gravity=1.0, lin_vel=1.0, ang_vel=1.0, dof_pos=1.0, dof_vel=0.1, push_force=0.1, push_torque=0.5)
    default = np.array([-0.2, 0, 0, 0.4, -0.25, 0] * 2)
    obs, priv = tr.compute_observations(s, norm, default, terrain, noisy=None)
    assert obs.shape == (64, 47) and priv.shape == (64, 14)
    assert np.allclose(obs, o["obs_buf"], atol=2e-6) and np.allclose(priv, o["privileged_obs_buf"], atol=2e-5)


def test_pd_torque_matches_reference(task):
    i, o, _ = task
    t = tr.pd_torque(i["pd_kp"], i["pd_kd"], i["pd_fric"], LIMITS["torque_limits"], i["pd_target"], i["dof_pos"], i["dof_vel"])
    assert np.allclose(t, o["pd_torque"], atol=1e-4)


def test_command_curriculum_matches_reference():
    """t1.py:391-435 via the reference's own `_update_curriculum` / `_resample_curriculum_commands` (RNG draws recorded in the fixture)."""
    d = G("curriculum.npz")
    cm = dict(lin_vel_levels=10, ang_vel_levels=10, update_rate=0.1, lin_vel_x_resolution=0.2, lin_vel_y_resolution=0.1, ang_vel_resolution=0.2,
              episode_length_toler=0.1, lin_vel_x_toler=0.4, lin_vel_y_toler=0.2, ang_vel_yaw_toler=0.2)
    prob = tr.update_curriculum(d["curr_prob"].astype(np.float64), d["curr_levels"], d["curr_ep_len"], d["curr_filt_lin"].astype(np.float64),
                                d["curr_filt_ang"].astype(np.float64), d["curr_cmd"].astype(np.float64), d["curr_ids"], cm, REW_CFG, 0.02)
    assert np.allclose(prob, d["curr_prob_after"], atol=1e-6)
    assert (np.abs(prob - np.minimum(d["curr_prob"], 1.0)) > 1e-6).sum() >= 10  # the update did something
    lin, ang, cmd = tr.curriculum_commands(d["curr_grid_idx"], d["curr_ux"].astype(np.float64), d["curr_uy"].astype(np.float64),
                                           d["curr_uyaw"].astype(np.float64), cm, 21)
    ids = d["curr_ids"]
    assert np.allclose(cmd, d["curr_commands"][ids], atol=1e-6)
    assert (lin == d["curr_levels_after"][ids, 0]).all() and (ang == d["curr_levels_after"][ids, 1]).all()
    lv = d["curr_levels_after"]
    assert np.allclose([np.abs(lv[:, 0]).mean(), np.abs(lv[:, 1]).mean(), np.abs(lv[:, 0]).max(), np.abs(lv[:, 1]).max()], d["curr_level_stats"], atol=1e-5)


def test_recalled_isaacgym_quaternion_helpers_agree_with_scipy():
    """The task-logic fixtures are generated through a stand-in `isaacgym.torch_utils` whose helpers are restated from recall
    (tests/golden/_stub/isaacgym/torch_utils.py); the oracle carries its own copies.  A wrong recall would make oracle and fixture agree with each other
    on wrong numbers, so both are held here to an implementation neither of them was written from -- scipy.spatial.transform.Rotation, xyzw quaternions:
    quat_rotate / quat_rotate_inverse = R v / R^T v, get_euler_xyz = the roll-pitch-yaw angles of R = Rz(yaw) Ry(pitch) Rx(roll) folded into [0, 2 pi)
    (the convention the reference's own uses need: t1.py:541-547 wraps them back to [-pi, pi), play_mujoco.py:282-297 pins quat_rotate_inverse),
    quat_from_euler_xyz = its inverse."""
    import importlib.util

    from scipy.spatial.transform import Rotation

    spec = importlib.util.spec_from_file_location("stub_torch_utils", os.path.join(HERE, "golden", "_stub", "isaacgym", "torch_utils.py"))
    stub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stub)
    rng = np.random.default_rng(7)
    q = rng.normal(size=(500, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    v = rng.normal(size=(500, 3))
    R = Rotation.from_quat(q)  # scipy: scalar-last (x, y, z, w), like Isaac Gym
    tq, tv = torch.tensor(q), torch.tensor(v)
    assert np.allclose(stub.quat_rotate(tq, tv).numpy(), R.apply(v), atol=1e-12) and np.allclose(tr.quat_rotate(q, v), R.apply(v), atol=1e-12)
    assert np.allclose(stub.quat_rotate_inverse(tq, tv).numpy(), R.inv().apply(v), atol=1e-12) and np.allclose(tr.quat_rotate_inverse(q, v), R.inv().apply(v), atol=1e-12)
    rpy = R.as_euler("xyz")  # extrinsic x-y-z = Rz(yaw) Ry(pitch) Rx(roll)
    keep = np.abs(np.abs(rpy[:, 1]) - np.pi / 2) > 1e-3  # (away from the gimbal lock, where the split between roll and yaw is a convention)
    wrap = lambda a: (a + np.pi) % (2 * np.pi) - np.pi
    for got in (np.stack([a.numpy() for a in stub.get_euler_xyz(tq)], axis=1), np.stack(tr.get_euler_xyz(q), axis=1)):
        assert got.min() >= 0 and got.max() < 2 * np.pi + 1e-12
        assert np.abs(wrap(got[keep] - rpy[keep])).max() < 1e-9
    back = stub.quat_from_euler_xyz(torch.tensor(rpy[:, 0]), torch.tensor(rpy[:, 1]), torch.tensor(rpy[:, 2])).numpy()
    assert np.allclose(np.abs(np.sum(back * q, axis=1)), 1.0, atol=1e-9)  # the same rotation (q and -q are)


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    o = tr.philox4x32_10(0, 0, np.uint32(0), np.uint32(0), np.uint32(0), np.uint32(0))
    assert [int(x) for x in o] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    o = tr.philox4x32_10(0xFFFFFFFF, 0xFFFFFFFF, np.uint32(0xFFFFFFFF), np.uint32(0xFFFFFFFF), np.uint32(0xFFFFFFFF), np.uint32(0xFFFFFFFF))
    assert [int(x) for x in o] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    o = tr.philox4x32_10(0xA4093822, 0x299F31D0, np.uint32(0x243F6A88), np.uint32(0x85A308D3), np.uint32(0x13198A2E), np.uint32(0x03707344))
    assert [int(x) for x in o] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


# ------------------------------------------------------------------ PPO math vs reference outputs
def test_gae_and_surrogate_match_reference():
    from oracle.ppo_ref import discount_values, surrogate_loss

    d = G("ppo_gae.npz")
    t = lambda k: torch.tensor(d[k])
    adv = discount_values(t("rewards"), t("dones") | t("time_outs"), t("values"), t("last_values"), float(d["gamma"]), float(d["lam"]))
    assert torch.allclose(adv, t("advantages"), atol=1e-6)
    sl = surrogate_loss(t("sl_old_logp"), t("sl_logp"), t("sl_adv"))
    assert abs(sl.item() - float(d["sl_value"])) < 1e-6


def test_ppo_epoch_restatement_matches_reference():
    """oracle/ppo_ref.py's first mini-epoch reproduces the reference-computed losses, advantages and parameter gradients."""
    from booster_gym_amd.utils.model import ActorCritic
    from oracle.ppo_ref import ppo_update_reference

    d = G("ppo_epoch.npz")
    model = ActorCritic(12, 47, 14)
    model.load_state_dict({k[3:]: torch.tensor(d[k]) for k in d.files if k.startswith("sd_")})
    t = lambda k: torch.tensor(d[k])

    class OldDist:  # the fixture's "old" policy differs from the current one; inject it
        pass

    rec = {}
    rewards = t("rewards").clone()
    # ppo_update_reference recomputes old_dist from the model; emulate the fixture by calling its pieces directly
    import oracle.ppo_ref as pr
    import torch.nn.functional as F

    vals = model.est_value(t("obses"), t("priv")); lastv = model.est_value(t("last_obs"), t("last_priv"))
    with torch.no_grad():
        rewards[t("time_outs")] = vals[t("time_outs")]
        adv = pr.discount_values(rewards, t("dones") | t("time_outs"), vals, lastv, 0.995, 0.95)
        ret = vals + adv
        advn = (adv - adv.mean()) / (adv.std() + 1e-8)
    dist = model.act(t("obses"))
    logp = dist.log_prob(t("actions")).sum(-1)
    loss = F.mse_loss(vals, ret) + pr.surrogate_loss(t("old_logp"), logp, advn)
    loss = loss + torch.clip(dist.loc - 1.0, min=0.0).square().mean() + torch.clip(dist.loc + 1.0, max=0.0).square().mean() - 0.01 * dist.entropy().sum(-1).mean()
    loss.backward()
    assert torch.allclose(adv, t("advantages"), atol=1e-5) and torch.allclose(rewards, t("rewards_after"), atol=1e-6)
    for k, p in model.named_parameters():
        assert torch.allclose(p.grad, t("grad_" + k), rtol=1e-4, atol=1e-6), k
    # and the packaged loop runs end to end
    m2 = ActorCritic(12, 47, 14)
    stats, lr = ppo_update_reference(m2, torch.optim.Adam(m2.parameters(), lr=1e-5), t("obses"), t("priv"), t("actions"), t("rewards").clone(),
                                     t("dones"), t("time_outs"), t("last_obs"), t("last_priv"), mini_epochs=2, record=rec)
    assert np.isfinite(list(stats.values())).all() and "grads" in rec


# ------------------------------------------------------------------ dynamics known-answer tests (build-owned)
@pytest.fixture(scope="module")
def dyn(flat_model):
    from oracle.dyn_ref import DynRef

    return DynRef(flat_model)


def test_loader_kat_collapsed_trunk(flat_model):
    """Collapsed trunk inertial vs reference resources/T1/T1_locomotion.xml:38 (mass, com, principal inertias)."""
    m = flat_model
    assert m.num_bodies == 13 and m.num_dofs == 12
    assert abs(m.mass[0] - 19.4304) < 1e-3 and abs(m.mass.sum() - 31.6144) < 1e-3
    assert np.allclose(m.com[0], [0.054281, 8.47449e-06, 0.0893932], atol=1e-6)
    I = m.inertia[0]
    T = np.array([[I[0], I[3], I[4]], [I[3], I[1], I[5]], [I[4], I[5], I[2]]])
    assert np.allclose(sorted(np.linalg.eigvalsh(T)), sorted([0.502019, 0.3531, 0.207324]), atol=2e-6)
    assert m.joint_axis.tolist() == [0, 2, 1, 3, 2, 2, 1, 2, 1, 3, 2, 2, 1]
    assert m.dof_effort.tolist() == [45, 30, 30, 60, 24, 15] * 2


def _mjcf_fixture():
    return G("mjcf_model.npz")


def _assert_model_matches_mjcf(mass, com, inertia6, body_pos, parent, joint_axis, lower, upper, names, dof_names, g):
    """Every body, hinge and limit of a flat model against the numbers of the reference's MJCF (tests/golden/make_model_fixture.py)."""
    assert list(names) == g["body_names"].tolist() and list(dof_names) == g["joint_names"].tolist()
    assert list(parent) == g["parent"].tolist()
    # the MJCF prints 6 significant digits (the collapsed trunk is a sum of 12 URDF links: 19.43035778 printed as 19.4304)
    assert np.allclose(mass, g["mass"], rtol=5e-6, atol=0) and abs(np.sum(mass) - 31.6144) < 1e-4
    assert np.allclose(com, g["inertial_pos"], rtol=5e-6, atol=5e-10)
    # body 0 sits at z = 0.7 in the MJCF's world (the keyframe height); its flat-model origin is the free joint's: zero
    assert np.allclose(np.asarray(body_pos)[1:], g["body_pos"][1:], rtol=0, atol=1e-9) and np.allclose(g["body_pos"][0], [0.0, 0.0, 0.7], atol=1e-7)
    I = np.asarray(inertia6)
    T = np.stack([np.array([[r[0], r[3], r[4]], [r[3], r[1], r[5]], [r[4], r[5], r[2]]]) for r in I])
    # the MJCF prints quat and diaginertia with 6 significant digits: tensors agree to that (legs 4e-8, the 12-link trunk 3e-7)
    assert np.abs(T[1:] - g["inertia_tensor"][1:]).max() < 1e-7 and np.abs(T[0] - g["inertia_tensor"][0]).max() < 1e-6
    for b in range(13):
        assert np.allclose(np.sort(np.linalg.eigvalsh(T[b])), np.sort(g["diaginertia"][b]), rtol=2e-5, atol=2e-9), b
    ax = np.asarray(joint_axis)
    assert ax[0] == 0 and g["joint_body"].tolist() == list(range(1, 13))
    for j in range(12):
        e = np.zeros(3); e[ax[j + 1] - 1] = 1.0
        assert g["joint_axis"][j].tolist() == e.tolist(), j
    assert np.allclose(lower, g["joint_range"][:, 0], atol=1e-9) and np.allclose(upper, g["joint_range"][:, 1], atol=1e-9)


def test_flat_model_matches_every_body_of_the_reference_mjcf(flat_model):
    """SURVEY 8(c) KAT 1 in full: the packaged flat model (what every kernel and the oracle run on) against resources/T1/T1_locomotion.xml:36-139 --
    all 13 bodies (parent, frame offset, mass, centre of mass, full inertia tensor), the 12 hinges (axis, range), the actuator limits and the 7
    collision primitives.  The MJCF is the model play_mujoco.py:717-756 steps: the only reference-held statement of the robot's inertial physics."""
    m, g = flat_model, _mjcf_fixture()
    _assert_model_matches_mjcf(m.mass, m.com, m.inertia, m.body_pos, m.parent, m.joint_axis, m.dof_lower, m.dof_upper, m.body_names, m.dof_names, g)
    # torque limits: the training env clips to the URDF effort (t1.py:67,448), the MuJoCo player to ctrlrange (play_mujoco.py:751-755); they
    # differ for Hip_Roll (30 / 45) and Knee (60 / 65) and agree elsewhere
    cr = g["ctrlrange"]
    assert np.allclose(cr[:, 0], -cr[:, 1])
    diff = {j: (m.dof_effort[j], cr[j, 1]) for j in range(12) if m.dof_effort[j] != cr[j, 1]}
    assert diff == {1: (30.0, 45.0), 3: (60.0, 65.0), 7: (30.0, 45.0), 9: (60.0, 65.0)}
    # collision primitives: MuJoCo half-sizes vs the URDF <collision> extents the flat model keeps; the ground plane is geom 0 of the world
    assert g["geom_body"][0] == -1 and g["geom_type"][0] == 0
    mj = sorted((int(b), int(t), tuple(np.round(s, 9)), tuple(np.round(p, 9))) for b, t, s, p in
                zip(g["geom_body"][1:], g["geom_type"][1:], g["geom_halfsize"][1:], g["geom_pos"][1:]))
    ours = []
    for sh in m.shapes:
        if sh["type"] == "box":
            ours.append((int(sh["body"]), 1, tuple(np.round(0.5 * np.array(sh["size"]), 9)), tuple(np.round(sh["pos"], 9))))
        else:
            assert sh["type"] == "cylinder"
            ours.append((int(sh["body"]), 2, (round(sh["size"][0], 9), round(0.5 * sh["size"][1], 9), 0.0), tuple(np.round(sh["pos"], 9))))
    assert sorted(ours) == mj and len(mj) == 7
    # the sole corners the contact model uses (asset.feet_edge_pos, T1.yaml:79-82) are the bottom corners of the foot boxes
    from booster_gym_amd.utils.config import load_cfg

    fe = np.array(load_cfg("T1", {})["asset"]["feet_edge_pos"])
    for b in (6, 12):
        k = [i for i in range(len(g["geom_body"])) if g["geom_body"][i] == b][0]
        c, h = g["geom_pos"][k], g["geom_halfsize"][k]
        corners = sorted((c[0] + sx * h[0], c[1] + sy * h[1], c[2] - h[2]) for sx in (-1, 1) for sy in (-1, 1))
        assert np.allclose(sorted(map(tuple, fe)), corners, atol=1e-9)
    # contact spheres and self-collision capsules are derived from exactly these primitives
    caps = m.self_collision_capsules([6, 12])
    assert [c[0][0] for c in caps] == [4, 10] and all(abs(c[0][3] - 0.05) < 1e-12 and abs(c[1][3] - 0.05) < 1e-12 for c in caps)


def test_urdf_loaders_match_every_body_of_the_reference_mjcf():
    """The same comparison for both URDF loaders (utils/urdf.py and bg_model_load_urdf through the C ABI) on the reference's URDF: fixed-joint
    collapsing of the 24-link URDF must land on the MJCF's 13 bodies.  Needs the reference tree (this container; skipped on the GPU box)."""
    path = "/root/reference/resources/T1/T1_locomotion.urdf"
    if not os.path.isfile(path):
        pytest.skip("reference asset not on this machine (GPU box)")
    import ctypes as C

    from booster_gym_amd import _lib
    from booster_gym_amd.utils.urdf import load_urdf

    g = _mjcf_fixture()
    m = load_urdf(path)
    _assert_model_matches_mjcf(m.mass, m.com, m.inertia, m.body_pos, m.parent, m.joint_axis, m.dof_lower, m.dof_upper, m.body_names, m.dof_names, g)
    lib = _lib.load()
    opt = _lib.AssetOptions()
    opt.collapse_fixed_joints, opt.body_contacts, opt.self_collisions = 1, 1, 1
    opt.foot_names[0], opt.foot_names[1] = b"left_foot_link", b"right_foot_link"
    h = C.c_void_p()
    assert lib.bg_model_load_urdf(path.encode(), C.byref(opt), C.byref(h)) == 0, lib.bg_last_error()
    d = _lib.ModelDesc()
    _lib.check(lib.bg_model_get(h, C.byref(d)))
    f = lambda a: np.array([list(r) if hasattr(r, "__len__") else r for r in a], dtype=np.float64)
    names = [lib.bg_model_body_name(h, i).decode() for i in range(d.num_bodies)]
    dofs = [lib.bg_model_dof_name(h, j).decode() for j in range(d.num_dofs)]
    # the native model holds fp32: one rounding of the MJCF's digits
    g32 = {k: (g[k].astype(np.float32).astype(np.float64) if g[k].dtype == np.float64 else g[k]) for k in g.files}
    _assert_model_matches_mjcf(f(d.mass)[:13], f(d.com)[:13], f(d.inertia)[:13], f(d.body_pos)[:13], list(d.parent)[:13], list(d.joint_axis)[:13],
                               f(d.dof_lower)[:12], f(d.dof_upper)[:12], names, dofs, g32)
    lib.bg_model_destroy(h)


def test_urdf_loader_on_reference_asset_if_present(flat_model):
    from booster_gym_amd.utils.urdf import load_urdf

    path = "/root/reference/resources/T1/T1_locomotion.urdf"
    if not os.path.isfile(path):
        pytest.skip("reference asset not on this machine (GPU box)")
    m = load_urdf(path)
    assert m.body_names == flat_model.body_names
    assert np.allclose(m.mass, flat_model.mass) and np.allclose(m.inertia, flat_model.inertia) and np.allclose(m.com, flat_model.com)


@pytest.fixture(scope="module")
def dyn_smooth(flat_model):
    """The smooth dynamics alone: leg-against-leg contacts off (random joint angles cross the legs)."""
    from oracle.dyn_ref import DynRef

    return DynRef(flat_model, phys={"self_collisions": 0})


def test_aba_equals_inverse_dynamics(dyn_smooth, flat_model):
    """ABA forward dynamics satisfies the independent RNEA inverse dynamics: residual < 1e-9 (double), with randomised inertias."""
    dyn = dyn_smooth
    m, rng, worst = flat_model, np.random.default_rng(0), 0.0
    for _ in range(500):
        root = np.zeros(13); root[2] = 5.0
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, 1.0)
        root[3:6], root[6] = ax * np.sin(ang / 2), np.cos(ang / 2)
        root[7:13] = rng.normal(size=6)
        q, qd = rng.uniform(m.dof_lower, m.dof_upper), rng.normal(size=12)
        tau, w = rng.uniform(-m.dof_effort, m.dof_effort), rng.normal(size=6) * 10
        ms, co = rng.uniform(0.8, 1.2, 13), rng.uniform(-0.05, 0.05, (13, 3))
        qacc, _ = dyn.forward(root, q, qd, tau, base_wrench=w, mass_scale=ms, com_off=co)
        res = dyn.inverse(root, q, qd, qacc, mass_scale=ms, com_off=co)
        worst = max(worst, np.abs(res - np.concatenate([w[3:], w[:3], tau])).max())
    assert worst < 1e-9


def _airborne_state(m, rng):
    root = np.zeros(13); root[2] = 5.0
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, 1.0)
    root[3:6], root[6] = ax * np.sin(ang / 2), np.cos(ang / 2)
    root[7:13] = rng.normal(size=6)
    return root, rng.uniform(m.dof_lower, m.dof_upper), rng.normal(size=12), rng.uniform(-m.dof_effort, m.dof_effort)


def test_self_collision_forces_are_internal(dyn, dyn_smooth, flat_model):
    """Leg-against-leg contacts (reference: self-collision enabled, envs/T1.yaml:69, envs/t1.py:128).  They are INTERNAL forces: with them on,
    the independent RNEA still returns the applied base wrench as the net force on the system (linear and angular momentum conserved), the
    contact rows of the two legs cancel, and only the joint rows of the residual change.  Hip rolls drawn inwards cross the legs in most samples."""
    m, rng = flat_model, np.random.default_rng(5)
    active, worst_base, worst_sum = 0, 0.0, 0.0
    for _ in range(300):
        root, q, qd, tau = _airborne_state(m, rng)
        q[[1, 7]] = rng.uniform(-0.3, 0.0), rng.uniform(0.0, 0.3)  # hip rolls inwards
        q[[0, 6]] = rng.uniform(-0.6, 0.2, 2)
        q[[3, 9]] = rng.uniform(0.0, 0.8, 2)
        w = rng.normal(size=6) * 10
        qacc, cf = dyn.forward(root, q, qd, tau, base_wrench=w)
        qacc0, cf0 = dyn_smooth.forward(root, q, qd, tau, base_wrench=w)
        res = dyn.inverse(root, q, qd, qacc)
        assert np.abs(cf0).max() == 0.0
        if np.abs(cf).max() > 0:
            active += 1
            assert np.abs(res[6:] - tau).max() > 1e-6 and np.abs(qacc - qacc0).max() > 1e-6
            assert np.abs(cf[[0, 1, 2, 3, 5, 7, 8, 9, 11]]).max() == 0.0  # only the shanks (4, 10) and the feet (6, 12) carry capsules
        else:
            assert np.abs(qacc - qacc0).max() == 0.0
        worst_base = max(worst_base, np.abs(res[:6] - np.concatenate([w[3:], w[:3]])).max() / max(1.0, np.abs(cf).max()))
        worst_sum = max(worst_sum, np.abs(cf.sum(axis=0)).max() / max(1.0, np.abs(cf).max()))
    assert active > 100, active
    assert worst_base < 1e-9 and worst_sum < 1e-12, (worst_base, worst_sum)


def test_self_collision_pushes_crossed_feet_apart(dyn, dyn_smooth, flat_model):
    """Feet pushed into each other sideways (hip roll inwards on both legs, zero velocities): the contact accelerates the hip-roll joints
    outwards, the force grows with the penetration and vanishes once the capsules are clear of each other."""
    m = flat_model
    root = np.zeros(13); root[2] = 5.0; root[6] = 1.0
    last = None
    for roll in (0.0, 0.14, 0.16, 0.18):
        q = np.array([-0.2, -roll, 0, 0.4, -0.25, roll, -0.2, roll, 0, 0.4, -0.25, -roll])  # left hip roll negative = inwards
        qacc, cf = dyn.forward(root, q, np.zeros(12), np.zeros(12))
        qacc0, _ = dyn_smooth.forward(root, q, np.zeros(12), np.zeros(12))
        f = cf[6, 1]  # lateral force on the left foot, trunk frame = world frame here
        if roll == 0.0:
            assert np.abs(cf).max() == 0.0
        else:
            assert f > 0 and cf[12, 1] < 0 and abs(cf[6, 1] + cf[12, 1] + cf[4, 1] + cf[10, 1]) < 1e-9
            assert qacc[6 + 1] - qacc0[6 + 1] > 0 and qacc[6 + 7] - qacc0[6 + 7] < 0  # left hip roll pushed positive (outwards), right negative
            if last is not None:
                assert f > last
            last = f


def test_self_collision_closest_points_regularised(dyn):
    """The narrow phase: (1) on random segment pairs the regularised closest points give the true segment distance (brute force on a grid) to
    second order; (2) for PARALLEL segments, where the plain problem has an interval of minimisers, the answer is unique, sits in the middle
    of the overlap and moves continuously under a rounding-size rotation of one segment."""
    rng = np.random.default_rng(9)
    g = np.linspace(0.0, 1.0, 201)
    for _ in range(200):
        a1, a2 = rng.normal(size=3) * 0.1, rng.normal(size=3) * 0.1
        b1, b2 = a1 + rng.normal(size=3) * 0.1, a2 + rng.normal(size=3) * 0.1
        s, t = dyn.segment_closest(a1, b1, a2, b2)
        d = np.linalg.norm(a1 + s * (b1 - a1) - a2 - t * (b2 - a2))
        P1, P2 = a1 + g[:, None] * (b1 - a1), a2 + g[:, None] * (b2 - a2)
        best = np.linalg.norm(P1[:, None, :] - P2[None, :, :], axis=2).min()
        assert d < best + 5e-3 * max(np.linalg.norm(b1 - a1), np.linalg.norm(b2 - a2)) and d > best - 1e-3
    a1, b1 = np.array([0.0, 0.05, 0.0]), np.array([0.2, 0.05, 0.0])
    a2, b2 = np.array([0.1, -0.05, 0.0]), np.array([0.3, -0.05, 0.0])
    s, t = dyn.segment_closest(a1, b1, a2, b2)
    assert abs((a1 + s * (b1 - a1))[0] - 0.15) < 5e-4 and abs((a2 + t * (b2 - a2))[0] - 0.15) < 5e-4  # middle of the overlap [0.1, 0.2]
    for eps, tol in ((1e-7, 1e-4), (-1e-7, 1e-4), (1e-5, 5e-3)):  # 1e-7: fp32 rounding size -> the contact point moves by < 20 micrometres
        b2e = b2 + np.array([0.0, eps, eps])
        s2, t2 = dyn.segment_closest(a1, b1, a2, b2e)
        assert abs(s2 - s) < tol and abs(t2 - t) < tol


def test_body_wrench_semantics(dyn, flat_model):
    """Per-body applied forces (gym.apply_rigid_body_force_tensors LOCAL_SPACE, t1.py:522-527): (1) a force on the trunk row acts at the
    trunk's centre of mass = base wrench (f, t + c x f); (2) m_i * (-g) on every body cancels gravity exactly; (3) the body-state rows
    agree with the pose query and with the root state."""
    from oracle.dyn_ref import DynRef

    m, rng = flat_model, np.random.default_rng(2)
    d0 = DynRef(m, phys={"g": (0.0, 0.0, 0.0)})
    for _ in range(50):
        root = np.zeros(13); root[2] = 5.0
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, 1.0)
        root[3:6], root[6] = ax * np.sin(ang / 2), np.cos(ang / 2)
        root[7:13] = rng.normal(size=6)
        q, qd = rng.uniform(m.dof_lower, m.dof_upper), rng.normal(size=12)
        tau = rng.uniform(-m.dof_effort, m.dof_effort)
        ms, co = rng.uniform(0.8, 1.2, 13), rng.uniform(-0.05, 0.05, (13, 3))
        f, t = rng.normal(size=3) * 10, rng.normal(size=3) * 2
        bf, bt = np.zeros((13, 3)), np.zeros((13, 3)); bf[0], bt[0] = f, t
        a1, _ = dyn.forward_bw(root, q, qd, tau, bf, bt, mass_scale=ms, com_off=co)
        a2, _ = dyn.forward(root, q, qd, tau, base_wrench=np.concatenate([f, t + np.cross(m.com[0] + co[0], f)]), mass_scale=ms, com_off=co)
        assert np.abs(a1 - a2).max() < 1e-9
        pos, R = dyn.body_poses(root, q)
        anti = np.stack([R[b].T @ (m.mass[b] * ms[b] * np.array([0, 0, 9.81])) for b in range(13)])
        a3, _ = dyn.forward_bw(root, q, qd, tau, anti, None, mass_scale=ms, com_off=co)
        a4, _ = d0.forward(root, q, qd, tau, mass_scale=ms, com_off=co)
        assert np.abs(a3 - a4).max() < 1e-8 * max(1.0, np.abs(a4).max())
        bs = dyn.body_states(root, q, qd)
        assert np.abs(bs[:, :3] - pos).max() < 1e-12 and np.abs(bs[0, 7:] - root[7:]).max() < 1e-12
        assert np.abs(np.abs(bs[0, 3:7]) - np.abs(root[3:7])).max() < 1e-12 and (bs[:, 6] >= 0).all()
    # a step with applied forces integrates exactly the accelerations of forward_bw
    r, qq, v = root.copy(), q.copy(), qd.copy()
    dyn.step_bw(r, qq, v, tau, bf, bt, mass_scale=ms, com_off=co)
    lim = m.dof_velocity
    assert np.allclose(r[7:], root[7:] + dyn.phys.dt * a1[:6], atol=1e-12) and np.allclose(v, np.clip(qd + dyn.phys.dt * a1[6:], -lim, lim), atol=1e-12)


def test_body_contacts_enter_newtons_law(dyn, flat_model):
    """Non-foot body contacts (trunk box corners, hip-yaw / shank cylinder spheres): for a robot at rest in a random low pose the sum of all
    reported contact forces plus gravity equals the rate of change of the total linear momentum (sum of m_i * a_com_i), and the spheres are
    only evaluated below the gate height."""
    m, rng, seen = flat_model, np.random.default_rng(4), 0
    g = np.array([0.0, 0.0, -9.81])
    for _ in range(60):
        root = np.zeros(13); root[2] = rng.uniform(0.1, 0.4)
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, 3.0)
        root[3:6], root[6] = ax * np.sin(ang / 2), np.cos(ang / 2)
        q = rng.uniform(m.dof_lower, m.dof_upper)
        tau = rng.uniform(-m.dof_effort, m.dof_effort) * 0.2
        qacc, cf, ab = dyn.forward(root, q, np.zeros(12), tau, want_body_acc=True)
        pos, R = dyn.body_poses(root, q)
        dp = sum(m.mass[b] * (R[b] @ (ab[b, 3:] + np.cross(ab[b, :3], m.com[b]))) for b in range(13))  # velocities are zero: a_com = a + alpha x c
        assert np.abs(dp - (cf.sum(axis=0) + m.mass.sum() * g)).max() < 1e-7 * max(1.0, np.abs(cf).max())
        seen += int(np.abs(cf[[0, 3, 4, 9, 10]]).max() > 1.0)
        assert np.abs(cf[[1, 2, 5, 7, 8, 11]]).max() == 0.0  # bodies without collision shapes
    assert seen > 30
    root = np.zeros(13); root[2], root[6] = dyn.phys.body_gate_height + 0.01, 1.0
    q = np.zeros(12); q[[0, 6]] = -1.5; q[[3, 9]] = 2.3  # legs folded forward: a shank sphere would dip under the ground plane
    _, cf = dyn.forward(root, q, np.zeros(12), np.zeros(12))
    assert np.abs(cf[[0, 3, 4, 9, 10]]).max() == 0.0  # above the gate height the spheres are not evaluated


def test_free_fall(dyn):
    root = np.zeros(13); root[2], root[6] = 5.0, 1.0
    qacc, cf = dyn.forward(root, np.zeros(12), np.zeros(12), np.zeros(12))
    assert np.allclose(qacc[:3], [0, 0, -9.81]) and np.abs(qacc[3:]).max() < 1e-12 and np.abs(cf).max() == 0


def test_momentum_conserved_in_flight(flat_model):
    """Airborne, zero gravity, internal torques only: the centre-of-mass velocity is constant up to the O(dt) error of the
    first-order integrator -- the drift over the same 0.2 s must shrink at least 2x when dt shrinks 4x."""
    from oracle.dyn_ref import DynRef

    m = flat_model

    def drift(dt):
        d0 = DynRef(m, phys={"g": (0.0, 0.0, 0.0), "dt": dt, "clamp_qd": 0})
        rng = np.random.default_rng(5)
        root = np.zeros(13); root[2], root[6] = 5.0, 1.0
        q = np.array([-0.2, 0, 0, 0.4, -0.25, 0] * 2, dtype=np.float64); qd = rng.normal(size=12)
        tau = rng.uniform(-0.3, 0.3, 12)  # small: large constant torques spin the light feet to speeds where O(dt) errors dominate

        def com(root, q):
            pos, R = d0.body_poses(root, q)
            return sum(m.mass[b] * (pos[b] + R[b] @ m.com[b]) for b in range(13)) / m.mass.sum()

        c_prev, v = com(root, q), []
        for s in range(int(round(0.2 / dt))):
            d0.step(root, q, qd, tau)
            c = com(root, q)
            v.append((c - c_prev) / dt); c_prev = c
        v = np.array(v)
        return np.abs(v[-1] - v[0]).max(), np.abs(v[0]).max()

    d_coarse, v0 = drift(0.002)
    d_fine, _ = drift(0.0005)
    assert d_coarse < 0.02 * max(v0, 0.05) and d_fine < 0.5 * d_coarse, (d_coarse, d_fine, v0)


def _system_totals(d, m, root, q, qd, g):
    """Total mechanical energy, angular momentum about the world origin and linear momentum from the body states (origin velocity + angular velocity,
    world frame) and the inertials of the model (com and inertia about it, body frame)."""
    def rot(qx):  # xyzw
        x, y, z, w = qx
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    bs = d.body_states(root, q, qd)
    E, L, P = 0.0, np.zeros(3), np.zeros(3)
    for b in range(13):
        R, w = rot(bs[b, 3:7]), bs[b, 10:13]
        xx, yy, zz, xy, xz, yz = m.inertia[b]
        Iw = R @ np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]]) @ R.T
        c = bs[b, :3] + R @ m.com[b]
        vc = bs[b, 7:10] + np.cross(w, R @ m.com[b])
        E += 0.5 * m.mass[b] * vc @ vc + 0.5 * w @ Iw @ w + m.mass[b] * g * c[2]
        L += m.mass[b] * np.cross(c, vc) + Iw @ w
        P += m.mass[b] * vc
    return E, L, P


@pytest.mark.parametrize("g", [0.0, 9.81])
def test_energy_and_angular_momentum_in_flight(flat_model, g):
    """SURVEY section 8c, known-answer test (4): the unforced airborne system under the semi-implicit Euler step (play_mujoco.py:751-756 structure,
    dt = 0.002).  Total energy drifts by O(dt) only -- bounded at dt = 0.002 and 4x smaller at dt / 4 -- and without gravity so do the angular momentum
    about the origin and the linear momentum.  Computed from the BODY STATES (13 poses and twists), i.e. through the kinematics as well."""
    from oracle.dyn_ref import DynRef

    m = flat_model

    def drift(dt):
        d = DynRef(m, phys={"g": (0.0, 0.0, -g), "dt": dt, "clamp_qd": 0}, body_contacts=False)
        rng = np.random.default_rng(11)
        root = np.zeros(13); root[2], root[6] = 5.0, 1.0; root[7:13] = rng.normal(size=6) * 0.5
        q = np.array([-0.2, 0, 0, 0.4, -0.25, 0] * 2, dtype=np.float64); qd = rng.normal(size=12) * 0.5
        E0, L0, P0 = _system_totals(d, m, root, q, qd, g)
        kin = E0 - m.mass.sum() * g * 5.0  # scale of the exchange: the kinetic energy at the start (+ what the fall adds)
        for _ in range(int(round(0.2 / dt))):
            d.step(root, q, qd, np.zeros(12))
        E1, L1, P1 = _system_totals(d, m, root, q, qd, g)
        return abs(E1 - E0), np.abs(L1 - L0).max(), np.abs(P1 - P0).max(), abs(kin) + m.mass.sum() * g * 0.2, np.abs(L0).max(), np.abs(P0).max()

    dE, dL, dP, Es, Ls, Ps = drift(0.002)
    dE4, dL4, dP4, *_ = drift(0.0005)
    assert dE < 0.02 * Es and dE4 < 0.3 * dE, (dE, dE4, Es)
    if g == 0.0:
        assert dL < 1e-4 * Ls and dL4 < 0.3 * dL, (dL, dL4, Ls)
        assert dP < 1e-4 * Ps and dP4 < 0.3 * dP, (dP, dP4, Ps)


@pytest.mark.parametrize("standing", [False, True], ids=["airborne", "standing_on_the_plane"])
def test_mirrored_pose_gives_mirrored_accelerations(flat_model, standing):
    """SURVEY section 8c, known-answer test (5): a left-right mirrored state gives mirrored accelerations.  The shipped inertials are NOT mirror images
    of each other (conftest.symmetrised), so the identity is asserted on a SYMMETRISED copy of the model, where it must hold to rounding -- any
    left / right slip in the axis handling, the link chain, the sole contacts or the leg-against-leg contacts would break it at order one -- and
    the shipped model's own asymmetry is reported beside it."""
    from conftest import MIRROR_SIGN, mirrored_states, symmetrised
    from oracle.dyn_ref import DynRef

    S, my = np.array(MIRROR_SIGN), np.array([1.0, -1.0, 1.0])

    def asymmetry(model):
        d = DynRef(model, body_contacts=False)
        root, q, qd, tau = mirrored_states(np.random.default_rng(3), 100, standing)
        worst, touching = 0.0, 0
        for e in range(100):
            qacc, cf = d.forward(root[e], q[e], qd[e], tau[e])
            touching += bool(np.abs(cf).max() > 0)
            aL, aR = qacc[6:12], qacc[12:18]
            scale = np.abs(aL) + np.abs(aR) + 1e-3 * np.abs(qacc[6:]).max()
            worst = max(worst, (np.abs(aR - S * aL) / scale).max(), max(abs(qacc[1]), abs(qacc[3]), abs(qacc[5])) / np.abs(qacc[:6]).max())
            if np.abs(cf).max() > 0:  # the two feet's forces mirror each other
                worst = max(worst, np.abs(cf[12] - cf[6] * my).max() / np.abs(cf).max())
        return worst, touching

    w_sym, touching = asymmetry(symmetrised(flat_model))
    w_real, _ = asymmetry(flat_model)
    print(f"mirror asymmetry of the accelerations: symmetrised model {w_sym:.2e}, shipped model {w_real:.2e}; {touching} of 100 states with a contact "
          "(sole or leg against leg)")
    assert w_sym < 1e-5, w_sym
    assert touching >= (50 if standing else 10)
    assert 1e-4 < w_real < 1.0  # (the shipped model IS asymmetric at the per-cent level; nothing wild)


def test_standing_normal_force(dyn, flat_model):
    """Held in the default pose by stiff PD, the robot settles with total normal force = weight (31.61 kg * 9.81)."""
    m = flat_model
    tgt = np.array([-0.2, 0, 0, 0.4, -0.25, 0] * 2, dtype=np.float64)
    root = np.zeros(13); root[2], root[6] = 0.70, 1.0
    q, qd = tgt.copy(), np.zeros(12)
    kp, kd = np.array([200.0, 200, 200, 200, 50, 50] * 2), np.array([5.0, 5, 5, 5, 1, 1] * 2)
    fz = []
    for s in range(300):
        tau = np.clip(kp * (tgt - q) - kd * qd, -m.dof_effort, m.dof_effort)
        cf = dyn.step(root, q, qd, tau)
        fz.append(cf[:, 2].sum())
    assert abs(np.mean(fz[200:300]) - m.mass.sum() * 9.81) < 0.05 * m.mass.sum() * 9.81


def test_trained_reference_policy_walks_in_the_oracle(dyn, flat_model):
    """Closed loop: the reference's trained actor (deploy/models/T1.pt weights, tests/golden/t1_actor.npz) commanded 0.5 m/s walks
    forward and stays upright for 6 s in the oracle simulator, with the observation layout of play_mujoco.py:734-744."""
    m = flat_model
    W = G("t1_actor.npz")

    def actor(o):
        x = o
        for i in (0, 2, 4):
            x = W[f"{i}.weight"] @ x + W[f"{i}.bias"]
            x = np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1)
        return W["6.weight"] @ x + W["6.bias"]

    root = np.zeros(13); root[2], root[6] = 0.72, 1.0
    default = np.array([-0.2, 0, 0, 0.4, -0.25, 0] * 2, dtype=np.float64)
    q, qd = default.copy(), np.zeros(12)
    kp, kd = np.array([200.0, 200, 200, 200, 50, 50] * 2), np.array([5.0, 5, 5, 5, 1, 1] * 2)
    cmd, gf, gp, actions, tgt = np.array([0.5, 0.0, 0.0]), 1.5, 0.0, np.zeros(12), default.copy()
    for s in range(3000):
        if s % 10 == 0:
            o = np.zeros(47)
            o[0:3] = tr.quat_rotate_inverse(root[3:7], np.array([0, 0, -1.0])); o[3:6] = tr.quat_rotate_inverse(root[3:7], root[10:13]); o[6:9] = cmd
            o[9], o[10] = np.cos(2 * np.pi * gp), np.sin(2 * np.pi * gp)
            o[11:23], o[23:35], o[35:47] = q - default, qd * 0.1, actions
            actions = np.clip(actor(o), -1, 1); tgt = default + actions
        dyn.step(root, q, qd, np.clip(kp * (tgt - q) - kd * qd, -m.dof_effort, m.dof_effort))
        gp = np.fmod(gp + 0.002 * gf, 1.0)
    up = -tr.quat_rotate_inverse(root[3:7], np.array([0, 0, -1.0]))[2]
    assert root[0] > 2.0 and root[2] > 0.55 and up > 0.95, (root[:3], up)


def test_cross_sim_player_gait_rule_and_walk():
    """tools/play_oracle.py (headless play_mujoco.py:692-764 equivalent): gait-frequency rule and a 5 s commanded walk with the reference's trained actor."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("play_oracle", os.path.join(HERE, "..", "tools", "play_oracle.py"))
    po = importlib.util.module_from_spec(spec); spec.loader.exec_module(po)
    cm = {"gait_frequency": [1.0, 2.0]}
    assert po.gait_frequency(np.zeros(3), cm) == 0.0 and po.gait_frequency(np.array([0.05, 0, 0]), cm) == 0.0
    assert po.gait_frequency(np.array([0.5, 0, 0]), cm) == pytest.approx(1.5) and po.gait_frequency(np.array([2.0, 0, 0]), cm) == pytest.approx(2.0)
    tr_ = po.rollout(po.load_actor(os.path.join(HERE, "golden", "t1_actor.npz")), [0.5, 0.0, 0.0], 5.0)
    assert tr_[-1, 0] > 1.5 and tr_[:, 2].min() > 0.55
    stand = po.rollout(po.load_actor(os.path.join(HERE, "golden", "t1_actor.npz")), [0.0, 0.0, 0.0], 3.0)
    assert abs(stand[-1, 0]) < 0.5 and stand[:, 2].min() > 0.55  # zero command: gait frequency 0, the policy stands


# ------------------------------------------------------------------ an independent derivation of the leg dynamics: Lagrange's equations (sympy)
@pytest.fixture(scope="module")
def lagrange_leg(flat_model):
    """tau(q, qd, qdd, g) of ONE fixed-base T1 leg (6 DoF; inertials of resources/T1/T1_locomotion.xml:55-79 / :88-112 via the flat model) from
    Lagrange's equations d/dt dL/dqd - dL/dq, L = sum_i 1/2 m v_c^2 + 1/2 w^T I_c w - m g z_c, derived symbolically (kinetic energy from
    the Jacobian of the centre-of-mass positions and from R^T Rdot): no spatial algebra, no recursion -- mathematics that is not the build's
    own.  Returns {leg: f(q, qd, qdd, g) -> tau[6]}."""
    import sympy as sp

    m = flat_model
    out = {}
    for leg in range(2):
        q, qd, qdd, g = sp.symbols("q0:6"), sp.symbols("qd0:6"), sp.symbols("qdd0:6"), sp.Symbol("g")

        def rot(ax, a):
            c, s = sp.cos(a), sp.sin(a)
            return {1: sp.Matrix([[1, 0, 0], [0, c, -s], [0, s, c]]), 2: sp.Matrix([[c, 0, s], [0, 1, 0], [-s, 0, c]]), 3: sp.Matrix([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[ax]

        R, p, T, V = sp.eye(3), sp.zeros(3, 1), 0, 0
        for i in range(6):
            b = 1 + 6 * leg + i
            p = p + R * sp.Matrix([float(v) for v in m.body_pos[b]])
            R = R * rot(int(m.joint_axis[b]), q[i])
            pc = p + R * sp.Matrix([float(v) for v in m.com[b]])
            vc = pc.jacobian(q) * sp.Matrix(qd)
            Rdot = sum((R.diff(q[k]) * qd[k] for k in range(6)), sp.zeros(3, 3))
            W = R.T * Rdot  # skew matrix of the angular velocity in body coordinates
            w = sp.Matrix([W[2, 1], W[0, 2], W[1, 0]])
            I6 = [float(v) for v in m.inertia[b]]
            Ic = sp.Matrix([[I6[0], I6[3], I6[4]], [I6[3], I6[1], I6[5]], [I6[4], I6[5], I6[2]]])
            T += sp.Rational(1, 2) * float(m.mass[b]) * (vc.T * vc)[0] + sp.Rational(1, 2) * (w.T * Ic * w)[0]
            V += float(m.mass[b]) * g * pc[2]
        L = T - V
        tau = []
        for i in range(6):
            dLdqd = sp.diff(L, qd[i])
            tau.append(sum(sp.diff(dLdqd, q[k]) * qd[k] + sp.diff(dLdqd, qd[k]) * qdd[k] for k in range(6)) - sp.diff(L, q[i]))
        out[leg] = sp.lambdify((q, qd, qdd, g), tau, modules="numpy", cse=True)
    return out


def test_oracle_rnea_and_aba_match_lagrange_equations(lagrange_leg, dyn_smooth, flat_model):
    """oracle/dyn_ref.c against Lagrange's equations of one leg (reference: the MuJoCo step the north star names, play_mujoco.py:751-756, solves
    the same equations of motion M(q) qdd + c(q, qd) + g(q) = tau).  (1) RNEA with the trunk held still in gravity: joint torques of each leg on
    1,000 random states, relative 1e-10.  (2) ABA: the floating trunk made 1e8 times heavier falls freely and the legs, in its frame, obey the
    gravity-free fixed-base equations: qdd = M^-1 (tau - c), relative 1e-6 (the trunk's reaction is O(1e-8)).  This cannot lift the
    'parity unpinned' label (only reference-held vectors could); it is the one check here whose mathematics is not the build's own recursion."""
    m, rng = flat_model, np.random.default_rng(31)
    root = np.zeros(13); root[2], root[6] = 5.0, 1.0
    worst_id, worst_fd = 0.0, 0.0
    ms = np.ones(13); ms[0] = 1.0e8
    for _ in range(1000):
        q, qd, qdd = rng.uniform(m.dof_lower, m.dof_upper), rng.normal(size=12) * 2.0, rng.normal(size=12) * 20.0
        res = dyn_smooth.inverse(root, q, qd, np.concatenate([np.zeros(6), qdd]))
        for leg in range(2):
            sl = slice(6 * leg, 6 * leg + 6)
            want = np.array(lagrange_leg[leg](q[sl], qd[sl], qdd[sl], 9.81), dtype=np.float64)
            worst_id = max(worst_id, np.abs(res[6:][sl] - want).max() / max(1.0, np.abs(want).max()))
    for _ in range(300):
        q, qd, tau = rng.uniform(m.dof_lower, m.dof_upper), rng.normal(size=12) * 2.0, rng.uniform(-m.dof_effort, m.dof_effort)
        qacc, _ = dyn_smooth.forward(root, q, qd, tau, mass_scale=ms)
        for leg in range(2):
            sl = slice(6 * leg, 6 * leg + 6)
            f = lagrange_leg[leg]
            c = np.array(f(q[sl], qd[sl], np.zeros(6), 0.0), dtype=np.float64)
            M = np.stack([np.array(f(q[sl], qd[sl], np.eye(6)[k], 0.0), dtype=np.float64) - c for k in range(6)], axis=1)
            assert np.allclose(M, M.T, atol=1e-12) and np.linalg.eigvalsh(M).min() > 0
            want = np.linalg.solve(M, tau[sl] - c)
            worst_fd = max(worst_fd, np.abs(qacc[6:][sl] - want).max() / max(1.0, np.abs(want).max()))
    assert worst_id < 1e-10 and worst_fd < 1e-6, (worst_id, worst_fd)
