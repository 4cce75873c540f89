"""oracle/task_ref.py -- TEST INFRASTRUCTURE, not product code.

numpy restatement of the T1 task logic that surrounds the physics in the reference's `envs/t1.py`, one function per
reference method, written array-at-a-time like the reference (no shared code with the HIP kernels):

    feet_state            t1.py:529-549   _refresh_feet_state
    check_termination     t1.py:551-558   _check_termination
    reward_terms          t1.py:606-730   the 26 `_reward_*` functions (unscaled)
    compute_observations  t1.py:574-603   _compute_observations
    pd_torque             t1.py:446-448
    terrain_heights       utils/terrain.py:101-121
    T1Ref.step / .reset   t1.py:437-497 / 294-341, chaining the above around oracle/dyn_ref.c physics

Pinned by tests/golden/task_logic.npz and terrain_heights.npz, which hold the outputs of the reference's own methods
(tests/golden/make_task_fixtures.py).  The per-step noise uses this build's counter-based Philox streams (the reference's torch
global RNG cannot be reproduced), restated here independently from the published Philox4x32-10 algorithm.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import math

import numpy as np

PI = np.pi

REWARD_NAMES = [
    "survival", "tracking_lin_vel_x", "tracking_lin_vel_y", "tracking_ang_vel", "base_height", "orientation", "torques",
    "torque_tiredness", "power", "lin_vel_z", "ang_vel_xy", "dof_vel", "dof_acc", "root_acc", "action_rate", "dof_pos_limits",
    "dof_vel_limits", "torque_limits", "collision", "feet_slip", "feet_vel_z", "feet_yaw_diff", "feet_yaw_mean", "feet_roll",
    "feet_distance", "feet_swing",
]

# RNG stream ids of this build (booster_gym_amd/csrc/bg_rng.h)
RS_OBS0, RS_OBS1, RS_OBS2, RS_DOFPOS, RS_DOFVEL = 0, 1, 2, 4, 8
RS_KICK0, RS_KICK1, RS_PUSH0, RS_PUSH1, RS_RESET0, RS_RESET1, RS_RESETDOF, RS_CMD0, RS_CMD1, RS_CURR, RS_ACTOR = 12, 13, 14, 15, 16, 17, 20, 24, 25, 26, 32


# ------------------------------------------------------------------ Philox4x32-10 (Salmon et al. 2011), vectorised
def philox4x32_10(k0, k1, c0, c1, c2, c3):
    m0, m1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = m0 * c0.astype(np.uint64)
            p1 = m1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + 0x9E3779B9) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + 0xBB67AE85) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def rand4(seed, env, step, stream):
    """4 uniforms in (0,1) and 4 standard normals per env (Box-Muller on the same bits), float32 arithmetic."""
    env = np.asarray(env, dtype=np.uint32)
    o = philox4x32_10(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, env, np.uint32(step & 0xFFFFFFFF), np.uint32(stream), np.uint32(0))
    u = np.stack([((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0) for x in o], axis=-1)
    n = np.empty_like(u)
    for i in range(2):
        rad = np.sqrt(np.float32(-2.0) * np.log(u[..., 2 * i]))
        ang = np.float32(6.283185307179586) * u[..., 2 * i + 1]
        n[..., 2 * i], n[..., 2 * i + 1] = rad * np.cos(ang), rad * np.sin(ang)
    return u, n


def apply_rand(x, spec, u, n):
    """utils/utils.py:5-30 with the draw supplied; spec = None or dict(distribution, operation, range)."""
    if spec is None:
        return x
    a, b = spec["range"]
    val = a + b * n if spec["distribution"] == "gaussian" else a + (b - a) * u
    return x + val if spec["operation"] == "additive" else x * val


# ------------------------------------------------------------------ quaternion helpers (xyzw), SURVEY appendix E
def quat_rotate(q, v):
    w, qv = q[..., 3:4], q[..., :3]
    return v * (2 * w * w - 1) + np.cross(qv, v) * w * 2 + qv * np.sum(qv * v, axis=-1, keepdims=True) * 2


def quat_rotate_inverse(q, v):  # play_mujoco.py:282-297
    w, qv = q[..., 3:4], q[..., :3]
    return v * (2 * w * w - 1) - np.cross(qv, v) * w * 2 + qv * np.sum(qv * v, axis=-1, keepdims=True) * 2


def get_euler_xyz(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    roll = np.arctan2(2 * (w * x + y * z), w * w - x * x - y * y + z * z)
    sinp = 2 * (w * y - z * x)
    pitch = np.where(np.abs(sinp) >= 1, np.sign(sinp) * (PI / 2), np.arcsin(np.clip(sinp, -1, 1)))
    yaw = np.arctan2(2 * (w * z + x * y), w * w + x * x - y * y - z * z)
    return roll % (2 * PI), pitch % (2 * PI), yaw % (2 * PI)


def wrap_pi(x):
    return (x + PI) % (2 * PI) - PI


def mat_to_quat(R):
    """rotation matrix -> xyzw quaternion (w >= 0 branch is enough for tests: feet never flip past 180 deg)"""
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        return np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
    q = np.zeros(4)
    q[i] = 0.25 * s
    q[j] = (R[j, i] + R[i, j]) / s
    q[k] = (R[k, i] + R[i, k]) / s
    q[3] = (R[k, j] - R[j, k]) / s
    return q


# ------------------------------------------------------------------ terrain (utils/terrain.py:101-121)
def terrain_heights(terrain, xy):
    """terrain: None (plane) or dict(height_field_raw, hscale, vscale, border_px).  Indices clamped to the field."""
    xy = np.asarray(xy, dtype=np.float64)
    if terrain is None:
        return np.zeros(len(xy))
    hf = terrain["height_field_raw"]
    x = terrain["border_px"] + xy[:, 0] / terrain["hscale"]
    y = terrain["border_px"] + xy[:, 1] / terrain["hscale"]
    x1 = np.clip(np.floor(x).astype(int), 0, hf.shape[0] - 2)
    y1 = np.clip(np.floor(y).astype(int), 0, hf.shape[1] - 2)
    x2, y2 = x1 + 1, y1 + 1
    return ((x2 - x) * (y2 - y) * hf[x1, y1] + (x - x1) * (y2 - y) * hf[x2, y1] + (x2 - x) * (y - y1) * hf[x1, y2]
            + (x - x1) * (y - y1) * hf[x2, y2]) * terrain["vscale"]


# ------------------------------------------------------------------ t1.py:529-549
def feet_state(feet_pos, feet_quat, feet_edge_pos, terrain):
    n = feet_pos.shape[0]
    roll, _, yaw = get_euler_xyz(feet_quat.reshape(-1, 4))
    roll, yaw = wrap_pi(roll.reshape(n, 2)), wrap_pi(yaw.reshape(n, 2))
    edges = np.asarray(feet_edge_pos, dtype=np.float64)  # [4,3]
    pos = np.repeat(feet_pos[:, :, None, :], 4, axis=2).reshape(-1, 3)
    quat = np.repeat(feet_quat[:, :, None, :], 4, axis=2).reshape(-1, 4)
    rel = np.broadcast_to(edges[None, None], (n, 2, 4, 3)).reshape(-1, 3)
    edge_w = pos + quat_rotate(quat, rel)
    contact = np.any((edge_w[:, 2] - terrain_heights(terrain, edge_w) < 0.01).reshape(n, 2, 4), axis=2)
    return roll, yaw, contact


# ------------------------------------------------------------------ t1.py:551-558
def check_termination(root_states, episode_length, cmd_resample_time, terrain, rew_cfg, dt, contact_forces=None, term_idx=()):
    n = root_states.shape[0]
    reset = np.zeros(n, dtype=bool)
    if len(term_idx):
        reset |= np.any(np.linalg.norm(contact_forces[:, list(term_idx), :], axis=-1) > 1.0, axis=1)
    reset |= np.sum(np.square(root_states[:, 7:13]), axis=-1) > rew_cfg["terminate_vel"]
    reset |= root_states[:, 2] - terrain_heights(terrain, root_states[:, :2]) < rew_cfg["terminate_height"]
    time_out = episode_length > np.ceil(rew_cfg["episode_length_s"] / dt)
    reset |= time_out
    time_out = time_out | (episode_length == cmd_resample_time)
    return reset, time_out


# ------------------------------------------------------------------ t1.py:606-730
def reward_terms(s, rew_cfg, dt, limits, terrain):
    """s: dict of arrays named like the reference attributes.  limits: dict(dof_pos_limits [12,2], dof_vel_limits, torque_limits).
    Returns {name: unscaled term}."""
    r = {}
    n = s["root_states"].shape[0]
    sig = rew_cfg["tracking_sigma"]
    r["survival"] = np.ones(n)
    r["tracking_lin_vel_x"] = np.exp(-np.square(s["commands"][:, 0] - s["filtered_lin_vel"][:, 0]) / sig)
    r["tracking_lin_vel_y"] = np.exp(-np.square(s["commands"][:, 1] - s["filtered_lin_vel"][:, 1]) / sig)
    r["tracking_ang_vel"] = np.exp(-np.square(s["commands"][:, 2] - s["filtered_ang_vel"][:, 2]) / sig)
    base_height = s["root_states"][:, 2] - terrain_heights(terrain, s["root_states"][:, :2])
    r["base_height"] = np.square(base_height - rew_cfg["base_height_target"])
    pen = s.get("penalized_contact_indices", [])
    r["collision"] = (np.sum(np.linalg.norm(s["contact_forces"][:, list(pen), :], axis=-1) > 1.0, axis=-1).astype(np.float64)
                      if len(pen) else np.zeros(n))
    r["lin_vel_z"] = np.square(s["filtered_lin_vel"][:, 2])
    r["ang_vel_xy"] = np.sum(np.square(s["base_ang_vel"][:, :2]), axis=-1)
    r["orientation"] = np.sum(np.square(s["projected_gravity"][:, :2]), axis=-1)
    r["torques"] = np.sum(np.square(s["torques"]), axis=-1)
    r["dof_vel"] = np.sum(np.square(s["dof_vel"]), axis=-1)
    r["dof_acc"] = np.sum(np.square((s["last_dof_vel"] - s["dof_vel"]) / dt), axis=-1)
    r["root_acc"] = np.sum(np.square((s["last_root_vel"] - s["root_states"][:, 7:13]) / dt), axis=-1)
    r["action_rate"] = np.sum(np.square(s["last_actions"] - s["actions"]), axis=-1)
    lo, hi = limits["dof_pos_limits"][:, 0], limits["dof_pos_limits"][:, 1]
    lower = lo + 0.5 * (1 - rew_cfg["soft_dof_pos_limit"]) * (hi - lo)
    upper = hi - 0.5 * (1 - rew_cfg["soft_dof_pos_limit"]) * (hi - lo)
    r["dof_pos_limits"] = np.sum(((s["dof_pos"] < lower) | (s["dof_pos"] > upper)).astype(np.float64), axis=-1)
    r["dof_vel_limits"] = np.sum(np.clip(np.abs(s["dof_vel"]) - limits["dof_vel_limits"] * rew_cfg["soft_dof_vel_limit"], 0.0, 1.0), axis=-1)
    r["torque_limits"] = np.sum(np.clip(np.abs(s["torques"]) - limits["torque_limits"] * rew_cfg["soft_torque_limit"], 0.0, None), axis=-1)
    r["torque_tiredness"] = np.sum(np.clip(np.square(s["torques"] / limits["torque_limits"]), None, 1.0), axis=-1)
    r["power"] = np.sum(np.clip(s["torques"] * s["dof_vel"], 0.0, None), axis=-1)
    dfeet = (s["last_feet_pos"] - s["feet_pos"]) / dt
    r["feet_slip"] = np.sum(np.sum(np.square(dfeet), axis=-1) * s["feet_contact"].astype(np.float64), axis=-1) * (s["episode_length_buf"] > 1)
    r["feet_vel_z"] = np.sum(np.square(dfeet)[:, :, 2], axis=-1)
    r["feet_roll"] = np.sum(np.square(s["feet_roll"]), axis=-1)
    r["feet_yaw_diff"] = np.square(wrap_pi(s["feet_yaw"][:, 1] - s["feet_yaw"][:, 0]))
    base_yaw = get_euler_xyz(s["root_states"][:, 3:7])[2]
    mean = s["feet_yaw"].mean(axis=-1) + PI * (np.abs(s["feet_yaw"][:, 1] - s["feet_yaw"][:, 0]) > PI)
    r["feet_yaw_mean"] = np.square(wrap_pi(base_yaw - mean))
    fd = np.abs(np.cos(base_yaw) * (s["feet_pos"][:, 1, 1] - s["feet_pos"][:, 0, 1]) - np.sin(base_yaw) * (s["feet_pos"][:, 1, 0] - s["feet_pos"][:, 0, 0]))
    r["feet_distance"] = np.clip(rew_cfg["feet_distance_ref"] - fd, 0.0, 0.1)
    on = s["gait_frequency"] > 1.0e-8
    left = (np.abs(s["gait_process"] - 0.25) < 0.5 * rew_cfg["swing_period"]) & on
    right = (np.abs(s["gait_process"] - 0.75) < 0.5 * rew_cfg["swing_period"]) & on
    r["feet_swing"] = (left & ~s["feet_contact"][:, 0]).astype(np.float64) + (right & ~s["feet_contact"][:, 1]).astype(np.float64)
    return r


def total_reward(terms, scales, only_positive):
    """t1.py:560-572: scales = {name: yaml_scale * dt} with zero entries dropped."""
    tot, scaled = 0.0, {}
    for name, sc in scales.items():
        scaled[name] = terms[name] * sc
        tot = tot + scaled[name]
    if only_positive:
        tot = np.clip(tot, 0.0, None)
    return tot, scaled


# ------------------------------------------------------------------ t1.py:574-603 (noise draws supplied by the caller; None = no noise)
def compute_observations(s, norm, default_dof_pos, terrain, noisy=None):
    z = lambda key, x: x if noisy is None else noisy[key](x)
    on = (s["gait_frequency"] > 1.0e-8).astype(np.float64)
    obs = np.concatenate([
        z("gravity", s["projected_gravity"]) * norm["gravity"],
        z("ang_vel", s["base_ang_vel"]) * norm["ang_vel"],
        s["commands"][:, :3] * np.array([norm["lin_vel"], norm["lin_vel"], norm["ang_vel"]]),
        (np.cos(2 * PI * s["gait_process"]) * on)[:, None],
        (np.sin(2 * PI * s["gait_process"]) * on)[:, None],
        z("dof_pos", s["dof_pos"] - default_dof_pos) * norm["dof_pos"],
        z("dof_vel", s["dof_vel"]) * norm["dof_vel"],
        s["actions"],
    ], axis=-1)
    height = s["root_states"][:, 2] - terrain_heights(terrain, s["root_states"][:, :2])
    priv = np.concatenate([
        s["base_mass_scaled"],
        z("lin_vel", s["base_lin_vel"]) * norm["lin_vel"],
        z("height", height)[:, None],
        s["push_force"] * norm["push_force"],
        s["push_torque"] * norm["push_torque"],
    ], axis=-1)
    return obs, priv


def pd_torque(kp, kd, friction, torque_limits, targets, dof_pos, dof_vel):
    """t1.py:446-448"""
    t = kp * (targets - dof_pos) - kd * dof_vel
    fr = np.minimum(friction, np.abs(t)) * np.sign(t)
    return np.clip(t - fr, -torque_limits, torque_limits)


# ------------------------------------------------------------------ command curriculum, t1.py:391-435
def update_curriculum(prob, levels, ep_len, filt_lin, filt_ang, commands, env_ids, cm, rew_cfg, dt):
    """t1.py:391-413: returns the new probability grid (clamped at 1)."""
    prob = prob.copy()
    lv, av = cm["lin_vel_levels"], cm["ang_vel_levels"]
    success = ep_len[env_ids] > np.ceil(rew_cfg["episode_length_s"] / dt) * (1 - cm["episode_length_toler"])
    success &= np.abs(filt_lin[env_ids, 0] - commands[env_ids, 0]) < cm["lin_vel_x_toler"]
    success &= np.abs(filt_lin[env_ids, 1] - commands[env_ids, 1]) < cm["lin_vel_y_toler"]
    success &= np.abs(filt_ang[env_ids, 2] - commands[env_ids, 2]) < cm["ang_vel_yaw_toler"]
    for i, e in enumerate(env_ids):
        if success[i]:
            x, y = int(levels[e, 0]) + lv, int(levels[e, 1]) + av
            prob[x, y] += cm["update_rate"]
            if x > 0:
                prob[x - 1, y] += cm["update_rate"]
            if x < prob.shape[0] - 1:
                prob[x + 1, y] += cm["update_rate"]
            if y > 0:
                prob[x, y - 1] += cm["update_rate"]
            if y < prob.shape[1] - 1:
                prob[x, y + 1] += cm["update_rate"]
    return np.minimum(prob, 1.0)


def curriculum_commands(grid_idx, ux, uy, uyaw, cm, ncols):
    """t1.py:415-435 given the multinomial draw `grid_idx` and the three uniform draws (U(-.5,.5), U(-1,1), U(-.5,.5)).
    The reference decodes lin = idx % ncols - L, ang = idx // ncols - A (transposed w.r.t. the update's prob[lin][ang]); kept."""
    lin = grid_idx % ncols - cm["lin_vel_levels"]
    ang = grid_idx // ncols - cm["ang_vel_levels"]
    cmd = np.stack([(lin + ux) * cm["lin_vel_x_resolution"], np.abs(lin) * uy * cm["lin_vel_y_resolution"], (ang + uyaw) * cm["ang_vel_resolution"]], axis=1)
    return lin, ang, cmd


def _mix32(x):
    x = np.uint32(x)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16); x = np.uint32(x * np.uint32(0x85EBCA6B)); x ^= x >> np.uint32(13); x = np.uint32(x * np.uint32(0xC2B2AE35)); x ^= x >> np.uint32(16)
    return np.uint32(x)


def keyed_perm(p, K, key):
    """This build's stand-in for torch.randperm(K) (t1.py:381; parallel.exact_still_count): a keyed pseudo-random permutation of [0, K) -- 4-round
    Feistel network on the smallest even-width bit field covering K, cycle-walked back into range (csrc/bg_env.h:keyed_perm)."""
    if K <= 1:
        return 0
    bits = 2
    while bits < 32 and (1 << bits) < K:
        bits += 2
    hb = bits >> 1
    hm = (1 << hb) - 1
    x = int(p)
    for _ in range(64):
        L, R = x >> hb, x & hm
        for r in range(4):
            f = int(_mix32((R ^ ((int(key) + r * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF)) & hm
            L, R = R, L ^ f
        x = (L << hb) | R
        if x < K:
            return x
    return int(p)


# ------------------------------------------------------------------ full env: t1.py:294-341, 437-497 around the C physics oracle
class T1Ref:
    """State arrays are float64 numpy, env-major.  `dyn` is an oracle.dyn_ref.DynRef (its terrain must match `terrain`)."""

    def __init__(self, cfg, model, dyn, params, terrain=None, seed=0, rank=0):
        self.cfg, self.m, self.dyn, self.terrain = cfg, model, dyn, terrain
        self.p = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}  # kp kd fric mass_scale com_off foot_mat bms origins
        self.n = self.p["kp"].shape[0]
        self.seed = (int(seed) & 0xFFFFFFFF) | ((int(rank) + 1) << 32)
        self.dt = cfg["control"]["decimation"] * cfg["sim"]["dt"]
        n = self.n
        dja = cfg["init_state"]["default_joint_angles"]
        self.default = np.zeros(12)
        for i, name in enumerate(model.dof_names):
            self.default[i] = dja["default"]
            for k in dja:
                if k in name:
                    self.default[i] = dja[k]
        ist = cfg["init_state"]
        self.base_init = np.array(ist["pos"] + ist["rot"] + ist["lin_vel"] + ist["ang_vel"], dtype=np.float64)
        self.root = np.zeros((n, 13)); self.root[:, 6] = 1.0
        self.q, self.qd = np.zeros((n, 12)), np.zeros((n, 12))
        self.last_tgt, self.actions, self.last_actions = np.zeros((n, 12)), np.zeros((n, 12)), np.zeros((n, 12))
        self.last_qd, self.last_rootvel = np.zeros((n, 12)), np.zeros((n, 6))
        self.cmd, self.gait_f, self.gait_p = np.zeros((n, 3)), np.zeros(n), np.zeros(n)
        self.filt_lin, self.filt_ang = np.zeros((n, 3)), np.zeros((n, 3))
        self.last_feet, self.push = np.zeros((n, 2, 3)), np.zeros((n, 6))
        self.ep_len, self.cmd_time, self.delay = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
        self.step_count = 0
        self.limits = {"dof_pos_limits": np.stack([model.dof_lower, model.dof_upper], axis=1), "dof_vel_limits": np.asarray(model.dof_velocity),
                       "torque_limits": np.asarray(model.dof_effort)}
        sc = cfg["rewards"]["scales"]
        self.scales = {k: v * self.dt for k, v in sc.items() if v != 0}
        self.env_ids = np.arange(n, dtype=np.uint32)
        par = cfg.get("parallel", {}) or {}
        self.shared_reset_noise = bool(par.get("shared_reset_noise", True))
        self.exact_still_count = bool(par.get("exact_still_count", False))        # t1.py:381-383 as written (exact count, random subset)
        self.same_step_curriculum = bool(par.get("same_step_curriculum", False))  # t1.py:305 before :365 (grid read after this step's updates)
        cm = cfg["commands"]
        self.curriculum = bool(cm.get("curriculum", False))
        self.curr_prob = np.zeros((2 * cm["lin_vel_levels"] + 1, 2 * cm["ang_vel_levels"] + 1), dtype=np.float64)
        self.curr_prob[cm["lin_vel_levels"], cm["ang_vel_levels"]] = 1.0
        self.curr_prob_read = self.curr_prob.copy()  # samplers see the grid as of the start of the step (bg_env.h)
        self.curr_levels = np.zeros((n, 2), dtype=np.int64)

    def quantize_state_fp16(self):
        """The product's optional fp16 state storage (bg_env_cfg.state_fp16, BASELINE configs[4]): everything the env keeps between steps except
        the root position, last_feet_pos and the per-env parameters is rounded to fp16 (nearest even) when it is stored at the end of a step."""
        h = lambda a: np.asarray(a, dtype=np.float64).astype(np.float16).astype(np.float64)
        self.root[:, 3:] = h(self.root[:, 3:])
        for k in ("q", "qd", "last_tgt", "actions", "last_actions", "last_qd", "last_rootvel", "cmd", "gait_f", "gait_p", "filt_lin", "filt_ang", "push"):
            setattr(self, k, h(getattr(self, k)))

    # -- helpers
    def _feet(self):
        """Foot rows of the rigid-body state tensor (t1.py:529-531): world position and orientation (xyzw) of both foot links."""
        bs = self.dyn.body_states_batch(self.root, self.q, self.qd)
        return bs[:, (6, 12), 0:3].copy(), bs[:, (6, 12), 3:7].copy()

    def _r4(self, stream, step, so, env=None):
        return rand4(self.seed, self.env_ids if env is None else env, step, so + stream)

    def _reset_and_observe(self, step, so, reset, base_lin, base_ang, proj_g, feet_pos, teleport):
        cfg, n = self.cfg, self.n
        rnd, noise = cfg["randomization"], cfg["noise"]
        # ---- _update_curriculum (t1.py:305, 391-413): pre-reset values; not in reset-all mode (episodes have length 0 there)
        if self.curriculum and so == 0 and reset.any():
            self.curr_prob = update_curriculum(self.curr_prob, self.curr_levels, self.ep_len, self.filt_lin, self.filt_ang, self.cmd,
                                               np.nonzero(reset)[0], cfg["commands"], cfg["rewards"], self.dt)
        # ---- _reset_idx (t1.py:301-341)
        if reset.any():
            nenv = np.full(n, 0xFFFFFFFF, dtype=np.uint32) if self.shared_reset_noise else self.env_ids
            for leg in range(2):
                d0u, d0n = self._r4(RS_RESETDOF + leg * 2, step, so, nenv)
                d1u, d1n = self._r4(RS_RESETDOF + leg * 2 + 1, step, so, nenv)
                for i in range(6):
                    u, nn = (d0u[:, i], d0n[:, i]) if i < 4 else (d1u[:, i - 4], d1n[:, i - 4])
                    j = leg * 6 + i
                    newq = apply_rand(np.full(n, self.default[j]), rnd.get("init_dof_pos"), u, nn)
                    self.q[reset, j] = newq[reset]
            self.qd[reset] = 0.0
            self.last_tgt[reset] = self.q[reset]
            r0u, r0n = self._r4(RS_RESET0, step, so)
            r1u, r1n = self._r4(RS_RESET1, step, so)
            px = apply_rand(self.base_init[0] + self.p["origins"][:, 0], rnd.get("init_base_pos_xy"), r0u[:, 0], r0n[:, 0])
            py = apply_rand(self.base_init[1] + self.p["origins"][:, 1], rnd.get("init_base_pos_xy"), r0u[:, 1], r0n[:, 1])
            pz = self.base_init[2] + terrain_heights(self.terrain, np.stack([px, py], axis=1))
            yaw = r0u[:, 2].astype(np.float64) * 2 * PI
            newroot = np.zeros((n, 13))
            newroot[:, 0], newroot[:, 1], newroot[:, 2] = px, py, pz
            newroot[:, 5], newroot[:, 6] = np.sin(0.5 * yaw), np.cos(0.5 * yaw)
            newroot[:, 7] = apply_rand(np.zeros(n), rnd.get("init_base_lin_vel_xy"), r1u[:, 0], r1n[:, 0])
            newroot[:, 8] = apply_rand(np.zeros(n), rnd.get("init_base_lin_vel_xy"), r1u[:, 1], r1n[:, 1])
            newroot[:, 9:13] = self.base_init[9:13]
            self.root[reset] = newroot[reset]
            self.ep_len[reset] = 0; self.cmd_time[reset] = 0
            self.filt_lin[reset] = 0.0; self.filt_ang[reset] = 0.0
            dec = cfg["control"]["decimation"]
            dl = np.minimum((r0u[:, 3] * np.float32(dec)).astype(np.int64), dec - 1)
            self.delay[reset] = dl[reset]
        # ---- _teleport_robot (t1.py:343-360)
        feet_store = feet_pos.copy()
        if teleport and self.terrain is not None:
            t = cfg["terrain"]
            ew, el, b = t["num_terrains"] * t["terrain_width"], t["terrain_length"], t["border_size"]
            sx = np.where(self.root[:, 0] < -0.75 * b, ew + b, 0.0) - np.where(self.root[:, 0] > ew + 0.75 * b, ew + b, 0.0)
            sy = np.where(self.root[:, 1] < -0.75 * b, el + b, 0.0) - np.where(self.root[:, 1] > el + 0.75 * b, el + b, 0.0)
            self.root[:, 0] += sx; self.root[:, 1] += sy
            feet_store[:, :, 0] += sx[:, None]; feet_store[:, :, 1] += sy[:, None]
        # ---- _resample_commands (t1.py:362-389); still envs by per-env Bernoulli (this build's documented deviation)
        rs = self.ep_len == self.cmd_time
        if rs.any():
            cm = cfg["commands"]
            c0u, _ = self._r4(RS_CMD0, step, so)
            c1u, _ = self._r4(RS_CMD1, step, so)
            new = np.stack([cm["lin_vel_x"][0] + (cm["lin_vel_x"][1] - cm["lin_vel_x"][0]) * c0u[:, 0],
                            cm["lin_vel_y"][0] + (cm["lin_vel_y"][1] - cm["lin_vel_y"][0]) * c0u[:, 1],
                            cm["ang_vel_yaw"][0] + (cm["ang_vel_yaw"][1] - cm["ang_vel_yaw"][0]) * c0u[:, 2]], axis=1).astype(np.float64)
            if self.curriculum:
                cru, _ = self._r4(RS_CURR, step, so)
                p = np.minimum(self.curr_prob if self.same_step_curriculum else self.curr_prob_read, 1.0).astype(np.float32).reshape(-1)
                total = np.float32(0.0)
                for v in p:
                    total = np.float32(total + v)
                cum = np.cumsum(p, dtype=np.float32)  # sequential float32 accumulation like the kernel
                target = cru[:, 0] * total
                idx = np.minimum(np.array([int(np.searchsorted(cum, t, side="right")) for t in target]), p.size - 1)
                lin, ang, ccmd = curriculum_commands(idx, cru[:, 1].astype(np.float64) - 0.5, 2.0 * cru[:, 2].astype(np.float64) - 1.0,
                                                     cru[:, 3].astype(np.float64) - 0.5, cm, self.curr_prob.shape[1])
                new = ccmd
                self.curr_levels[rs, 0] = lin[rs]; self.curr_levels[rs, 1] = ang[rs]
            gf = (cm["gait_frequency"][0] + (cm["gait_frequency"][1] - cm["gait_frequency"][0]) * c0u[:, 3]).astype(np.float64)
            if self.exact_still_count:
                ids = np.nonzero(rs)[0]  # positions in env order
                K = len(ids)
                m = int(float(np.float32(cm["still_proportion"])) * K)
                key = _mix32(np.uint32(self.seed & 0xFFFFFFFF) ^ _mix32(np.uint32((step + 0x632BE5AB * (so + 1)) & 0xFFFFFFFF)))
                still = np.zeros(n, dtype=bool)
                for pos, e in enumerate(ids):
                    still[e] = keyed_perm(pos, K, key) < m
            else:
                still = c1u[:, 0] < np.float32(cm["still_proportion"])
            new[still] = 0.0; gf[still] = 0.0
            lo, hi = int(cm["resampling_time_s"][0] / self.dt), int(cm["resampling_time_s"][1] / self.dt)
            add = lo + np.minimum((c1u[:, 1] * np.float32(hi - lo)).astype(np.int64), hi - lo - 1) if hi > lo else np.full(n, lo)
            self.cmd[rs] = new[rs]; self.gait_f[rs] = gf[rs]; self.cmd_time[rs] += add[rs]
        # ---- _compute_observations (t1.py:574-603) with this build's noise streams
        o0u, o0n = self._r4(RS_OBS0, step, so); o1u, o1n = self._r4(RS_OBS1, step, so); o2u, o2n = self._r4(RS_OBS2, step, so)
        dpu, dpn, dvu, dvn = np.zeros((n, 12), np.float32), np.zeros((n, 12), np.float32), np.zeros((n, 12), np.float32), np.zeros((n, 12), np.float32)
        for leg in range(2):
            for k in range(2):
                pu, pn = self._r4(RS_DOFPOS + leg * 2 + k, step, so); vu, vn = self._r4(RS_DOFVEL + leg * 2 + k, step, so)
                cols = range(4) if k == 0 else range(2)
                for c in cols:
                    j = leg * 6 + 4 * k + c
                    dpu[:, j], dpn[:, j], dvu[:, j], dvn[:, j] = pu[:, c], pn[:, c], vu[:, c], vn[:, c]
        gu = np.stack([o0u[:, 0], o0u[:, 1], o0u[:, 2]], 1); gn = np.stack([o0n[:, 0], o0n[:, 1], o0n[:, 2]], 1)
        au = np.stack([o0u[:, 3], o1u[:, 0], o1u[:, 1]], 1); an = np.stack([o0n[:, 3], o1n[:, 0], o1n[:, 1]], 1)
        lu = np.stack([o1u[:, 2], o1u[:, 3], o2u[:, 0]], 1); ln = np.stack([o1n[:, 2], o1n[:, 3], o2n[:, 0]], 1)
        noisy = {"gravity": lambda x: apply_rand(x, noise.get("gravity"), gu, gn), "ang_vel": lambda x: apply_rand(x, noise.get("ang_vel"), au, an),
                 "lin_vel": lambda x: apply_rand(x, noise.get("lin_vel"), lu, ln), "height": lambda x: apply_rand(x, noise.get("height"), o2u[:, 1], o2n[:, 1]),
                 "dof_pos": lambda x: apply_rand(x, noise.get("dof_pos"), dpu, dpn), "dof_vel": lambda x: apply_rand(x, noise.get("dof_vel"), dvu, dvn)}
        s = {"projected_gravity": proj_g, "base_ang_vel": base_ang, "base_lin_vel": base_lin, "commands": self.cmd, "gait_frequency": self.gait_f,
             "gait_process": self.gait_p, "dof_pos": self.q, "dof_vel": self.qd, "actions": self.actions, "root_states": self.root,
             "base_mass_scaled": self.p["bms"], "push_force": self.push[:, :3], "push_torque": self.push[:, 3:]}
        obs, priv = compute_observations(s, cfg["normalization"], self.default, self.terrain, noisy)
        return obs, priv, feet_store

    def reset(self):
        """T1.reset(): t1.py:294-299 (uses stream offset 64 like the HIP kernel's reset mode)."""
        base_lin = quat_rotate_inverse(self.root[:, 3:7], self.root[:, 7:10])
        base_ang = quat_rotate_inverse(self.root[:, 3:7], self.root[:, 10:13])
        proj_g = quat_rotate_inverse(self.root[:, 3:7], np.tile(np.array([0.0, 0.0, -1.0]), (self.n, 1)))
        feet_pos, _ = self._feet()
        obs, priv, feet_store = self._reset_and_observe(self.step_count, 64, np.ones(self.n, dtype=bool), base_lin, base_ang, proj_g, feet_pos, False)
        self.last_feet = feet_store
        self.last_qd = self.qd.copy()
        self.last_rootvel = self.root[:, 7:13].copy()
        return obs, priv

    def step(self, actions):
        cfg, n, step = self.cfg, self.n, self.step_count
        nz, rnd, rw = cfg["normalization"], cfg["randomization"], cfg["rewards"]
        dec = cfg["control"]["decimation"]
        # ---- pre-physics + substeps (t1.py:439-456)
        self.actions = np.clip(np.asarray(actions, dtype=np.float64), -nz["clip_actions"], nz["clip_actions"])
        targets = self.default + cfg["control"]["action_scale"] * self.actions
        # pushing force acts at the trunk's centre of mass, trunk frame (apply_rigid_body_force_tensors LOCAL_SPACE, t1.py:522-527):
        # wrench about the trunk origin = (f, t + c x f)
        com0 = np.array(self.dyn.model.com[0][:]) + self.p["com_off"].reshape(n, 13, 3)[:, 0]
        wrench = self.push.copy()
        wrench[:, 3:] += np.cross(com0, self.push[:, :3])
        tmean, cf = self.dyn.substeps_batch(dec, self.p["mass_scale"], self.p["com_off"].reshape(n, 39), self.p["foot_mat"].reshape(n, 6), self.p["kp"],
                                            self.p["kd"], self.p["fric"], self.limits["torque_limits"], self.root, self.q, self.qd, targets, self.last_tgt,
                                            self.delay.astype(np.int32), wrench)
        # ---- post-physics (t1.py:460-478)
        quat = self.root[:, 3:7]
        base_lin, base_ang = quat_rotate_inverse(quat, self.root[:, 7:10]), quat_rotate_inverse(quat, self.root[:, 10:13])
        proj_g = quat_rotate_inverse(quat, np.tile(np.array([0.0, 0.0, -1.0]), (n, 1)))
        w = nz["filter_weight"]
        self.filt_lin = base_lin * w + self.filt_lin * (1 - w)
        self.filt_ang = base_ang * w + self.filt_ang * (1 - w)
        feet_pos, feet_quat = self._feet()
        roll, yaw, contact = feet_state(feet_pos, feet_quat, cfg["asset"]["feet_edge_pos"], self.terrain)
        self.ep_len += 1
        cnt = step + 1
        self.gait_p = np.fmod(self.gait_p + self.dt * self.gait_f, 1.0)
        # ---- kick / push (t1.py:499-527)
        ki, pi_, pd_ = (int(math.ceil(rnd[k] / self.dt)) for k in ("kick_interval_s", "push_interval_s", "push_duration_s"))
        if cnt % ki == 0:
            k0u, k0n = self._r4(RS_KICK0, step, 0); k1u, k1n = self._r4(RS_KICK1, step, 0)
            for a in range(3):
                self.root[:, 7 + a] = apply_rand(self.root[:, 7 + a], rnd.get("kick_lin_vel"), k0u[:, a], k0n[:, a])
            self.root[:, 10] = apply_rand(self.root[:, 10], rnd.get("kick_ang_vel"), k0u[:, 3], k0n[:, 3])
            self.root[:, 11] = apply_rand(self.root[:, 11], rnd.get("kick_ang_vel"), k1u[:, 0], k1n[:, 0])
            self.root[:, 12] = apply_rand(self.root[:, 12], rnd.get("kick_ang_vel"), k1u[:, 1], k1n[:, 1])
        if cnt % pi_ == 0:
            p0u, p0n = self._r4(RS_PUSH0, step, 0); p1u, p1n = self._r4(RS_PUSH1, step, 0)
            for a in range(3):
                self.push[:, a] = apply_rand(np.zeros(n), rnd.get("push_force"), p0u[:, a], p0n[:, a])
            self.push[:, 3] = apply_rand(np.zeros(n), rnd.get("push_torque"), p0u[:, 3], p0n[:, 3])
            self.push[:, 4] = apply_rand(np.zeros(n), rnd.get("push_torque"), p1u[:, 0], p1n[:, 0])
            self.push[:, 5] = apply_rand(np.zeros(n), rnd.get("push_torque"), p1u[:, 1], p1n[:, 1])
        elif cnt % pi_ == pd_:
            self.push[:] = 0.0
        # ---- termination + rewards (t1.py:551-572)
        # net contact force per body of the LAST substep (contact_collection: last substep, T1.yaml:56); the penalised / terminating body
        # lists are the name matches of t1.py:85-100
        contact_forces = cf
        names = list(self.m.body_names)
        pen_idx = sorted({i for key in rw.get("penalize_contacts_on", []) for i, nm in enumerate(names) if key in nm})
        term_idx = sorted({i for key in rw.get("terminate_contacts_on", []) for i, nm in enumerate(names) if key in nm})
        reset, tout = check_termination(self.root, self.ep_len, self.cmd_time, self.terrain, rw, self.dt, contact_forces, term_idx)
        s = {"root_states": self.root, "commands": self.cmd, "filtered_lin_vel": self.filt_lin, "filtered_ang_vel": self.filt_ang,
             "base_ang_vel": base_ang, "projected_gravity": proj_g, "torques": tmean, "dof_pos": self.q, "dof_vel": self.qd, "last_dof_vel": self.last_qd,
             "last_root_vel": self.last_rootvel, "actions": self.actions, "last_actions": self.last_actions, "contact_forces": contact_forces,
             "penalized_contact_indices": pen_idx, "feet_pos": feet_pos, "last_feet_pos": self.last_feet, "feet_contact": contact, "feet_roll": roll,
             "feet_yaw": yaw, "episode_length_buf": self.ep_len, "gait_frequency": self.gait_f, "gait_process": self.gait_p}
        terms = reward_terms(s, rw, self.dt, self.limits, self.terrain)
        rew, scaled = total_reward(terms, self.scales, rw["only_positive_rewards"])
        derived = {"feet_pos": feet_pos, "feet_roll": roll, "feet_yaw": yaw, "feet_contact": contact, "torques": tmean, "base_lin_vel": base_lin,
                   "base_ang_vel": base_ang, "projected_gravity": proj_g, "contact": cf[:, [6, 12], :], "contact_all": cf}
        # ---- reset / teleport / resample / observe (t1.py:485-490)
        obs, priv, feet_store = self._reset_and_observe(step, 0, reset, base_lin, base_ang, proj_g, feet_pos, True)
        # ---- history (t1.py:492-495)
        self.last_actions = self.actions.copy(); self.last_qd = self.qd.copy()
        self.last_rootvel = self.root[:, 7:13].copy(); self.last_feet = feet_store
        self.curr_prob_read = self.curr_prob.copy()
        self.step_count += 1
        return obs, priv, rew, reset, tout, scaled, derived
