"""T1 task: the environment class the PPO runner drives.

Same contract as reference `envs/t1.py:24-730` (class `T1`): `cls(cfg)`, attributes `num_envs / num_obs /
num_privileged_obs / num_actions / dt / curriculum_prob / mean_*_level / max_*_level`, `reset() -> (obs, extras)`,
`step(actions) -> (obs, rew, done, extras)` with `extras = {"privileged_obs", "time_outs", "rew_terms"}`; returned
tensors are views of internal buffers that the next `step` overwrites (reference behaviour).

All per-step work of the reference class (`step`, `_reset_idx`, `_resample_commands`, `_kick_robots`, `_push_robots`,
`_refresh_feet_state`, `_check_termination`, `_compute_reward` + 26 `_reward_*`, `_compute_observations`, t1.py:294-730)
runs inside one HIP launch, `bg_env_step` (booster_gym_amd/csrc/bg_env.h); this file only does what the reference does
once at start-up (`_create_envs` t1.py:33-137, `_get_env_origins` :169-185, `_init_buffers` :187-272,
`_prepare_reward_function` :274-292) and marshals tensors across the C ABI.
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from ..utils.urdf import load_model
from ..utils.utils import rand_spec
from .base_task import BaseTask


def _draw(rng, shape, params):
    """Set-up time randomisation on numpy arrays: returns (noise_value, raw_draw) or (None, None) when disabled."""
    if params is None:
        return None, None
    mode, a, b = rand_spec(params)
    raw = rng.standard_normal(shape) if mode <= 2 else rng.random(shape)
    val = a + b * raw if mode <= 2 else a + (b - a) * raw
    return val, raw


def _apply(x, params, rng, return_raw=False):
    val, raw = _draw(rng, x.shape, params)
    if val is None:
        return (x, np.zeros_like(x)) if return_raw else x
    out = x + val if params["operation"] == "additive" else x * val
    return (out, raw) if return_raw else out


class T1(BaseTask):
    def __init__(self, cfg):
        super().__init__(cfg)
        self._lib = _lib.load()
        self._model = None
        self._env = None
        self._create_envs()
        self._init_buffers()
        self._prepare_reward_function()
        self._create_native()

    # ------------------------------------------------------------------ set-up (t1.py:33-185)
    def _create_envs(self):
        cfg = self.cfg
        self.num_envs = int(cfg["env"]["num_envs"])
        asset_cfg = cfg["asset"]
        self.model = load_model(asset_cfg["file"], asset_cfg.get("collapse_fixed_joints", True))
        m = self.model
        self.num_dofs, self.num_bodies = m.num_dofs, m.num_bodies
        self.dof_names, self.body_names = list(m.dof_names), list(m.body_names)
        self.dof_pos_limits = torch.tensor(np.stack([m.dof_lower, m.dof_upper], axis=1), dtype=torch.float, device=self.device)
        self.dof_vel_limits = torch.tensor(m.dof_velocity, dtype=torch.float, device=self.device)
        self.torque_limits = torch.tensor(m.dof_effort, dtype=torch.float, device=self.device)

        N, nd, nb = self.num_envs, self.num_dofs, self.num_bodies
        rng = np.random.default_rng(int(cfg["basic"].get("seed", 0)) + 7919 * int(cfg["basic"].get("rank", 0)))
        self._rng = rng
        kp, kd = np.zeros((N, nd)), np.zeros((N, nd))
        for i, name in enumerate(self.dof_names):  # substring match on the DoF name (t1.py:72-80)
            found = False
            for key in cfg["control"]["stiffness"].keys():
                if key in name:
                    kp[:, i] = cfg["control"]["stiffness"][key]
                    kd[:, i] = cfg["control"]["damping"][key]
                    found = True
            if not found:
                raise ValueError(f"PD gain of joint {name} were not defined")
        rnd = cfg["randomization"]
        self._kp = _apply(kp, rnd.get("dof_stiffness"), rng)
        self._kd = _apply(kd, rnd.get("dof_damping"), rng)
        self._fric = _apply(np.zeros((N, nd)), rnd.get("dof_friction"), rng)

        self.base_indice = m.find_body(asset_cfg["base_name"])
        if self.base_indice != 0:
            raise ValueError("asset.base_name must be the root link of the robot")
        pen = []
        for key in cfg["rewards"]["penalize_contacts_on"]:
            pen.extend([s for s in self.body_names if key in s])
        term = []
        for key in cfg["rewards"]["terminate_contacts_on"]:
            term.extend([s for s in self.body_names if key in s])
        self.penalized_contact_indices = torch.tensor([m.find_body(s) for s in pen], dtype=torch.long, device=self.device)
        self.termination_contact_indices = torch.tensor([m.find_body(s) for s in term], dtype=torch.long, device=self.device)
        self.feet_indices = torch.tensor([m.find_body(s) for s in asset_cfg["foot_names"]], dtype=torch.long, device=self.device)
        if self.feet_indices.tolist() != [6, 12]:
            raise ValueError("asset.foot_names must name the last link of the left and of the right leg")

        # rigid-body randomisation (t1.py:139-160): base com / mass, other com / mass; base_mass_scaled keeps the RAW draws
        mass_scale, com_off, bms = np.ones((N, nb)), np.zeros((N, nb, 3)), np.zeros((N, 4))
        for b in range(nb):
            ck, mk = ("base_com", "base_mass") if b == self.base_indice else ("other_com", "other_mass")
            c, craw = _apply(np.zeros((N, 3)), rnd.get(ck), rng, return_raw=True)
            s, sraw = _apply(np.ones((N,)), rnd.get(mk), rng, return_raw=True)
            com_off[:, b, :], mass_scale[:, b] = c, s
            if b == self.base_indice:
                bms[:, 0:3], bms[:, 3] = craw, sraw
        self._mass_scale, self._com_off, self._bms = mass_scale, com_off, bms
        # foot shape material (t1.py:162-167): friction, compliance, restitution drawn on top of 0.0
        fm = np.zeros((N, 2, 3))
        for k, key in enumerate(("friction", "compliance", "restitution")):
            fm[:, :, k] = _apply(np.zeros((N, 2)), rnd.get(key), rng)
        if rnd.get("friction") is None:
            fm[:, :, 0] = cfg["terrain"]["static_friction"]
        if rnd.get("compliance") is None:
            fm[:, :, 1] = 1.0
        fm[:, :, 1] = np.maximum(fm[:, :, 1], 1e-3)
        self._foot_mat = fm
        self.base_mass_scaled = torch.tensor(bms, dtype=torch.float, device=self.device)
        self._get_env_origins()

    def _get_env_origins(self):
        N = self.num_envs
        origins = np.zeros((N, 3))
        if self.cfg["terrain"]["type"] == "plane":
            num_cols = np.floor(np.sqrt(N))
            num_rows = np.ceil(N / num_cols)
            xx, yy = np.meshgrid(np.arange(num_rows), np.arange(num_cols), indexing="ij")
            spacing = self.cfg["env"]["env_spacing"]
            origins[:, 0] = spacing * xx.flatten()[:N]
            origins[:, 1] = spacing * yy.flatten()[:N]
        else:
            t = self.terrain
            num_cols = max(1.0, np.floor(np.sqrt(N * t.env_length / t.env_width)))
            num_rows = np.ceil(N / num_cols)
            xx, yy = np.meshgrid(np.arange(num_rows), np.arange(num_cols), indexing="ij")
            origins[:, 0] = t.env_width / (num_rows + 1) * (xx.flatten()[:N] + 1)
            origins[:, 1] = t.env_length / (num_cols + 1) * (yy.flatten()[:N] + 1)
            origins[:, 2] = t.terrain_heights(origins).cpu().numpy()
        self._origins = origins
        self.env_origins = torch.tensor(origins, dtype=torch.float, device=self.device)

    def _init_buffers(self):
        cfg = self.cfg
        self.num_obs = cfg["env"]["num_observations"]
        self.num_privileged_obs = cfg["env"]["num_privileged_obs"]
        self.num_actions = cfg["env"]["num_actions"]
        if (self.num_obs, self.num_privileged_obs, self.num_actions) != (_lib.NUM_OBS, _lib.NUM_PRIV, _lib.NUM_DOFS):
            raise ValueError("this build computes 47 observations, 14 privileged observations and 12 actions (envs/T1.yaml env.*)")
        self.dt = cfg["control"]["decimation"] * cfg["sim"]["dt"]
        N, dev = self.num_envs, self.device
        self.obs_buf = torch.zeros(N, self.num_obs, dtype=torch.float, device=dev)
        self.privileged_obs_buf = torch.zeros(N, self.num_privileged_obs, dtype=torch.float, device=dev)
        self.rew_buf = torch.zeros(N, dtype=torch.float, device=dev)
        self.reset_buf = torch.ones(N, dtype=torch.bool, device=dev)
        self.time_out_buf = torch.zeros(N, dtype=torch.bool, device=dev)
        self._rew_terms = torch.zeros(_lib.NUM_REWARD_TERMS, N, dtype=torch.float, device=dev)
        self.extras = {"rew_terms": {}}
        self.default_dof_pos = torch.zeros(1, self.num_dofs, dtype=torch.float, device=dev)
        dja = cfg["init_state"]["default_joint_angles"]
        for i, name in enumerate(self.dof_names):
            val = dja["default"]
            for key in dja.keys():
                if key in name:
                    val = dja[key]
            self.default_dof_pos[:, i] = val
        self.base_init_state = torch.tensor(
            cfg["init_state"]["pos"] + cfg["init_state"]["rot"] + cfg["init_state"]["lin_vel"] + cfg["init_state"]["ang_vel"],
            dtype=torch.float, device=dev)
        lv, av = cfg["commands"]["lin_vel_levels"], cfg["commands"]["ang_vel_levels"]
        self._curriculum_init = torch.zeros(1 + 2 * lv, 1 + 2 * av, dtype=torch.float, device=dev)
        self._curriculum_init[lv, av] = 1.0
        self._curriculum_shape = (1 + 2 * lv, 1 + 2 * av)
        self.mean_lin_vel_level = self.mean_ang_vel_level = self.max_lin_vel_level = self.max_ang_vel_level = 0.0

    def _prepare_reward_function(self):
        scales = dict(self.cfg["rewards"]["scales"])
        unknown = [k for k in scales if k not in _lib.REWARD_NAMES]
        if unknown:
            raise AttributeError(f"T1 has no reward term(s) {unknown}")  # reference: getattr(self, '_reward_' + name) fails
        self.reward_scales = {k: v * self.dt for k, v in scales.items() if v != 0}
        self.reward_names = list(self.reward_scales.keys())

    # ------------------------------------------------------------------ native objects
    def _cfg_struct(self):
        cfg = self.cfg
        c = _lib.EnvCfg()
        c.num_envs, c.device = self.num_envs, self.sim_device_id
        c.seed = (int(cfg["basic"].get("seed", 0)) & 0xFFFFFFFF) | ((int(cfg["basic"].get("rank", 0)) + 1) << 32)
        c.sim_dt, c.decimation = cfg["sim"]["dt"], cfg["control"]["decimation"]
        for a in range(3):
            c.gravity[a] = cfg["sim"]["gravity"][a]
        ct = cfg.get("contact", {}) or {}
        c.contact_k = float(ct.get("stiffness", 4.0e4))
        c.contact_d = float(ct.get("damping", 600.0))
        c.contact_ramp = float(ct.get("damping_ramp", 1.0e-3))
        c.friction_visc = float(ct.get("friction_viscosity", 1.0e4))
        c.limit_k = float(ct.get("joint_limit_stiffness", 2000.0))
        c.limit_d = float(ct.get("joint_limit_damping", 20.0))
        c.clamp_qd = int(bool(ct.get("clamp_dof_velocity", True)))
        c.terrain_mu = 0.5 * (cfg["terrain"]["static_friction"] + cfg["terrain"]["dynamic_friction"])
        c.terrain_restitution = cfg["terrain"]["restitution"]
        nz = cfg["normalization"]
        c.action_scale, c.clip_actions = cfg["control"]["action_scale"], nz["clip_actions"]
        c.norm_gravity, c.norm_lin_vel, c.norm_ang_vel = nz["gravity"], nz["lin_vel"], nz["ang_vel"]
        c.norm_dof_pos, c.norm_dof_vel, c.filter_weight = nz["dof_pos"], nz["dof_vel"], nz["filter_weight"]
        c.norm_push_force, c.norm_push_torque = nz["push_force"], nz["push_torque"]
        for j in range(self.num_dofs):
            c.default_dof_pos[j] = float(self.default_dof_pos[0, j])
        for k in range(13):
            c.base_init_state[k] = float(self.base_init_state[k])

        def setr(dst, params):
            dst.mode, dst.a, dst.b = rand_spec(params)

        nc, rc = cfg["noise"], cfg["randomization"]
        for key in ("gravity", "lin_vel", "ang_vel", "dof_pos", "dof_vel", "height"):
            setr(getattr(c, "noise_" + key), nc.get(key))
        for key in ("init_dof_pos", "init_base_pos_xy", "init_base_lin_vel_xy", "kick_lin_vel", "kick_ang_vel", "push_force", "push_torque"):
            setr(getattr(c, key), rc.get(key))
        c.kick_interval = int(math.ceil(rc["kick_interval_s"] / self.dt))
        c.push_interval = int(math.ceil(rc["push_interval_s"] / self.dt))
        c.push_duration = int(math.ceil(rc["push_duration_s"] / self.dt))
        par = cfg.get("parallel", {}) or {}
        c.shared_reset_noise = int(bool(par.get("shared_reset_noise", True)))
        c.exact_still_count = int(bool(par.get("exact_still_count", False)))
        c.same_step_curriculum = int(bool(par.get("same_step_curriculum", False)))
        cm = cfg["commands"]
        for k in range(2):
            c.cmd_lin_vel_x[k], c.cmd_lin_vel_y[k] = cm["lin_vel_x"][k], cm["lin_vel_y"][k]
            c.cmd_ang_vel_yaw[k], c.cmd_gait_frequency[k] = cm["ang_vel_yaw"][k], cm["gait_frequency"][k]
            c.resample_steps[k] = int(cm["resampling_time_s"][k] / self.dt)
        c.still_proportion = cm["still_proportion"]
        c.curriculum = int(bool(cm.get("curriculum", False)))
        c.lin_vel_levels, c.ang_vel_levels = int(cm["lin_vel_levels"]), int(cm["ang_vel_levels"])
        c.curriculum_update_rate = cm["update_rate"]
        c.lin_vel_x_resolution, c.lin_vel_y_resolution, c.ang_vel_resolution = cm["lin_vel_x_resolution"], cm["lin_vel_y_resolution"], cm["ang_vel_resolution"]
        c.episode_length_toler, c.lin_vel_x_toler = cm["episode_length_toler"], cm["lin_vel_x_toler"]
        c.lin_vel_y_toler, c.ang_vel_yaw_toler = cm["lin_vel_y_toler"], cm["ang_vel_yaw_toler"]
        rw = cfg["rewards"]
        for k, name in enumerate(_lib.REWARD_NAMES):
            c.reward_scale[k] = self.reward_scales.get(name, 0.0)
        c.only_positive_rewards = int(bool(rw["only_positive_rewards"]))
        c.tracking_sigma, c.base_height_target = rw["tracking_sigma"], rw["base_height_target"]
        c.soft_dof_pos_limit, c.soft_dof_vel_limit, c.soft_torque_limit = rw["soft_dof_pos_limit"], rw["soft_dof_vel_limit"], rw["soft_torque_limit"]
        c.swing_period, c.feet_distance_ref = rw["swing_period"], rw["feet_distance_ref"]
        c.max_episode_length = int(math.ceil(rw["episode_length_s"] / self.dt))
        c.terminate_height, c.terminate_vel = rw["terminate_height"], rw["terminate_vel"]
        if self.terrain.type == "plane":
            c.terrain_type = 0
        else:
            c.terrain_type = 1
            c.terrain_env_width, c.terrain_env_length, c.terrain_border = self.terrain.env_width, self.terrain.env_length, self.terrain.border_size
        sd = str(cfg["sim"].get("state_dtype", "fp32")).lower()
        if sd not in ("fp32", "float32", "fp16", "float16", "half"):
            raise ValueError(f"sim.state_dtype must be fp32 or fp16, got {sd!r}")
        c.state_fp16 = int(sd in ("fp16", "float16", "half"))
        c.body_gate_height = float(cfg.get("contact", {}).get("body_gate_height", 0.45))
        # asset.self_collisions is Isaac Gym's collision-filter bitmask (create_actor, t1.py:128): 0 = the actor's shapes collide with each other
        c.self_collisions = int(int(cfg["asset"].get("self_collisions", 0)) == 0)
        c.self_k = float(ct.get("self_stiffness", 4.0e4))
        c.self_d = float(ct.get("self_damping", 150.0))
        c.self_mu = float(ct.get("self_friction", 1.0))
        c.self_visc = float(ct.get("self_friction_viscosity", 100.0))
        c.penalized_body_mask = sum(1 << int(b) for b in set(self.penalized_contact_indices.tolist()))
        c.terminate_body_mask = sum(1 << int(b) for b in set(self.termination_contact_indices.tolist()))
        return c

    def _model_struct(self):
        m = self.model
        d = _lib.ModelDesc()
        d.num_bodies, d.num_dofs = m.num_bodies, m.num_dofs
        for b in range(m.num_bodies):
            d.parent[b], d.joint_axis[b], d.mass[b] = int(m.parent[b]), int(m.joint_axis[b]), float(m.mass[b])
            for a in range(3):
                d.body_pos[b][a], d.com[b][a] = float(m.body_pos[b, a]), float(m.com[b, a])
            for a in range(6):
                d.inertia[b][a] = float(m.inertia[b, a])
        for j in range(m.num_dofs):
            d.dof_lower[j], d.dof_upper[j] = float(m.dof_lower[j]), float(m.dof_upper[j])
            d.dof_velocity[j], d.dof_effort[j] = float(m.dof_velocity[j]), float(m.dof_effort[j])
        fe = self.cfg["asset"]["feet_edge_pos"]
        if len(fe) != 4:
            raise ValueError("asset.feet_edge_pos must list the 4 sole corners")
        for k in range(4):
            for a in range(3):
                d.feet_edge_pos[k][a] = float(fe[k][a])
        # non-foot collision shapes (URDF <collision>) as contact spheres; contact.body_contacts: false switches them off
        sph = m.contact_spheres(exclude_bodies=self.feet_indices.tolist()) if self.cfg.get("contact", {}).get("body_contacts", True) else []
        if len(sph) > _lib.MAX_BODY_SPHERES:
            raise ValueError(f"{len(sph)} body contact spheres, the library holds {_lib.MAX_BODY_SPHERES}")
        d.num_body_spheres = len(sph)
        for k, (b, c, r) in enumerate(sph):
            d.sphere_body[k], d.sphere_radius[k] = b, r
            for a in range(3):
                d.sphere_pos[k][a] = c[a]
        # self-collision capsules (shank cylinders, foot boxes); asset.self_collisions != 0 filters them out as in Isaac Gym
        if int(self.cfg["asset"].get("self_collisions", 0)) == 0:
            for leg, caps in enumerate(m.self_collision_capsules(self.feet_indices.tolist())):
                for k, (_, a, b, r) in enumerate(caps):
                    d.self_capsule_r[leg][k] = r
                    for ax in range(3):
                        d.self_capsule_a[leg][k][ax], d.self_capsule_b[leg][k][ax] = a[ax], b[ax]
        return d

    def _create_native(self):
        lib = self._lib
        torch.cuda.set_device(self.sim_device_id)
        mh = C.c_void_p()
        desc = self._model_struct()
        _lib.check(lib.bg_model_create(C.byref(desc), C.byref(mh)), "bg_model_create")
        self._model = mh
        self._cfg_c = self._cfg_struct()
        eh = C.c_void_p()
        _lib.check(lib.bg_env_create(C.byref(self._cfg_c), mh, C.byref(eh)), "bg_env_create")
        self._env = eh
        if self.terrain.type != "plane":
            hf = np.ascontiguousarray(self.terrain.height_field_raw, dtype=np.int16)
            _lib.check(lib.bg_env_set_heightfield(eh, _lib.ptr(hf), hf.shape[0], hf.shape[1], self.terrain.border_pixels,
                                                  self.terrain.horizontal_scale, self.terrain.vertical_scale), "bg_env_set_heightfield")
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        arrs = [f32(self._kp), f32(self._kd), f32(self._fric), f32(self._mass_scale), f32(self._com_off), f32(self._foot_mat),
                f32(self._bms), f32(self._origins)]
        _lib.check(lib.bg_env_set_params(eh, *[_lib.ptr(a) for a in arrs]), "bg_env_set_params")
        _lib.check(lib.bg_env_bind_outputs(eh, _lib.ptr(self.obs_buf), _lib.ptr(self.privileged_obs_buf), _lib.ptr(self.rew_buf),
                                           _lib.ptr(self.reset_buf), _lib.ptr(self.time_out_buf), _lib.ptr(self._rew_terms)),
                   "bg_env_bind_outputs")
        self.extras["privileged_obs"] = self.privileged_obs_buf
        self.extras["time_outs"] = self.time_out_buf
        for name in self.reward_names:
            self.extras["rew_terms"][name] = self._rew_terms[_lib.REWARD_NAMES.index(name)]
        self._actions_scratch = torch.zeros(self.num_envs, self.num_actions, dtype=torch.float, device=self.device)
        self._stale_tout = bool((self.cfg.get("parallel", {}) or {}).get("stale_time_outs", False))
        self._time_outs_at_last_reset = torch.zeros_like(self.time_out_buf)  # T1.reset(): _reset_idx(all) binds the initial (all False) buffer

    def __del__(self):
        try:
            if self._env:
                self._lib.bg_env_destroy(self._env)
                self._env = None
            if self._model:
                self._lib.bg_model_destroy(self._model)
                self._model = None
        except Exception:
            pass

    # ------------------------------------------------------------------ L3 contract
    def reset(self):
        """Reset all robots (reference t1.py:294-299)."""
        _lib.check(self._lib.bg_env_reset(self._env, _lib.current_stream_ptr()), "bg_env_reset")
        return self.obs_buf, self.extras

    def _stale_time_outs(self, done, time_outs):
        """parallel.stale_time_outs: the reference rebinds extras["time_outs"] only inside _reset_idx (t1.py:317), i.e. on steps in
        which at least one env was reset; on every other step the runner reads the flags of the last step that had one.  One host sync."""
        if bool(done.any()):
            self._time_outs_at_last_reset = time_outs.clone()
        else:
            time_outs.copy_(self._time_outs_at_last_reset)

    def step(self, actions):
        """One control step = `decimation` physics substeps + task logic (reference t1.py:437-497)."""
        a = self._as_actions(actions)
        _lib.check(self._lib.bg_env_step(self._env, _lib.ptr(a), _lib.current_stream_ptr()), "bg_env_step")
        if self._stale_tout:
            self._stale_time_outs(self.reset_buf, self.time_out_buf)
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def step_to(self, actions, obs, privileged_obs, rew, done, time_outs):
        """`step` that writes its per-step outputs straight into rows of the caller's rollout buffers."""
        a = self._as_actions(actions)
        for t in (obs, privileged_obs, rew, done, time_outs):
            if not (t.is_cuda and t.is_contiguous()):
                raise RuntimeError("step_to needs contiguous CUDA output tensors")
        _lib.check(self._lib.bg_env_step_to(self._env, _lib.ptr(a), _lib.ptr(obs), _lib.ptr(privileged_obs), _lib.ptr(rew), _lib.ptr(done),
                                            _lib.ptr(time_outs), _lib.current_stream_ptr()), "bg_env_step_to")
        if self._stale_tout:
            self._stale_time_outs(done, time_outs)

    def _as_actions(self, actions):
        if actions.shape != (self.num_envs, self.num_actions):
            raise ValueError(f"actions must have shape {(self.num_envs, self.num_actions)}, got {tuple(actions.shape)}")
        if actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous() and actions.device.index == self.sim_device_id:
            return actions
        self._actions_scratch.copy_(actions)
        return self._actions_scratch

    # ------------------------------------------------------------------ state access (what the reference exposes as tensor attributes)
    def get_field(self, name):
        comps, is_int = C.c_int32(), C.c_int32()
        _lib.check(self._lib.bg_env_field_info(self._env, name.encode(), C.byref(comps), C.byref(is_int)), "bg_env_field_info")
        out = torch.empty(self.num_envs, comps.value, dtype=torch.int32 if is_int.value else torch.float32, device=self.device)
        _lib.check(self._lib.bg_env_get_field(self._env, name.encode(), _lib.ptr(out), _lib.current_stream_ptr()), "bg_env_get_field")
        return out

    def set_field(self, name, value):
        comps, is_int = C.c_int32(), C.c_int32()
        _lib.check(self._lib.bg_env_field_info(self._env, name.encode(), C.byref(comps), C.byref(is_int)), "bg_env_field_info")
        v = torch.as_tensor(value, device=self.device).to(torch.int32 if is_int.value else torch.float32).reshape(self.num_envs, comps.value).contiguous()
        _lib.check(self._lib.bg_env_set_field(self._env, name.encode(), _lib.ptr(v), _lib.current_stream_ptr()), "bg_env_set_field")
        torch.cuda.current_stream().synchronize()  # `v` may be a temporary

    def episode_stats(self, reset=True):
        """Device-side replacement of recorder.py:36-53: float[30] = (finished episodes, sum of lengths, sum of reward,
        26 per-term sums, resets caused by a non-finite state)."""
        s = torch.empty(4 + _lib.NUM_REWARD_TERMS, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.bg_env_get_field(self._env, b"episode_stats", _lib.ptr(s), _lib.current_stream_ptr()), "bg_env_get_field")
        if reset:
            z = torch.zeros_like(s)
            _lib.check(self._lib.bg_env_set_field(self._env, b"episode_stats", _lib.ptr(z), _lib.current_stream_ptr()), "bg_env_set_field")
        return s

    def forward_dynamics(self, root, dof_pos, dof_vel, tau, base_wrench=None, packed=False):
        """d/dt of (root lin vel, root ang vel, dof vel) for the given states: [N,18].  Uses this env's model / randomisation / terrain.
        packed: through the kernel with one env per wavefront lane and both legs in 64-bit register pairs (bg_env_forward_dynamics_packed)
        instead of one leg per lane."""
        qacc = torch.empty(self.num_envs, 18, dtype=torch.float32, device=self.device)
        args = [t.contiguous().float() for t in (root, dof_pos, dof_vel, tau)]
        w = base_wrench.contiguous().float() if base_wrench is not None else None
        name = "bg_env_forward_dynamics_packed" if packed else "bg_env_forward_dynamics"
        _lib.check(getattr(self._lib, name)(self._env, *[_lib.ptr(t) for t in args], _lib.ptr(w), _lib.ptr(qacc), _lib.current_stream_ptr()), name)
        torch.cuda.current_stream().synchronize()
        return qacc

    # ---- command curriculum state (t1.py:249-262; r/w by the runner for checkpoints, runner.py:91,211)
    @property
    def curriculum_prob(self):
        out = torch.empty(self._curriculum_shape, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.bg_env_get_curriculum(self._env, _lib.ptr(out), _lib.current_stream_ptr()), "bg_env_get_curriculum")
        return out

    @curriculum_prob.setter
    def curriculum_prob(self, value):
        v = torch.as_tensor(value, dtype=torch.float32).to(self.device).reshape(self._curriculum_shape).contiguous()
        _lib.check(self._lib.bg_env_set_curriculum(self._env, _lib.ptr(v), _lib.current_stream_ptr()), "bg_env_set_curriculum")
        torch.cuda.current_stream().synchronize()

    def refresh_curriculum_levels(self):
        """mean / max |level| over all envs (t1.py:421-424); one small device->host read, called once per iteration by the runner."""
        if not self.cfg["commands"].get("curriculum", False):
            return
        lin = self.get_field("env_curriculum_level_lin").abs().float()
        ang = self.get_field("env_curriculum_level_ang").abs().float()
        self.mean_lin_vel_level, self.mean_ang_vel_level = float(lin.mean()), float(ang.mean())
        self.max_lin_vel_level, self.max_ang_vel_level = float(lin.max()), float(ang.max())

    @property
    def root_states(self):
        return self.get_field("root_states")

    @property
    def dof_pos(self):
        return self.get_field("dof_pos")

    @property
    def dof_vel(self):
        return self.get_field("dof_vel")

    @property
    def commands(self):
        return self.get_field("commands")

    @property
    def episode_length_buf(self):
        return self.get_field("episode_length_buf").squeeze(-1)

    @property
    def common_step_counter(self):
        return int(self._lib.bg_env_step_count(self._env))

    @common_step_counter.setter
    def common_step_counter(self, value):
        _lib.check(self._lib.bg_env_set_step_count(self._env, int(value)), "bg_env_set_step_count")
