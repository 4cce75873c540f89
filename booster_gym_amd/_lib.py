"""ctypes binding of libbooster_gym_amd.so (include/booster_gym_amd.h).

The library is the product: if it is missing or cannot be loaded this module raises --
there is no Python / CPU fallback for the simulator or the fused PPO kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BG_LIB") or os.path.join(_HERE, "libbooster_gym_amd.so")  # BG_LIB: alternative build (kernel experiments)

NUM_BODIES, NUM_DOFS, NUM_OBS, NUM_PRIV, NUM_REWARD_TERMS = 13, 12, 47, 14, 26

# order = envs/T1.yaml rewards.scales = enum in include/booster_gym_amd.h
REWARD_NAMES = [
    "survival", "tracking_lin_vel_x", "tracking_lin_vel_y", "tracking_ang_vel", "base_height", "orientation", "torques",
    "torque_tiredness", "power", "lin_vel_z", "ang_vel_xy", "dof_vel", "dof_acc", "root_acc", "action_rate", "dof_pos_limits",
    "dof_vel_limits", "torque_limits", "collision", "feet_slip", "feet_vel_z", "feet_yaw_diff", "feet_yaw_mean", "feet_roll",
    "feet_distance", "feet_swing",
]


MAX_BODY_SPHERES = 16


class ModelDesc(C.Structure):
    _fields_ = [
        ("num_bodies", C.c_int32), ("num_dofs", C.c_int32),
        ("parent", C.c_int32 * NUM_BODIES), ("joint_axis", C.c_int32 * NUM_BODIES),
        ("body_pos", C.c_float * 3 * NUM_BODIES), ("mass", C.c_float * NUM_BODIES), ("com", C.c_float * 3 * NUM_BODIES),
        ("inertia", C.c_float * 6 * NUM_BODIES),
        ("dof_lower", C.c_float * NUM_DOFS), ("dof_upper", C.c_float * NUM_DOFS), ("dof_velocity", C.c_float * NUM_DOFS),
        ("dof_effort", C.c_float * NUM_DOFS),
        ("feet_edge_pos", C.c_float * 3 * 4),
        ("num_body_spheres", C.c_int32), ("sphere_body", C.c_int32 * MAX_BODY_SPHERES), ("sphere_pos", C.c_float * 3 * MAX_BODY_SPHERES),
        ("sphere_radius", C.c_float * MAX_BODY_SPHERES),
        ("self_capsule_a", C.c_float * 3 * 2 * 2), ("self_capsule_b", C.c_float * 3 * 2 * 2), ("self_capsule_r", C.c_float * 2 * 2),
    ]


class AssetOptions(C.Structure):
    _fields_ = [("collapse_fixed_joints", C.c_int32), ("body_contacts", C.c_int32), ("self_collisions", C.c_int32), ("foot_names", C.c_char_p * 2), ("feet_edge_pos", C.c_float * 3 * 4)]


class WgradProblem(C.Structure):
    _fields_ = [("G", C.c_void_p), ("A", C.c_void_p), ("dW", C.c_void_p), ("scratch", C.c_void_p), ("M", C.c_int32), ("C_out", C.c_int32), ("C_in", C.c_int32),
                ("C_in_real", C.c_int32), ("slices", C.c_int32), ("tiles_per_workgroup", C.c_int32)]


class ReduceProblem(C.Structure):
    """bg_reduce_problem: a deferred fixed-order reduction (include/booster_gym_amd.h)."""
    _fields_ = [("partial", C.c_void_p), ("groups", C.c_int32), ("record", C.c_int32), ("n_out", C.c_int32), ("out", C.c_void_p * 3), ("n", C.c_int32 * 3),
                ("stat_base", C.c_uint64), ("n_stat", C.c_int32), ("n_ls", C.c_int32), ("stat_skip", C.c_uint32), ("entropy_coef", C.c_double),
                ("grad_logstd", C.c_void_p), ("stats", C.c_void_p)]


class ParamMirror(C.Structure):
    """bg_param_mirror: a transposed / re-strided copy of one weight matrix kept current by bg_optimizer_step (include/booster_gym_amd.h)."""
    _fields_ = [("offset", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("transpose", C.c_int32), ("ld", C.c_int32), ("pad", C.c_int32),
                ("dst", C.c_void_p)]


class MlpChain(C.Structure):
    """bg_mlp_chain: the forward chain of one network for bg_mlp_chain_forward_group (include/booster_gym_amd.h)."""
    _fields_ = [("M", C.c_int32), ("K0", C.c_int32), ("N1", C.c_int32), ("N2", C.c_int32), ("N3", C.c_int32), ("workgroups", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("X", "W1", "b1", "W2", "b2", "W3", "b3", "Y1", "Y2", "Y3", "v_w", "v_b", "v_out")]


class MlpChainSplit(C.Structure):
    """bg_mlp_chain_split: the forward chain of one network on the bf16 matrix pipe (9 exact products) for bg_mlp_chain_forward_split."""
    _fields_ = [("M", C.c_int32), ("K0", C.c_int32), ("N1", C.c_int32), ("N2", C.c_int32), ("N3", C.c_int32), ("workgroups", C.c_int32),
                ("alternate", C.c_int32), ("pad", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("X", "P1", "P2", "P3", "b1", "b2", "b3", "Y1", "Y2", "Y3", "v_w", "v_b", "v_out")]


class MlpChainSplitBwd(C.Structure):
    """bg_mlp_chain_split_bwd: the backward-data chain of one network on the bf16 matrix pipe (9 exact products) for bg_mlp_chain_backward_split."""
    _fields_ = [("M", C.c_int32), ("N1", C.c_int32), ("N2", C.c_int32), ("N3", C.c_int32), ("workgroups", C.c_int32), ("alternate", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("G3", "PT3", "PT2", "A2", "A1", "G2", "G1", "colsum_partial", "bias_grad2", "bias_grad1")]


class Rand(C.Structure):
    _fields_ = [("mode", C.c_int32), ("a", C.c_float), ("b", C.c_float)]


class EnvCfg(C.Structure):
    _fields_ = [
        ("num_envs", C.c_int32), ("device", C.c_int32), ("seed", C.c_uint64),
        ("sim_dt", C.c_float), ("decimation", C.c_int32), ("gravity", C.c_float * 3),
        ("contact_k", C.c_float), ("contact_d", C.c_float), ("contact_ramp", C.c_float), ("friction_visc", C.c_float),
        ("limit_k", C.c_float), ("limit_d", C.c_float), ("terrain_mu", C.c_float), ("terrain_restitution", C.c_float),
        ("clamp_qd", C.c_int32),
        ("action_scale", C.c_float), ("clip_actions", C.c_float),
        ("norm_gravity", C.c_float), ("norm_lin_vel", C.c_float), ("norm_ang_vel", C.c_float), ("norm_dof_pos", C.c_float),
        ("norm_dof_vel", C.c_float), ("filter_weight", C.c_float), ("norm_push_force", C.c_float), ("norm_push_torque", C.c_float),
        ("default_dof_pos", C.c_float * NUM_DOFS), ("base_init_state", C.c_float * 13),
        ("noise_gravity", Rand), ("noise_lin_vel", Rand), ("noise_ang_vel", Rand), ("noise_dof_pos", Rand), ("noise_dof_vel", Rand),
        ("noise_height", Rand),
        ("init_dof_pos", Rand), ("init_base_pos_xy", Rand), ("init_base_lin_vel_xy", Rand), ("kick_lin_vel", Rand), ("kick_ang_vel", Rand),
        ("push_force", Rand), ("push_torque", Rand),
        ("kick_interval", C.c_int32), ("push_interval", C.c_int32), ("push_duration", C.c_int32), ("shared_reset_noise", C.c_int32),
        ("cmd_lin_vel_x", C.c_float * 2), ("cmd_lin_vel_y", C.c_float * 2), ("cmd_ang_vel_yaw", C.c_float * 2),
        ("cmd_gait_frequency", C.c_float * 2), ("still_proportion", C.c_float), ("resample_steps", C.c_int32 * 2),
        ("curriculum", C.c_int32), ("lin_vel_levels", C.c_int32), ("ang_vel_levels", C.c_int32),
        ("curriculum_update_rate", C.c_float), ("lin_vel_x_resolution", C.c_float), ("lin_vel_y_resolution", C.c_float), ("ang_vel_resolution", C.c_float),
        ("episode_length_toler", C.c_float), ("lin_vel_x_toler", C.c_float), ("lin_vel_y_toler", C.c_float), ("ang_vel_yaw_toler", C.c_float),
        ("reward_scale", C.c_float * NUM_REWARD_TERMS), ("only_positive_rewards", C.c_int32),
        ("tracking_sigma", C.c_float), ("base_height_target", C.c_float), ("soft_dof_pos_limit", C.c_float),
        ("soft_dof_vel_limit", C.c_float), ("soft_torque_limit", C.c_float), ("swing_period", C.c_float), ("feet_distance_ref", C.c_float),
        ("max_episode_length", C.c_int32), ("terminate_height", C.c_float), ("terminate_vel", C.c_float),
        ("terrain_type", C.c_int32), ("terrain_env_width", C.c_float), ("terrain_env_length", C.c_float), ("terrain_border", C.c_float),
        ("state_fp16", C.c_int32),
        ("body_gate_height", C.c_float), ("penalized_body_mask", C.c_int32), ("terminate_body_mask", C.c_int32),
        ("exact_still_count", C.c_int32), ("same_step_curriculum", C.c_int32),
        ("self_collisions", C.c_int32), ("self_k", C.c_float), ("self_d", C.c_float), ("self_mu", C.c_float), ("self_visc", C.c_float),
    ]


# every symbol include/booster_gym_amd.h declares (tests check that the .so exports all of them)
HEAD_SCRATCH_FLOATS = 768 * 1720  # BG_HEAD_SCRATCH_FLOATS

SYMBOLS = [
    "bg_model_create", "bg_model_get", "bg_model_destroy", "bg_model_load_urdf", "bg_model_body_name", "bg_model_dof_name", "bg_model_find_body", "bg_env_create", "bg_env_destroy", "bg_env_set_heightfield",
    "bg_env_set_params", "bg_env_bind_outputs", "bg_env_reset", "bg_env_step", "bg_env_step_to", "bg_env_get_state",
    "bg_env_set_state", "bg_env_get_field", "bg_env_set_field", "bg_env_field_info", "bg_env_get_curriculum", "bg_env_set_curriculum", "bg_env_step_count", "bg_env_set_step_count",
    "bg_env_forward_dynamics", "bg_env_forward_dynamics_packed", "bg_sim_bind_state", "bg_sim_set_actuation", "bg_sim_apply_body_wrench_local", "bg_sim_simulate",
    "bg_sim_refresh_body_state", "bg_sim_write_root_state", "bg_sim_write_dof_state", "bg_gae", "bg_ppo_loss", "bg_gaussian_logp", "bg_actor_sample", "bg_adam_step", "bg_adapt_lr", "bg_optimizer_step", "bg_elu_backward_colsum", "bg_mlp_layer_forward", "bg_mlp_chain_forward", "bg_critic_values_gae", "bg_mlp_chain_forward_group", "bg_mlp_chain_forward_split", "bg_mlp_chain_backward_split", "bg_mlp_split_weights_pm", "bg_mlp_layer_backward", "bg_mlp_split_weights", "bg_mlp_layer_forward_split", "bg_mlp_layer_backward_split", "bg_mlp_weight_grad", "bg_mlp_weight_grad_group", "bg_mlp_weight_grad_group_partial", "bg_update_tail", "bg_update_tail_sums", "bg_mlp_weight_grad_group_split", "bg_mlp_weight_grad_group_split_partial",
    "bg_critic_head_forward", "bg_actor_head", "bg_critic_head_backward",
    "bg_reduce_group", "bg_actor_head_partial", "bg_critic_head_backward_partial", "bg_mlp_layer_backward_partial",
    "bg_last_error", "bg_version",
]

_lib = None


def load():
    """Load the native library (once).  Raises if it has not been built: `python -c 'import __graft_entry__ as g; g.build()'`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing. Build it with `make -C booster_gym_amd/csrc` (hipcc, gfx950). "
            "booster_gym_amd has no CPU / pure-Python fallback for its kernels."
        )
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32, u64, i64 = C.c_void_p, C.c_int32, C.c_float, C.c_uint64, C.c_int64
    sig = {
        "bg_model_create": (i32, [C.POINTER(ModelDesc), C.POINTER(vp)]),
        "bg_model_get": (i32, [vp, C.POINTER(ModelDesc)]),
        "bg_model_load_urdf": (i32, [C.c_char_p, C.POINTER(AssetOptions), C.POINTER(vp)]),
        "bg_model_body_name": (C.c_char_p, [vp, i32]),
        "bg_model_dof_name": (C.c_char_p, [vp, i32]),
        "bg_model_find_body": (i32, [vp, C.c_char_p]),
        "bg_model_destroy": (None, [vp]),
        "bg_env_create": (i32, [C.POINTER(EnvCfg), vp, C.POINTER(vp)]),
        "bg_env_destroy": (None, [vp]),
        "bg_env_set_heightfield": (i32, [vp, vp, i32, i32, i32, f32, f32]),
        "bg_env_set_params": (i32, [vp] + [vp] * 8),
        "bg_env_bind_outputs": (i32, [vp] + [vp] * 6),
        "bg_env_reset": (i32, [vp, vp]),
        "bg_env_step": (i32, [vp, vp, vp]),
        "bg_env_step_to": (i32, [vp, vp, vp, vp, vp, vp, vp, vp]),
        "bg_env_get_state": (i32, [vp, vp, vp, vp, vp]),
        "bg_env_set_state": (i32, [vp, vp, vp, vp]),
        "bg_env_get_field": (i32, [vp, C.c_char_p, vp, vp]),
        "bg_env_set_field": (i32, [vp, C.c_char_p, vp, vp]),
        "bg_env_field_info": (i32, [vp, C.c_char_p, C.POINTER(i32), C.POINTER(i32)]),
        "bg_env_get_curriculum": (i32, [vp, vp, vp]),
        "bg_env_set_curriculum": (i32, [vp, vp, vp]),
        "bg_env_step_count": (i64, [vp]),
        "bg_env_set_step_count": (i32, [vp, i64]),
        "bg_env_forward_dynamics": (i32, [vp, vp, vp, vp, vp, vp, vp, vp]),
        "bg_env_forward_dynamics_packed": (i32, [vp, vp, vp, vp, vp, vp, vp, vp]),
        "bg_sim_bind_state": (i32, [vp, vp, vp, vp, vp]),
        "bg_sim_set_actuation": (i32, [vp, vp, vp]),
        "bg_sim_apply_body_wrench_local": (i32, [vp, vp, vp, vp]),
        "bg_sim_simulate": (i32, [vp, vp]),
        "bg_sim_refresh_body_state": (i32, [vp, vp]),
        "bg_sim_write_root_state": (i32, [vp, vp, i32, vp]),
        "bg_sim_write_dof_state": (i32, [vp, vp, i32, vp]),
        "bg_gae": (i32, [i32, i32, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp]),
        "bg_ppo_loss": (i32, [i32, i32] + [vp] * 10 + [f32, f32, f32] + [vp] * 5),
        "bg_gaussian_logp": (i32, [i32, i32, vp, vp, vp, vp, vp]),
        "bg_actor_sample": (i32, [i32] + [vp] * 10 + [u64, u64, vp, vp, vp]),
        "bg_adam_step": (i32, [i32, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, vp, vp]),
        "bg_adapt_lr": (i32, [vp, f32, f32, f32, f32, vp, vp]),
        "bg_optimizer_step": (i32, [i32, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, vp, i32, i32, vp, vp, vp, i32, i32, f32, f32, f32, f32, vp, vp, i32, vp]),
        "bg_elu_backward_colsum": (i32, [i32, i32, vp, vp, vp, vp, vp]),
        "bg_mlp_layer_forward": (i32, [i32, i32, i32, vp, vp, vp, vp, i32, vp]),
        "bg_mlp_chain_forward": (i32, [i32] * 5 + [vp] * 10 + [vp]),
        "bg_mlp_chain_forward_group": (i32, [vp, i32, vp]),
        "bg_mlp_chain_forward_split": (i32, [vp, i32, vp]),
        "bg_mlp_chain_backward_split": (i32, [vp, i32, C.POINTER(ReduceProblem), vp]),
        "bg_critic_values_gae": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp]),
        "bg_mlp_layer_backward": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
        "bg_mlp_split_weights": (i32, [i32, i32, vp, i32, i32, i32, i32, vp, vp]),
        "bg_mlp_split_weights_pm": (i32, [i32, i32, vp, i32, i32, i32, i32, vp, vp]),
        "bg_mlp_layer_forward_split": (i32, [i32, i32, i32, vp, vp, vp, vp, i32, i32, vp]),
        "bg_mlp_layer_backward_split": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
        "bg_mlp_weight_grad": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, i32, vp]),
        "bg_mlp_weight_grad_group": (i32, [C.POINTER(WgradProblem), i32, vp]),
        "bg_mlp_weight_grad_group_partial": (i32, [C.POINTER(WgradProblem), i32, vp]),
        "bg_update_tail_sums": (i32, [C.POINTER(WgradProblem), i32, C.POINTER(ReduceProblem), i32, vp, vp]),
        "bg_update_tail": (i32, [C.POINTER(WgradProblem), i32, C.POINTER(ReduceProblem), i32, i32, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, vp, i32, i32, vp, vp,
                                 vp, i32, i32, f32, f32, f32, f32, vp, vp, vp, i32, vp]),
        "bg_mlp_weight_grad_group_split": (i32, [C.POINTER(WgradProblem), i32, i32, vp]),
        "bg_mlp_weight_grad_group_split_partial": (i32, [C.POINTER(WgradProblem), i32, i32, vp]),
        "bg_critic_head_forward": (i32, [i32, vp, vp, vp, vp, vp]),
        "bg_actor_head": (i32, [i32, i32] + [vp] * 10 + [f32, f32, f32] + [vp] * 9),
        "bg_critic_head_backward": (i32, [i32] + [vp] * 11),
        "bg_reduce_group": (i32, [C.POINTER(ReduceProblem), i32, vp]),
        "bg_actor_head_partial": (i32, [i32] + [vp] * 10 + [f32, f32, f32] + [vp] * 8 + [C.POINTER(ReduceProblem), vp]),
        "bg_critic_head_backward_partial": (i32, [i32] + [vp] * 10 + [C.POINTER(ReduceProblem), vp]),
        "bg_mlp_layer_backward_partial": (i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(ReduceProblem), vp]),
        "bg_last_error": (C.c_char_p, []),
        "bg_version": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().bg_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what or 'booster_gym_amd'} failed ({rc}): {msg}")


def current_stream_ptr():
    """hipStream_t of torch's current stream as an integer (0 = default stream)."""
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Raw device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)
