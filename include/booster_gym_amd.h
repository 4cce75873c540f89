/*
 * booster_gym_amd.h -- C ABI of libbooster_gym_amd.so (MI355X / gfx950, HIP).
 *
 * Drop-in boundary for the hot path of Nyro-Robotics/booster_gym: the Isaac Gym
 * tensor API the reference's task code calls plus the
 * GAE / PPO-loss loop of utils/runner.py.  Every entry point is extern "C",
 * takes plain pointers and sizes (device pointers are raw `void*` from
 * `torch.Tensor.data_ptr()`), returns 0 on success and a negative code on error
 * (`bg_last_error()` returns a thread-local message).  No function synchronises
 * the device or allocates after create; all work is enqueued on the caller's
 * `stream` (a hipStream_t passed as void*, NULL = default stream).
 *
 * Citations `file:line` refer to the reference tree.
 */
#ifndef BOOSTER_GYM_AMD_H
#define BOOSTER_GYM_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BG_NUM_BODIES 13
#define BG_NUM_DOFS 12
#define BG_NUM_OBS 47
#define BG_NUM_PRIV 14
#define BG_NUM_REWARD_TERMS 26
#define BG_MAX_BODY_SPHERES 16

typedef struct bg_model bg_model;
typedef struct bg_env bg_env;

/* Flat articulated-body model (what Isaac Gym's load_asset + asset queries return:
 * envs/t1.py:54-67, 85-108).  Bodies depth-first: trunk, left leg 1..6, right leg 7..12. */
typedef struct {
    int32_t num_bodies, num_dofs;
    int32_t parent[BG_NUM_BODIES];
    int32_t joint_axis[BG_NUM_BODIES]; /* 0 floating base, 1/2/3 revolute about x/y/z */
    float body_pos[BG_NUM_BODIES][3];
    float mass[BG_NUM_BODIES];
    float com[BG_NUM_BODIES][3];
    float inertia[BG_NUM_BODIES][6]; /* about com: xx yy zz xy xz yz */
    float dof_lower[BG_NUM_DOFS], dof_upper[BG_NUM_DOFS], dof_velocity[BG_NUM_DOFS], dof_effort[BG_NUM_DOFS];
    float feet_edge_pos[4][3]; /* cfg asset.feet_edge_pos, envs/T1.yaml:79-82 */
    /* contact spheres of the NON-foot collision shapes of the asset (URDF <collision>: box corners with radius 0, two spheres inscribed in the
     * ends of a cylinder), sorted by body; bodies are the trunk or leg links.  num_body_spheres = 0 switches these contacts off. */
    int32_t num_body_spheres;
    int32_t sphere_body[BG_MAX_BODY_SPHERES];
    float sphere_pos[BG_MAX_BODY_SPHERES][3];
    float sphere_radius[BG_MAX_BODY_SPHERES];
    /* Self-collision capsules (gym.create_actor(..., self_collisions), envs/t1.py:128; asset.self_collisions, envs/T1.yaml:69): [leg][0] the shank
     * (4th link of the leg chain; the capsule inscribed in its URDF cylinder, axis z), [leg][1] the foot (last link; a capsule along x standing in
     * for the URDF box).  Segment end points a / b in link coordinates and the radius; radius 0 on any of them = the model has no self-collision
     * geometry.  Left-leg capsules meet right-leg capsules; links of one chain never meet each other. */
    float self_capsule_a[2][2][3], self_capsule_b[2][2][3], self_capsule_r[2][2];
} bg_model_desc;

/* randomisation / noise entry (utils/utils.py:5-30): mode 0 none, 1 gaussian additive,
 * 2 gaussian scaling, 3 uniform additive, 4 uniform scaling; (a,b) = the yaml `range`. */
typedef struct { int32_t mode; float a, b; } bg_rand;

/* order of reward terms = order of envs/T1.yaml:252-278; scale 0 drops the term (t1.py:280-285) */
enum {
    BG_REW_SURVIVAL = 0, BG_REW_TRACKING_LIN_VEL_X, BG_REW_TRACKING_LIN_VEL_Y, BG_REW_TRACKING_ANG_VEL, BG_REW_BASE_HEIGHT,
    BG_REW_ORIENTATION, BG_REW_TORQUES, BG_REW_TORQUE_TIREDNESS, BG_REW_POWER, BG_REW_LIN_VEL_Z, BG_REW_ANG_VEL_XY, BG_REW_DOF_VEL,
    BG_REW_DOF_ACC, BG_REW_ROOT_ACC, BG_REW_ACTION_RATE, BG_REW_DOF_POS_LIMITS, BG_REW_DOF_VEL_LIMITS, BG_REW_TORQUE_LIMITS,
    BG_REW_COLLISION, BG_REW_FEET_SLIP, BG_REW_FEET_VEL_Z, BG_REW_FEET_YAW_DIFF, BG_REW_FEET_YAW_MEAN, BG_REW_FEET_ROLL,
    BG_REW_FEET_DISTANCE, BG_REW_FEET_SWING
};

/* Everything numeric the env needs from envs/T1.yaml (sections sim, control, normalization,
 * noise, randomization, commands, rewards, init_state, terrain) plus this build's contact model. */
typedef struct {
    int32_t num_envs;
    int32_t device;     /* HIP device ordinal; there is no CPU path in this library */
    uint64_t seed;
    /* sim (T1.yaml:39-44) + contact model of this build */
    float sim_dt;
    int32_t decimation; /* control.decimation, T1.yaml:95 */
    float gravity[3];
    float contact_k, contact_d, contact_ramp, friction_visc, limit_k, limit_d;
    float terrain_mu, terrain_restitution; /* T1.yaml:99-101 */
    int32_t clamp_qd;
    /* control / normalization (T1.yaml:91-95, 135-145) */
    float action_scale, clip_actions;
    float norm_gravity, norm_lin_vel, norm_ang_vel, norm_dof_pos, norm_dof_vel, filter_weight, norm_push_force, norm_push_torque;
    float default_dof_pos[BG_NUM_DOFS]; /* t1.py:264-272 */
    float base_init_state[13];          /* t1.py:109-112 */
    /* noise (T1.yaml:147-171) */
    bg_rand noise_gravity, noise_lin_vel, noise_ang_vel, noise_dof_pos, noise_dof_vel, noise_height;
    /* run-time randomisation (T1.yaml:173-206) */
    bg_rand init_dof_pos, init_base_pos_xy, init_base_lin_vel_xy, kick_lin_vel, kick_ang_vel, push_force, push_torque;
    int32_t kick_interval, push_interval, push_duration; /* in env steps: ceil(seconds / dt), t1.py:501,508,517 */
    int32_t shared_reset_noise; /* 1 = reference quirk Q1 (t1.py:320): one noise vector per reset call */
    /* commands (T1.yaml:115-133) */
    float cmd_lin_vel_x[2], cmd_lin_vel_y[2], cmd_ang_vel_yaw[2], cmd_gait_frequency[2];
    float still_proportion;
    int32_t resample_steps[2]; /* int(seconds / dt), t1.py:384-385 */
    /* command curriculum (T1.yaml:124-133, t1.py:391-435); levels are +-lin_vel_levels x +-ang_vel_levels */
    int32_t curriculum;
    int32_t lin_vel_levels, ang_vel_levels;
    float curriculum_update_rate, lin_vel_x_resolution, lin_vel_y_resolution, ang_vel_resolution;
    float episode_length_toler, lin_vel_x_toler, lin_vel_y_toler, ang_vel_yaw_toler;
    /* rewards (T1.yaml:251-291) */
    float reward_scale[BG_NUM_REWARD_TERMS]; /* yaml value * dt; 0 = dropped */
    int32_t only_positive_rewards;
    float tracking_sigma, base_height_target, soft_dof_pos_limit, soft_dof_vel_limit, soft_torque_limit, swing_period, feet_distance_ref;
    int32_t max_episode_length; /* ceil(episode_length_s / dt), t1.py:556 */
    float terminate_height, terminate_vel;
    /* terrain extents for the teleport wrap (t1.py:343-360); plane => teleport off */
    int32_t terrain_type; /* 0 plane, 1 heightfield ("trimesh" in the yaml) */
    float terrain_env_width, terrain_env_length, terrain_border;
    /* 1 = keep the per-env dynamic state in fp16 between env steps (BASELINE.json configs[4]; T1.yaml sim.state_dtype: fp16): orientation,
     * velocities, joint state, targets / actions and their histories, commands, gait, filters, push wrench.  Arithmetic stays fp32; root position,
     * last feet positions, parameters and outputs stay fp32.  bg_env_get/set_state / get/set_field convert transparently. */
    int32_t state_fp16;
    /* non-foot body contacts: evaluated for envs whose trunk origin starts the step lower than body_gate_height above
     * the terrain, and only if body_gate_height > terminate_height (otherwise every such env has been reset before its next step and the
     * second launch that handles them is left out altogether).
     * Bit b of the masks = body b: rewards.penalize_contacts_on / terminate_contacts_on resolved against the body names (t1.py:85-100);
     * a body counts when its net contact force exceeds 1 N (t1.py:553,629). */
    float body_gate_height;
    int32_t penalized_body_mask, terminate_body_mask;
    /* Reference-exact command resampling (T1.yaml parallel.exact_still_count / parallel.same_step_curriculum; default 0 = this build's
     * single-launch forms).  exact_still_count: exactly int(still_proportion * K) of the K envs that resample in a step stand still, a uniformly
     * random subset (t1.py:381-383 randperm prefix) instead of a per-env Bernoulli draw.  same_step_curriculum: the curriculum sampler reads the
     * grid AFTER this step's resets have updated it (t1.py:305 before :365) instead of the grid as of the start of the step.  Either one moves
     * the cross-env part of _resample_commands into two small follow-up launches per env step (count, apply); not available with state_fp16. */
    int32_t exact_still_count, same_step_curriculum;
    /* Leg-against-leg contacts (asset.self_collisions: 0 in envs/T1.yaml:69 is Isaac Gym's "collide with everything", passed to create_actor at
     * envs/t1.py:128): 1 = the shank / foot capsules of the two legs repel each other with an explicit penalty force:
     * normal stiffness [N/m] and damping [N s/m], Coulomb coefficient, and the viscous cap [N s/m] of the regularised friction. */
    int32_t self_collisions;
    float self_k, self_d, self_mu, self_visc;
} bg_env_cfg;

/* ---- model (replaces gym.load_asset and the asset queries, t1.py:54-108) */
int bg_model_create(const bg_model_desc* desc, bg_model** out);
/* gym.load_asset(sim, root, file, asset_options) with collapse_fixed_joints (envs/t1.py:39-54, envs/T1.yaml:61-83) for hosts without the Python
 * loader: parses the URDF at `path`, folds links behind fixed joints into their parents (masses, centres of mass, inertias, collision
 * primitives), orders bodies / DoFs depth-first like Isaac Gym and fills the flat model, including the contact spheres of the non-foot
 * collision primitives.  foot_names = cfg asset.foot_names (their boxes are the sole-corner contacts, not spheres), feet_edge_pos = cfg
 * asset.feet_edge_pos.  The same algorithm as booster_gym_amd/utils/urdf.py (tests/test_host_logic.py compares the two on the same files). */
typedef struct {
    int32_t collapse_fixed_joints;  /* envs/T1.yaml:67 */
    int32_t body_contacts;          /* 0: no contact spheres (feet-only contact) */
    int32_t self_collisions;        /* 0: no self-collision capsules */
    const char* foot_names[2];      /* envs/T1.yaml:78 */
    float feet_edge_pos[4][3];      /* envs/T1.yaml:79-82 */
} bg_asset_options;
int bg_model_load_urdf(const char* path, const bg_asset_options* options, bg_model** out);
/* gym.get_asset_rigid_body_names / get_asset_dof_names / find_asset_rigid_body_index (t1.py:57,85,92-106); NULL / -1 when out of range or
 * when the model was made by bg_model_create (no names).  Counts, limits and efforts (t1.py:55-67): bg_model_get. */
const char* bg_model_body_name(const bg_model* m, int32_t i);
const char* bg_model_dof_name(const bg_model* m, int32_t j);
int32_t bg_model_find_body(const bg_model* m, const char* name);
int bg_model_get(const bg_model* m, bg_model_desc* out);
void bg_model_destroy(bg_model* m);

/* ---- env = simulator + T1 task logic (replaces create_sim/create_env/create_actor/prepare_sim,
 *      base_task.py:20-79, t1.py:33-137, and owns the per-env state of t1.py:187-272) */
int bg_env_create(const bg_env_cfg* cfg, const bg_model* model, bg_env** out);
void bg_env_destroy(bg_env* env);
/* height field for terrain.type "trimesh" (utils/terrain.py:30-99): host int16 [rows][cols], copied to the device */
int bg_env_set_heightfield(bg_env* env, const int16_t* hf_host, int32_t rows, int32_t cols, int32_t border_px, float hscale, float vscale);
/* per-env build-time randomisation (t1.py:69-83, 122-167), host float arrays, env-major:
 * kp/kd/friction [N][12], mass_scale [N][13], com_offset [N][13][3], foot_material [N][2][3] = (friction, compliance,
 * restitution), base_mass_scaled [N][4] (raw draws shown to the critic, t1.py:141-152), env_origins [N][3] (t1.py:169-185) */
int bg_env_set_params(bg_env* env, const float* kp, const float* kd, const float* friction, const float* mass_scale, const float* com_offset,
                      const float* foot_material, const float* base_mass_scaled, const float* env_origins);
/* outputs of reset/step, device pointers owned by the caller (torch tensors): obs [N][47], privileged [N][14],
 * rew [N], done uint8 [N], time_outs uint8 [N], rew_terms [26][N] (rows of dropped terms stay 0) */
int bg_env_bind_outputs(bg_env* env, float* obs, float* privileged_obs, float* rew, uint8_t* done, uint8_t* time_outs, float* rew_terms);
/* T1.reset(): t1.py:294-299 */
int bg_env_reset(bg_env* env, void* stream);
/* T1.step(actions): t1.py:437-497.  actions: device float [N][12] */
int bg_env_step(bg_env* env, const float* actions, void* stream);
/* Same, writing obs/privileged/rew/done/time_outs of this step to the given device pointers instead of
 * the bound ones (lets the rollout loop write straight into rows of the experience buffer, runner.py:107-118) */
int bg_env_step_to(bg_env* env, const float* actions, float* obs, float* privileged_obs, float* rew, uint8_t* done, uint8_t* time_outs,
                   void* stream);
/* Isaac-Gym-layout views of the simulator state for inspection / tests (device pointers, may be NULL):
 * root [N][13] (pos, quat xyzw, lin vel, ang vel: t1.py:215), dof [N][12][2] (pos, vel: t1.py:216-218),
 * contact [N][13][3] net contact force per body, world frame (t1.py:219) */
int bg_env_get_state(bg_env* env, float* root, float* dof, float* contact, void* stream);
int bg_env_set_state(bg_env* env, const float* root, const float* dof, void* stream);
/* named per-env arrays (task state of t1.py:187-272), float or int32 depending on the field; see bg_env_field_info */
int bg_env_get_field(bg_env* env, const char* name, void* dst_device, void* stream);
int bg_env_set_field(bg_env* env, const char* name, const void* src_device, void* stream);
int bg_env_field_info(bg_env* env, const char* name, int32_t* components, int32_t* is_int);
/* curriculum_prob grid [(2*lin_vel_levels+1)][(2*ang_vel_levels+1)] (t1.py:249-255), device float pointers.  The library keeps the
 * UNCLAMPED running sums (increments are positive, so min(sum, 1) equals the reference's clamp_(max=1) after every update);
 * get returns min(sum, 1). */
int bg_env_get_curriculum(bg_env* env, float* prob_device, void* stream);
int bg_env_set_curriculum(bg_env* env, const float* prob_device, void* stream);
int64_t bg_env_step_count(const bg_env* env);
int bg_env_set_step_count(bg_env* env, int64_t count);

/* ---- dynamics only (what north_star calls "per-step joint accelerations"): forward dynamics of N
 * independent states.  Device float arrays, env-major: root [N][13], dof_pos/dof_vel/tau [N][12],
 * base_wrench [N][6] (force, torque in base coords) or NULL; qacc out [N][18] = d/dt(lin vel world,
 * ang vel world, dof vel).  Uses the env's model, per-env parameters and terrain. */
int bg_env_forward_dynamics(bg_env* env, const float* root, const float* dof_pos, const float* dof_vel, const float* tau,
                            const float* base_wrench, float* qacc, void* stream);
/* The same call through the PACKED kernel: one env per wavefront lane, its two legs in the halves of 64-bit register pairs (v_pk_fma_f32 ...),
 * instead of one leg per lane.  Same inputs, same outputs to fp32 rounding; returns -4 when the env was created with the trunk-low gate
 * (body_gate_height > terminate_height: the non-foot body contacts need the two-kernel launch).  Replaces the same reference line
 * (envs/t1.py:451 gym.simulate, dynamics only). */
int bg_env_forward_dynamics_packed(bg_env* env, const float* root, const float* dof_pos, const float* dof_vel, const float* tau,
                            const float* base_wrench, float* qacc, void* stream);

/* ---- granular simulator calls: the Isaac Gym tensor API of envs/t1.py one call at a time (lower seam).
 * A maintainer who keeps the reference's Python task logic replaces each gym.* call by the entry point named here; the fused
 * bg_env_step above is the same physics + task logic in one launch.  The tensors are CALLER-OWNED device memory in the Isaac Gym
 * layouts and are the simulator state for these calls (no hidden copy): root [N][13] = pos, quat xyzw, lin vel, ang vel (world)
 * (t1.py:215,221-222); dof [N][12][2] = (pos, vel) (t1.py:216-218); contact [N][13][3] net contact force per body, world frame
 * (t1.py:219; only the feet carry collision geometry in this build, other rows are 0); body [N][13][13] = origin pos, quat xyzw
 * (w >= 0), lin vel of the origin, ang vel, world frame (t1.py:220).  contact / body may be NULL.  Per-env parameters, model and
 * terrain are those of the env (bg_env_set_params / bg_env_set_heightfield).  These calls do not touch the env's own task state;
 * bg_env_get_state / bg_env_set_state move state between the two. */
/* gym.acquire_actor_root_state_tensor / acquire_dof_state_tensor / acquire_net_contact_force_tensor / acquire_rigid_body_state_tensor
 * (t1.py:203-220).  Allocates the library's actuation / applied-force copies on first use. */
int bg_sim_bind_state(bg_env* env, float* root, float* dof, float* contact, float* body);
/* gym.set_dof_actuation_force_tensor (t1.py:450): tau [N][12], copied; stays in force until set again */
int bg_sim_set_actuation(bg_env* env, const float* tau, void* stream);
/* gym.apply_rigid_body_force_tensors(sim, forces, torques, LOCAL_SPACE) (t1.py:522-527): force / torque [N][13][3] in each body's own
 * frame, the force acting at the body's centre of mass; copied; either may be NULL (= zero); consumed by the next bg_sim_simulate only */
int bg_sim_apply_body_wrench_local(bg_env* env, const float* force, const float* torque, void* stream);
/* gym.simulate (t1.py:451): one sim_dt step of all envs, in place on root / dof; writes contact and body.  The refresh_* calls
 * that follow in the reference (t1.py:452-455, 460-462) have nothing left to do. */
int bg_sim_simulate(bg_env* env, void* stream);
/* gym.refresh_rigid_body_state_tensor without a step (t1.py:462 after a reset): body rows from the current root / dof tensors */
int bg_sim_refresh_body_state(bg_env* env, void* stream);
/* gym.set_actor_root_state_tensor_indexed (t1.py:341,359,504) / gym.set_dof_state_tensor_indexed (t1.py:323-325): the bound tensors
 * already are the state, so these only re-derive the body rows; env_ids (device int32 [count]) is accepted for call-site parity */
int bg_sim_write_root_state(bg_env* env, const int32_t* env_ids, int32_t count, void* stream);
int bg_sim_write_dof_state(bg_env* env, const int32_t* env_ids, int32_t count, void* stream);

/* ---- PPO math (utils/utils.py:33-52, utils/runner.py:123-180), all pointers device float unless noted */
/* GAE + returns + advantage moments.  rewards [T][N] is modified in place where time_outs is set (runner.py:135).
 * sums out [3] = (sum adv, sum adv^2, count) as float64, accumulated with atomics; caller zeroes it. */
int bg_gae(int32_t T, int32_t N, float* rewards, const uint8_t* dones, const uint8_t* time_outs, const float* values,
           const float* last_values, float gamma, float lam, float* advantages, float* returns, double* sums, void* stream);
/* Fused PPO loss forward+backward over B samples with A actions (runner.py:144-174):
 * in:  mu [B][A], logstd [A], actions [B][A], old_mu [B][A], old_logstd [A], old_logp [B], adv [B] (un-normalised),
 *      adv_stats [3] (sum, sumsq, count), values [B], returns [B]
 * out: grad_mu [B][A], grad_values [B], grad_logstd [A] float64 (atomic, caller zeroes), stats [5] float64 (atomic,
 *      caller zeroes) = sums over samples of (value error^2, surrogate, bound penalty, entropy, kl).
 * adv_stats [3] float64 = bg_gae's sums (all-reduced across ranks in data-parallel runs). */
int bg_ppo_loss(int32_t B, int32_t A, const float* mu, const float* logstd, const float* actions, const float* old_mu,
                const float* old_logstd, const float* old_logp, const float* adv, const double* adv_stats, const float* values,
                const float* returns, float e_clip, float bound_coef, float entropy_coef, float* grad_mu, float* grad_values,
                double* grad_logstd, double* stats, void* stream);
/* log-prob of actions under N(mu, exp(logstd)) summed over A (runner.py:123-125) */
int bg_gaussian_logp(int32_t B, int32_t A, const float* mu, const float* logstd, const float* actions, float* logp, void* stream);
/* Fused actor inference for the rollout (utils/model.py:29-32 + dist.sample(), runner.py:109-111):
 * 47->256->128->128->12 ELU MLP + Gaussian sample.  weights: w0[256][47] b0[256] w1[128][256] b1 w2[128][128] b2 w3[12][128] b3,
 * logstd[12]; obs [N][47]; out: mu [N][12] (may be NULL), actions [N][12].  Noise from Philox(seed, counter). */
int bg_actor_sample(int32_t N, const float* obs, const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                    const float* b2, const float* w3, const float* b3, const float* logstd, uint64_t seed, uint64_t counter,
                    float* mu, float* actions, void* stream);
/* Fused global-norm clip + Adam over one flat parameter buffer (runner.py:162-165); lr is read from device memory
 * so the KL-adaptive schedule (runner.py:174-180) needs no host sync.  gnorm_scratch [1] device float64. */
int bg_adam_step(int32_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* lr_device, int32_t step,
                 float beta1, float beta2, float eps, float max_grad_norm, double* gnorm_scratch, void* stream);
/* KL-adaptive learning rate on the device (runner.py:174-180): kl_sum [1] float64 (= stats[4] of bg_ppo_loss), count = samples -> lr_device [1] updated in place */
int bg_adapt_lr(const double* kl_sum, float count, float desired_kl, float lr_min, float lr_max, float* lr_device, void* stream);
/* A second copy of one [rows][cols] row-major weight matrix of the flat parameter buffer that bg_optimizer_step keeps current while it updates the
 * parameters: transpose = 0: dst[r * ld + c] (ld >= cols: e.g. the first layer with its input columns zero-padded; the padding is not touched),
 * transpose = 1: dst[c * ld + r] (ld >= rows: the operand layout of bg_mlp_layer_backward).  offset = index of W[0][0] in params.
 * transpose = 2 / 3: dst = the bf16 planes of W / of W^T as bg_mlp_split_weights(transpose = 0 / 1) writes them (uint16_t [n_out][ld / 32][3][32],
 * 16-byte aligned, ld = k_out a multiple of 32 and >= cols / rows): what the chained split kernels read.  Elements outside the matrix (the
 * zero-padded input columns of a first layer) are not touched: write them once with bg_mlp_split_weights.  pad > 0 (kinds 2 / 3): the planes of
 * -W are kept as well, pad uint16 behind dst (bg_mlp_split_weights_pm's layout: pad = n_out * k_out * 3). */
typedef struct bg_param_mirror {
    int32_t offset, rows, cols, transpose, ld, pad;
    float* dst;
} bg_param_mirror;
/* The tail of a mini-epoch in one launch (runner.py:162-180): global-norm clip + Adam on the flat buffers (as bg_adam_step), then the KL rule on
 * lr_device (as bg_adapt_lr, with kl_sum = stats[kl_index]), and the bookkeeping of the float64 loss statistics: stats_last = stats,
 * stats_acc += stats, stats = 0 (and grad_logstd = 0) for the next mini-epoch.  grad_logstd (optional, float64 [ls_n]) is the log-std gradient as
 * the head kernels accumulate it; it is written into grads[ls_off .. ls_off + ls_n) first.  stats may be NULL (no learning-rate rule, no
 * bookkeeping).  ticket: one zero-initialised uint32 of device memory owned by the caller.  mirrors (optional, up to 16): see bg_param_mirror.
 * Deterministic (no float atomics). */
int bg_optimizer_step(int32_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* lr_device, int32_t step, float beta1, float beta2,
                      float eps, float max_grad_norm, double* grad_logstd, int32_t ls_off, int32_t ls_n, double* stats, double* stats_acc,
                      double* stats_last, int32_t n_stats, int32_t kl_index, float kl_count, float desired_kl, float lr_min, float lr_max,
                      uint32_t* ticket, const bg_param_mirror* mirrors, int32_t n_mirrors, void* stream);

/* MLP backward helper for the ELU layers of utils/model.py:9-26: grad [B][C] <- grad * elu'(.) in place, expressed through the layer OUTPUT
 * act [B][C] (1 if act > 0 else act + 1; act == NULL: identity), and colsum [C] = column sums of the result (= bias gradient).
 * scratch: ceil(B/128) * C floats.  Deterministic (no atomics). */
int bg_elu_backward_colsum(int32_t B, int32_t C, float* grad, const float* act, float* colsum, float* scratch, void* stream);

/* Fused MLP layer forward: Y [M][N] = act(X [M][K] . W[N][K]^T + bias[N]) in fp32 MFMA with bias + ELU in the GEMM epilogue
 * (utils/model.py:9-26 Linear + ELU).  Supported: K in {64, 128, 256}, N a multiple of 128; other shapes return -4 (caller uses the library GEMM). */
int bg_mlp_layer_forward(int32_t M, int32_t K, int32_t N, const float* X, const float* W, const float* bias, float* Y, int32_t elu, void* stream);
/* The three hidden Linear + ELU layers of one network (utils/model.py:9-26, forward of runner.py:132,147) in ONE launch: Y1 = elu(X W1^T + b1) [M][N1],
 * Y2 = elu(Y1 W2^T + b2) [M][N2], Y3 = elu(Y2 W3^T + b3) [M][N3], every activation stored for the backward pass; between the layers the
 * activations stay in registers (bg_mlp_chain.hip).  X [M][K0] with K0 = 64 (zero-padded input columns), W row-major [out][in] as torch's Linear.
 * Supported widths (N1, N2, N3): (256, 128, 128) and (256, 256, 128), the reference's actor and critic.  Y1 / Y2 / Y3 must hold
 * ceil(M / 128) * 128 rows (rows >= M of the last slab are written with unspecified values).  Bit-identical to three bg_mlp_layer_forward
 * launches.  -4: unsupported widths. */
typedef struct bg_mlp_chain {
    int32_t M, K0, N1, N2, N3;
    int32_t workgroups;  /* 0: one workgroup per 128-row slab; > 0: that many workgroups walk the slabs (each fills a CU: two launches side by side then
                          * share the chip by CUs -- see bg_mlp_chain.hip).  Same outputs either way. */
    const float *X, *W1, *b1, *W2, *b2, *W3, *b3;
    float *Y1, *Y2, *Y3;
    /* optional scalar output layer on Y3 (the critic's value head, utils/model.py:21): v_out[row] = v_w . Y3[row] + v_b[0], taken from the registers that
     * hold Y3 (v_w [N3], v_b [1]; v_out [M]); all three NULL: none.  Another summation order than bg_critic_head_forward (same values to fp32 rounding). */
    const float *v_w, *v_b;
    float* v_out;
} bg_mlp_chain;
int bg_mlp_chain_forward(int32_t M, int32_t K0, int32_t N1, int32_t N2, int32_t N3, const float* X, const float* W1, const float* b1, const float* W2,
                         const float* b2, const float* W3, const float* b3, float* Y1, float* Y2, float* Y3, void* stream);
/* 1 to 4 networks in one launch; the 128-row slabs of nets[0] are dispatched first, those of the next network fill the machine as they retire. */
int bg_mlp_chain_forward_group(const bg_mlp_chain* nets, int32_t count, void* stream);

/* Fused MLP layer backward through one Linear and the ELU below it:  Gout [M][N] = (G [M][K] . Wt[N][K]^T) * elu'(act_below [M][N]) and
 * bias_grad_below [N] = column sums of Gout.  G = dL/dz of the upper layer (K = its width), Wt = that layer's weight TRANSPOSED to [N][K]
 * (N = width of the layer below), act_below = the lower layer's output activations.  scratch: ceil(M/128) * N floats.
 * Replaces torch.mm(G, W) + elu_backward + the bias-gradient reduction.  K in {128, 256}, N a multiple of 128; otherwise -4. */
int bg_mlp_layer_backward(int32_t M, int32_t K, int32_t N, const float* G, const float* Wt, const float* act_below, float* Gout,
                          float* bias_grad_below, float* scratch, void* stream);

/* Split form of the two layer kernels above (opt-in; same inputs, outputs and epilogues): the fp32 x fp32 products run on the bf16 matrix
 * pipe.  Every fp32 operand is EXACTLY the sum of three bf16 numbers (hi / mid / lo: 8 + 8 + 8 significant bits) and a bf16 x bf16 product
 * is exact in the fp32 accumulator, so terms = 9 (all cross terms) accumulates the exact products x * w in fp32 -- no precision is given up
 * against the fp32 MFMA form; terms = 6 drops mid*lo, lo*mid, lo*lo (<= 2^-23 of each product).  The weights come pre-split:
 * bg_mlp_split_weights writes planes[n_out][k_out / 32][3][32] bf16 from W [src_rows][ldw] (transpose != 0: element (n, k) = W[k][n]);
 * elements outside the source are zero (the zero-padded first layers), k_out a multiple of 32, 6 bytes per element.
 * Same shape limits and error codes as bg_mlp_layer_forward / _backward. */
int bg_mlp_split_weights(int32_t n_out, int32_t k_out, const float* W, int32_t ldw, int32_t src_rows, int32_t src_cols, int32_t transpose,
                         uint16_t* planes, void* stream);
/* bg_mlp_split_weights, and behind its planes (n_out * k_out * 3 uint16 further) the planes of -W: what the chained split kernels read with
 * `alternate` set (planes must hold 2 * n_out * k_out * 3 uint16). */
int bg_mlp_split_weights_pm(int32_t n_out, int32_t k_out, const float* W, int32_t ldw, int32_t src_rows, int32_t src_cols, int32_t transpose,
                            uint16_t* planes, void* stream);
int bg_mlp_layer_forward_split(int32_t M, int32_t K, int32_t N, const float* X, const uint16_t* planes, const float* bias, float* Y, int32_t elu,
                               int32_t terms, void* stream);
int bg_mlp_layer_backward_split(int32_t M, int32_t K, int32_t N, const float* G, const uint16_t* planes_t, const float* act_below, float* Gout,
                                float* bias_grad_below, float* scratch, int32_t terms, void* stream);

/* bg_mlp_chain_forward_group on the bf16 matrix pipe with fp32 semantics (bg_mlp_chain_split.hip; reference utils/model.py:9-26 under
 * utils/runner.py:132,147): the same three Linear + ELU layers per network in one launch, activations handed on in registers, every fp32 x fp32 product
 * formed exactly from the 9 cross products of the operands' three bf16 planes (see bg_mlp_layer_forward_split; terms = 9 only).  P1 / P2 / P3: the
 * layers' weight planes as bg_mlp_split_weights writes them (transpose = 0; P1 with k_out = K0 = 64, the zero-padded input width).  Same shapes, slab
 * padding of Y1 / Y2 / Y3, `workgroups` and value head as bg_mlp_chain.  Not bit-identical to the fp32-MFMA chain (another summation order; the bias is
 * the accumulators' initial value): both are exact-product fp32 sums.  1 to 4 networks per launch (each on its own `workgroups` persistent
 * workgroups inside one grid: how the update shares the chip between the critic and the actor); which workgroup walks which slab changes no bit of
 * any output (tests/test_gpu_mlp_chain_split.py). */
typedef struct bg_mlp_chain_split {
    int32_t M, K0, N1, N2, N3;
    int32_t workgroups;
    /* alternate != 0: odd slabs accumulate the NEGATED sums (the planes of -W: P1 / P2 / P3 then hold both sets, bg_mlp_split_weights_pm) and the
     * sign is put back where a tile is finished.  The bf16 MFMA's accumulator does not round to nearest: every accumulated element carries a small
     * bias of one sign (measured: -0.05 of the rms error), which adds up in everything summed over rows downstream (bias and weight gradients);
     * alternating the sign of the accumulation slab by slab makes the bias cancel.  Same products, same exactness. */
    int32_t alternate, pad;
    const float* X;
    const uint16_t *P1, *P2, *P3;
    const float *b1, *b2, *b3;
    float *Y1, *Y2, *Y3;
    const float *v_w, *v_b;
    float* v_out;
} bg_mlp_chain_split;
int bg_mlp_chain_forward_split(const bg_mlp_chain_split* nets, int32_t count, void* stream);

/* Weight gradient of one Linear layer over the batch (the dW part of `loss.backward()`, utils/runner.py:163, for model.py:9-26's layers):
 * dW [C_out][C_in_real] = G [M][C_out]^T . A [M][C_in][:, :C_in_real], fp32 MFMA, the sum over the M rows split over `slices` x 4 waves
 * inside the launch and finished in a fixed order (deterministic, no atomics).  G = dL/dz of the layer, A = its input activations with
 * the feature dimension C_in possibly zero-padded (the first layers: 47 / 61 -> 64); only the first C_in_real columns are written, with
 * row stride C_in_real, so dW can be the parameter's .grad view in the flat gradient buffer.
 * C_out a multiple of 128, C_in 64 or a multiple of 128, M even, slices a multiple of 8 with slices * 8 <= M;
 * scratch: slices * C_out * C_in floats.  Otherwise -4 / -1. */
int bg_mlp_weight_grad(int32_t M, int32_t C_out, int32_t C_in, int32_t C_in_real, const float* G, const float* A, float* dW, float* scratch,
                       int32_t slices, void* stream);

/* The weight gradients of several layers in ONE launch pair (the dW part of `loss.backward()` for all hidden layers of both networks, after both
 * backward chains): same arithmetic and argument meaning per layer as bg_mlp_weight_grad; `slices` of a layer may be any value in [1, M/8] -- size
 * them so that every workgroup of the launch does the same amount of work (booster_gym_amd/utils/model.py:plan_wgrad_slices).  One workgroup per
 * (group of tiles_per_workgroup tiles, slice).  count <= 8. */
typedef struct {
    const float* G;      /* [M][C_out] dL/dz of the layer */
    const float* A;      /* [M][C_in] its input activations (feature dimension possibly zero-padded) */
    float* dW;           /* [C_out][C_in_real] */
    float* scratch;      /* slices * C_out * C_in floats */
    int32_t M, C_out, C_in, C_in_real, slices;
    int32_t tiles_per_workgroup; /* 1 (default for 0), 2 or 4, dividing the layer's count of 128-wide output tiles: the 4 waves of a workgroup cover
                                  * this many tiles x 4 / this many sub-ranges of the slice's rows; waves on the same rows share the fetched rows */
} bg_wgrad_problem;
int bg_mlp_weight_grad_group(const bg_wgrad_problem* problems, int32_t count, void* stream);
/* bg_mlp_weight_grad_group without its finishing launch: the per-slice partial tiles stay in the problems' scratch, to be summed by bg_update_tail. */
int bg_mlp_weight_grad_group_partial(const bg_wgrad_problem* problems, int32_t count, void* stream);
/* Split form of the grouped launch (the training loop's default since round 6, terms = 9; reference utils/runner.py:163): the same sums with every fp32
 * operand split exactly into three bf16 numbers on the bf16 matrix pipe, terms = 9 (every product exact) or 6.  The sub-ranges of the batch take turns
 * accumulating the NEGATED sums (the bf16 MFMA's accumulator truncates; the offsets then cancel in the sum over the batch): measured error against
 * float64 0.84-0.89 of bg_mlp_weight_grad_group's.  The waves of a workgroup that work on the same rows share them through LDS, so here
 * tiles_per_workgroup must equal the layer's tile count (1, 2 or 4).  Shapes: 256 x 256, 128 x 256, 128 x 128, 256 x 64 (C_out x C_in padded), M a
 * multiple of 32; anything else returns -4 and the caller uses bg_mlp_weight_grad_group.  Same scratch layout and fixed-order finish. */
int bg_mlp_weight_grad_group_split(const bg_wgrad_problem* problems, int32_t count, int32_t terms, void* stream);
/* ... without its finish, for bg_update_tail (as bg_mlp_weight_grad_group_partial: the same scratch layout, the same finish). */
int bg_mlp_weight_grad_group_split_partial(const bg_wgrad_problem* problems, int32_t count, int32_t terms, void* stream);

/* ---- output ("head") layers fused with the loss: the 128 -> 12 / 128 -> 1 Linear layers of utils/model.py:13,21 together with
 * runner.py:145-174.  h [rows][128] = activations of the last hidden (ELU) layer, 16-byte aligned.  One launch reads h once instead of
 * the seven library GEMM / elementwise passes these skinny layers otherwise take per network and mini-epoch.
 * scratch: BG_HEAD_SCRATCH_FLOATS floats of device memory per concurrent call (per-workgroup partial sums, added in a fixed order). */
#define BG_HEAD_SCRATCH_FLOATS (768 * 1720)
/* critic.6 forward: values [rows] = h w + b   (w [128], b [1]) */
int bg_critic_head_forward(int32_t rows, const float* h, const float* w, const float* b, float* values, void* stream);
/* bg_critic_head_forward on all (T + 1) N rows of h (time-major: row t N + e), then bg_gae on the result, in ONE launch (utils/runner.py:132-141: values,
 * last values, timeout bootstrap, discount_values, returns): values_all [(T + 1) N] (the last N = last_values), advantages / returns [T][N], rewards
 * overwritten at time-outs as bg_gae does, sums float64 [3] WRITTEN (sum, sum of squares, count; fixed order: deterministic).  scratch: float64
 * [3 * ceil(N / 16) + 1], its last element zero-initialised once by the caller (the launch leaves it zero).  Values bit-identical to
 * bg_critic_head_forward, advantages to bg_gae.  h == NULL: values_all is an INPUT (e.g. written by bg_mlp_chain_forward's value head) and only the
 * GAE half runs (w, b unused).  -4: T > 32. */
int bg_critic_values_gae(int32_t T, int32_t N, const float* h, const float* w, const float* b, float* rewards, const uint8_t* dones,
                         const uint8_t* time_outs, float gamma, float lam, float* values_all, float* advantages, float* returns, double* sums,
                         double* scratch, void* stream);
/* actor.6 forward (mode 0: mu_out [B][12] = h W^T + b, nothing else is touched) or forward + PPO actor loss + backward (mode 1), with the
 * argument meaning of bg_ppo_loss: out g_hidden [B][128] = dL/dz of the last hidden layer, grad_W [12][128], grad_b [12],
 * grad_b_hidden [128] (bias gradient of the last hidden layer = column sums of g_hidden), grad_logstd [12] float64 and
 * stats[1..4] float64 (surrogate, bound penalty, entropy, kl sums; atomic, caller zeroes; stats[0] is left to the critic head).
 * mu_out may be NULL in mode 1. */
int bg_actor_head(int32_t B, int32_t mode, const float* h, const float* W, const float* bias, const float* logstd, const float* actions,
                  const float* old_mu, const float* old_logstd, const float* old_logp, const float* adv, const double* adv_stats, float e_clip,
                  float bound_coef, float entropy_coef, float* mu_out, float* g_hidden, float* grad_W, float* grad_b, float* grad_b_hidden,
                  double* grad_logstd, double* stats, float* scratch, void* stream);
/* critic.6 backward of the value loss mean((v - ret)^2) (runner.py:148): g_hidden [B][128], grad_w [128], grad_b [1], grad_b_hidden [128],
 * stats[0] += sum of squared value errors (float64, atomic) */
int bg_critic_head_backward(int32_t B, const float* h, const float* w, const float* values, const float* returns, float* g_hidden, float* grad_w,
                            float* grad_b, float* grad_b_hidden, double* stats, float* scratch, void* stream);

/* ---- deferred fixed-order reductions.  The head kernels and the backward layer kernel leave per-workgroup partial sums that a small second
 * kernel adds up (head: output-layer weight / bias gradients, last hidden layer's bias gradient, float64 loss statistics; backward layer: the
 * bias gradient of the layer below).  None of these sums is needed before the optimiser step (utils/runner.py:162-165), so instead of one small
 * launch in the middle of each network's chain (40 + 80 launches per PPO iteration that wait for workgroup slots between the GEMMs) the
 * `_partial` forms below launch the main kernel only and fill a descriptor; bg_reduce_group then runs all descriptors of a mini-epoch in ONE
 * launch, e.g. on a second stream beside the weight-gradient launch.  Same sums, same fixed order as the immediate forms for the heads; the
 * column sums of the backward layer are added in a different (still fixed) order than bg_mlp_layer_backward's own finish. */
typedef struct {
    const float* partial;           /* [groups][record] floats */
    int32_t groups, record, n_out;  /* out[i] = sum over g of partial[g * record + i], i < n_out */
    float* out[3]; int32_t n[3];    /* element i < n[0] goes to out[0][i], the next n[1] to out[1], the rest to out[2] (unused: NULL / 0) */
    /* optional float64 statistics, statistic-major [n_stat][groups] at (partial + stat_base): statistic k < n_ls is added (+ entropy_coef) to
     * grad_logstd[k], the others to stats[k - n_ls]; bit k of stat_skip skips statistic k */
    uint64_t stat_base; int32_t n_stat, n_ls; uint32_t stat_skip; double entropy_coef; double* grad_logstd; double* stats;
} bg_reduce_problem;
int bg_reduce_group(const bg_reduce_problem* problems, int32_t count, void* stream); /* count <= 8 */
/* The backward-data chain of one network in ONE launch on the bf16 matrix pipe with fp32 semantics (bg_mlp_chain_split_bwd.hip; the dX part of
 * `loss.backward()`, utils/runner.py:163, through utils/model.py:9-26's hidden layers):  G2 [M][N2] = (G3 [M][N3] . W3 [N3][N2]) * elu'(A2),
 * G1 [M][N1] = (G2 . W2 [N2][N1]) * elu'(A1), and the column sums of G2 / G1 (the bias gradients of layers 2 / 1: bg_mlp_layer_backward twice).
 * PT3 / PT2: the planes of W3^T / W2^T as bg_mlp_split_weights(transpose = 1) writes them ([N2][N3 / 32][3][32], [N1][N2 / 32][3][32]); A2 / A1: the
 * layers' stored outputs, holding ceil(M / 128) * 128 rows of finite values as bg_mlp_chain_forward_split leaves them (whole 128-byte rows of a
 * slab are copied; rows >= M do not enter the results).  G2 / G1 must hold ceil(M / 128) * 128 rows (rows >= M are written with zeros).  Every wave
 * leaves one record of column sums per slab in colsum_partial ([ceil(M / 128) * 4][N2 + N1] floats); `finishes[k]` receives the descriptor of the
 * fixed-order reduction that produces bias_grad2 [N2] and bias_grad1 [N1] when handed to bg_reduce_group / bg_update_tail.  Widths (N1, N2, N3):
 * (256, 128, 128), (256, 256, 128).  1 to 4 networks per launch; `workgroups` as in bg_mlp_chain_split: which workgroup walks which slab changes no
 * bit of any output. */
typedef struct bg_mlp_chain_split_bwd {
    int32_t M, N1, N2, N3;
    int32_t workgroups, alternate;   /* alternate: as in bg_mlp_chain_split; PT3 / PT2 then hold the planes of W^T and of -W^T */
    const float* G3;
    const uint16_t *PT3, *PT2;
    const float *A2, *A1;
    float *G2, *G1;
    float* colsum_partial;
    float *bias_grad2, *bias_grad1;
} bg_mlp_chain_split_bwd;
int bg_mlp_chain_backward_split(const bg_mlp_chain_split_bwd* nets, int32_t count, bg_reduce_problem* finishes, void* stream);

/* The tail of a mini-epoch behind bg_mlp_weight_grad_group_partial (runner.py:162-180: the last sums of loss.backward(), clip_grad_norm_,
 * optimizer.step(), the KL rule) as two launches: (1) the weight gradients' fixed-order finish over their slices (wgrad: the problems handed to the
 * _partial launch; 0: none) and the deferred reductions (reduce: as bg_reduce_group; 0: none), every block also leaving the sum of the squares of the
 * gradient values it wrote; (2) everything bg_optimizer_step does (same arguments), with the global norm taken from those sums instead of a pass over
 * the whole gradient per workgroup.  REQUIRES that the two lists together produce every element of grads except the log-std slice (true for the
 * reference's networks: hidden-layer weights from the weight gradients, all biases and the output layers from the reductions); elements they do not
 * write must be zero.  Gradients: the same bits as the separate launches; the norm is summed in another (fixed) order.  sync: uint32 [1],
 * zero-initialised once by the caller (the launch leaves it zero); norm_scratch: float64 [8192]. */
int bg_update_tail(const bg_wgrad_problem* wgrad, int32_t n_wgrad, const bg_reduce_problem* reduce, int32_t n_reduce, int32_t n, float* params, float* grads,
                   float* exp_avg, float* exp_avg_sq, float* lr_device, int32_t step, float beta1, float beta2, float eps, float max_grad_norm,
                   double* grad_logstd, int32_t ls_off, int32_t ls_n, double* stats, double* stats_acc, double* stats_last, int32_t n_stats, int32_t kl_index,
                   float kl_count, float desired_kl, float lr_min, float lr_max, uint32_t* sync, double* norm_scratch, const bg_param_mirror* mirrors,
                   int32_t n_mirrors, void* stream);
/* Launch (1) of bg_update_tail alone, for the ranks of a multi-GPU job: the gradient's last sums, THEN the all-reduce over the ranks, THEN
 * bg_optimizer_step, which takes the norm of the averaged gradient itself (the reference clips what `loss.backward()` left on its one process,
 * runner.py:162-165; under data parallelism that is the mean over the ranks).  norm_scratch: float64 [8192] (written, not used by the caller). */
int bg_update_tail_sums(const bg_wgrad_problem* wgrad, int32_t n_wgrad, const bg_reduce_problem* reduce, int32_t n_reduce, double* norm_scratch, void* stream);
/* bg_actor_head mode 1 / bg_critic_head_backward / bg_mlp_layer_backward without their finishing launch: same arguments, plus the descriptor
 * of the reduction that produces grad_W, grad_b, grad_b_hidden, grad_logstd, stats / bias_grad_below when handed to bg_reduce_group. */
int bg_actor_head_partial(int32_t B, const float* h, const float* W, const float* bias, const float* logstd, const float* actions,
                          const float* old_mu, const float* old_logstd, const float* old_logp, const float* adv, const double* adv_stats, float e_clip,
                          float bound_coef, float entropy_coef, float* mu_out, float* g_hidden, float* grad_W, float* grad_b, float* grad_b_hidden,
                          double* grad_logstd, double* stats, float* scratch, bg_reduce_problem* finish, void* stream);
int bg_critic_head_backward_partial(int32_t B, const float* h, const float* w, const float* values, const float* returns, float* g_hidden,
                                    float* grad_w, float* grad_b, float* grad_b_hidden, double* stats, float* scratch, bg_reduce_problem* finish,
                                    void* stream);
int bg_mlp_layer_backward_partial(int32_t M, int32_t K, int32_t N, const float* G, const float* Wt, const float* act_below, float* Gout,
                                  float* bias_grad_below, float* scratch, bg_reduce_problem* finish, void* stream);

const char* bg_last_error(void);
const char* bg_version(void);

#ifdef __cplusplus
}
#endif
#endif
