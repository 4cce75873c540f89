// One environment step of the T1 task, per wavefront lane (lane pair = one env, one leg per lane).
//
// Restates, fused into a single pass with the state held in registers:
//   reference envs/t1.py:437-497  T1.step       (PD + latency loop, post-physics state, kick/push,
//                                                termination, rewards, reset, teleport, command resample,
//                                                observations, history update)
//   reference envs/t1.py:294-341  reset / _reset_idx
//   reference utils/terrain.py:101-121  terrain_heights (device bilinear, no host round trip)
//   reference utils/recorder.py:36-53   per-episode sums (device accumulators)
// The physics substep is bg_dyn.h.  `X` supplies the lane-pair exchange (DPP lane swap on the GPU,
// a two-thread rendezvous in the host test harness).
#pragma once
#include "../../include/booster_gym_amd.h"
#include "bg_dyn.h"
#include "bg_rng.h"

namespace bg {

// ---- per-env float fields, SoA: value(field, comp, env) = f[(field + comp) * n + env]
enum {
    F_ROOT = 0,            // 13: pos3 quat4(xyzw) linvel3 angvel3 (world)   t1.py:215
    F_Q = 13,              // 12
    F_QD = 25,             // 12
    F_LAST_TGT = 37,       // 12  last_dof_targets   t1.py:247
    F_ACT = 49,            // 12  actions (clipped)  t1.py:243
    F_LAST_ACT = 61,       // 12
    F_LAST_QD = 73,        // 12
    F_LAST_ROOTVEL = 85,   // 6
    F_CMD = 91,            // 3
    F_GAIT_F = 94,         // 1
    F_GAIT_P = 95,         // 1
    F_FILT_LIN = 96,       // 3
    F_FILT_ANG = 99,       // 3
    F_LAST_FEET = 102,     // 6   last_feet_pos [foot][xyz]
    F_PUSH = 108,          // 6   pushing force xyz, torque xyz on the trunk (base coords)
    F_CONTACT = 114,       // 6   net contact force on each foot, world frame
    // per-env constants
    F_KP = 120, F_KD = 132, F_FRIC = 144,  // 12 each   t1.py:69-83
    F_MASS_SCALE = 156,    // 13
    F_COM_OFF = 169,       // 39
    F_FOOT_MAT = 208,      // 6   (friction, compliance, restitution) x 2
    F_BMS = 214,           // 4   base_mass_scaled  t1.py:121,141-152
    F_ORIGIN = 218,        // 3   env_origins
    // derived quantities kept for inspection / parity tests (written every step)
    F_FEET_POS = 221,      // 6
    F_FEET_ROLL = 227, F_FEET_YAW = 229, F_FEET_CONTACT = 231,  // 2 each
    F_TORQUES = 233,       // 12  substep-mean torque  t1.py:456
    F_BASE_LIN = 245, F_BASE_ANG = 248, F_PROJ_G = 251,  // 3 each
    // episode statistics (recorder.py:36-53): running sums of total reward + 26 terms
    F_EP_SUMS = 254,       // 27
    F_COUNT = 281
};
enum { I_EP_LEN = 0, I_CMD_TIME = 1, I_DELAY = 2, I_EP_STEPS = 3, I_CURR_LIN = 4, I_CURR_ANG = 5, I_RESAMPLED = 6, I_COUNT = 7 };  // I_RESAMPLED: this step resampled the env's command (exact-resampling modes)
// global statistics accumulator: [0] finished episodes, [1] sum of their lengths, [2] sum reward, [3..28] sum of terms, [29] non-finite resets
constexpr int STATS_COUNT = 4 + BG_NUM_REWARD_TERMS;  // last entry: resets caused by a non-finite state

// Optional fp16 STORAGE of the per-env dynamic state (BASELINE.json configs[4]: "16384 envs/GPU, fp16 state"; cfg.state_fp16).  Arithmetic
// stays fp32: values are widened on load at the start of the env step and rounded on store at its end, nothing is rounded between substeps.
// Stored as fp16: orientation, linear / angular velocity, joint positions and velocities, targets, actions and their histories, commands,
// gait state, velocity filters, push wrench (105 of the 120 state floats).  Kept fp32: the root POSITION (robots walk tens of metres from
// their origin and fp16 resolves 16 mm at 30 m), last_feet_pos (world positions, same reason), contact forces, all per-env parameters and
// the derived / statistics fields.
typedef uint16_t bg_half_bits;
constexpr int FP16_SLAB_FIELDS = 120;  // [0, F_CONTACT + 6)
BG_HD constexpr bool fp16_state_field(int idx) { return (idx >= 3 && idx < 102) || (idx >= 108 && idx < 114); }
#if defined(__HIPCC__)
BG_HD float half_bits_to_float(bg_half_bits b) { _Float16 h; __builtin_memcpy(&h, &b, 2); return (float)h; }
BG_HD bg_half_bits float_to_half_bits(float v) { _Float16 h = (_Float16)v; bg_half_bits b; __builtin_memcpy(&b, &h, 2); return b; }  // round to nearest even
#endif

struct EnvDev {
    float* f;
    bg_half_bits* h;  // fp16 slab [FP16_SLAB_FIELDS][n], or null
    int32_t* i;
    float* stats;  // [STATS_COUNT], atomics
    float* curr;   // curriculum_prob running sums [(2L+1)*(2A+1)], atomics
    const float* curr_read;  // snapshot taken before this launch: what the samplers read (deterministic)
    int zmask;               // Phys::zmask of this model (which leg links sit on their parent's z axis)
    const ModelDev* model;
    TerrainDev terrain;
    bg_env_cfg cfg;
    int n;
};

struct StepOut {
    float* obs;      // [N][47]
    float* priv;     // [N][14]
    float* rew;      // [N]
    uint8_t* done;   // [N]
    uint8_t* tout;   // [N]
    float* terms;    // [26][N] or null
};

BG_HD float apply_rand(float x, const bg_rand& r, float u, float nrm) {
    // utils/utils.py:5-30 ; gaussian range = (mean, std)  
    if (r.mode == 0) return x;
    float nv = (r.mode <= 2) ? r.a + r.b * nrm : r.a + (r.b - r.a) * u;
    return (r.mode & 1) ? x + nv : x * nv;
}
BG_HD float pymod(float x, float m) {  // python / torch `%` (floor mod) for m > 0
    float r = fmodf(x, m);
    return r < 0.f ? r + m : r;
}
BG_HD float wrap_pi(float x) { return pymod(x + 3.14159265358979f, 6.28318530717959f) - 3.14159265358979f; }

// t1.py:415-435: multinomial draw of a curriculum grid cell from `grid` (values clamped at 1 like curriculum_prob.clamp_), then the command inside
// the cell.  cr.u[0] picks the cell, cr.u[1..3] jitter.
BG_HD void curriculum_draw(const bg_env_cfg& C, const float* grid, const Rand4& cr, float* cmd, int* lin_level, int* ang_level) {
    const int ny = 2 * C.ang_vel_levels + 1, cells = (2 * C.lin_vel_levels + 1) * ny;
    float total = 0.f;
    for (int k = 0; k < cells; k++) total += fminf(grid[k], 1.0f);
    float target = cr.u[0] * total, run = 0.f;
    int idx = cells - 1;
    for (int k = 0; k < cells; k++) {
        run += fminf(grid[k], 1.0f);
        if (run > target) { idx = k; break; }
    }
    *lin_level = idx % ny - C.lin_vel_levels; *ang_level = idx / ny - C.ang_vel_levels;
    cmd[0] = ((float)*lin_level + (cr.u[1] - 0.5f)) * C.lin_vel_x_resolution;
    cmd[1] = fabsf((float)*lin_level) * (2.0f * cr.u[2] - 1.0f) * C.lin_vel_y_resolution;
    cmd[2] = ((float)*ang_level + (cr.u[3] - 0.5f)) * C.ang_vel_resolution;
}

// A keyed pseudo-random PERMUTATION of [0, K): 4-round Feistel network on the smallest even-width bit field covering K, cycle-walked back into
// range.  perm(p) < m for position p selects exactly m of K positions, a random subset (the role of torch.randperm(K)[:m], t1.py:381).
BG_HD uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
BG_HD uint32_t keyed_perm(uint32_t p, uint32_t K, uint32_t key) {
    if (K <= 1u) return 0u;
    uint32_t bits = 2u;
    while (bits < 32u && (1u << bits) < K) bits += 2u;
    const uint32_t hb = bits >> 1, hm = (1u << hb) - 1u;
    uint32_t x = p;
    for (int walk = 0; walk < 64; walk++) {
        uint32_t L = x >> hb, R = x & hm;
        for (uint32_t r = 0; r < 4u; r++) {
            const uint32_t f = mix32(R ^ (key + r * 0x9E3779B9u)) & hm;
            const uint32_t t = L ^ f;
            L = R; R = t;
        }
        x = (L << hb) | R;
        if (x < K) return x;
    }
    return p;  // unreachable in practice (each walk step lands in range with probability > 1/4)
}

// foot pose after the last substep (positions only)
template <int I, class St>
BG_HD void leg_fk(const St& st, const LegParams& lp, const LegState& ls, M3 Rpar, V3 ppar, M3* Rf, V3* pf) {
    constexpr int AX = LEG_AXIS[I], J = Plane<AX>::J, K = Plane<AX>::K;
    float s, c;
    bg_sincos(ls.q[I], &s, &c);
    V3 p = ppar + mul(Rpar, st.template link_pos<I>(lp));
    M3 R = Rpar;
    for (int r = 0; r < 3; r++) {
        R.e[r][J] = c * Rpar.e[r][J] + s * Rpar.e[r][K];
        R.e[r][K] = -s * Rpar.e[r][J] + c * Rpar.e[r][K];
    }
    if constexpr (I + 1 < LEG_LINKS) leg_fk<I + 1>(st, lp, ls, R, p, Rf, pf);
    else { *Rf = R; *pf = p; }
}

BG_HD Phys make_phys(const bg_env_cfg& c) {
    Phys ph;
    ph.dt = c.sim_dt; ph.g = v3(c.gravity[0], c.gravity[1], c.gravity[2]);
    ph.contact_ramp = c.contact_ramp; ph.friction_visc = c.friction_visc; ph.limit_k = c.limit_k; ph.limit_d = c.limit_d; ph.clamp_qd = c.clamp_qd;
    // non-foot shapes: default material (friction 1, restitution 0) averaged with the terrain's, nominal stiffness / damping
    ph.body_gate = c.body_gate_height; ph.body_kn = c.contact_k; ph.body_dn = c.contact_d * (1.0f - 0.5f * c.terrain_restitution);
    ph.body_mu = 0.5f * (1.0f + c.terrain_mu);
    ph.zmask = 0;  // callers that run the sweeps set it from EnvDev::zmask
    ph.self_on = c.self_collisions; ph.self_k = c.self_k; ph.self_d = c.self_d; ph.self_mu = c.self_mu; ph.self_visc = c.self_visc;
    return ph;
}
BG_HD ContactCfg make_contact_cfg(const bg_env_cfg& c) {
    ContactCfg cc; cc.k = c.contact_k; cc.d = c.contact_d; cc.terrain_mu = c.terrain_mu; cc.terrain_restitution = c.terrain_restitution;
    return cc;
}

// yaw of an xyzw quaternion, isaacgym get_euler_xyz convention
BG_HD float quat_yaw(const float q[4]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    return atan2f(2.f * (w * z + x * y), w * w + x * x - y * y - z * z);
}

// Sink: where obs / privileged obs values go.  put(row_local, k, v).
struct GlobalSink {
    static constexpr int SELF = SELF_INLINE;  // how the sink's kernel runs the leg-against-leg narrow phase (bg_dyn.h)
    float* obs; float* priv; int e;
    BG_HD void put_obs(int k, float v) { obs[(size_t)e * BG_NUM_OBS + k] = v; }
    BG_HD void put_priv(int k, float v) { priv[(size_t)e * BG_NUM_PRIV + k] = v; }
};

// mode: 0 = step, 1 = reset-all (T1.reset(), t1.py:294-299: no physics, every env reset + resampled + observed)
// one per-env field: fp32 slab, or the fp16 slab when the env stores its state in fp16 and the field is part of it
template <bool H16>
struct FieldRef {
    float* pf;
    bg_half_bits* ph;
    BG_HD operator float() const {
        if constexpr (H16) { if (ph) return half_bits_to_float(*ph); }
        return *pf;
    }
    BG_HD void operator=(float v) const {
        if constexpr (H16) { if (ph) { *ph = float_to_half_bits(v); return; } }
        *pf = v;
    }
};
template <bool H16>
BG_HD FieldRef<H16> field_ref(float* F, bg_half_bits* H, int idx, int n, int e) {
    const size_t o = (size_t)idx * n + e;
    return FieldRef<H16>{F + o, (H16 && fp16_state_field(idx)) ? H + o : nullptr};
}

// BODY: this env step evaluates the non-foot body contacts (decided by the caller from the trunk height at the START of the env step)
template <class X, class Sink, bool H16, bool BODY>
BG_HD void env_step_lane(const EnvDev& E, X& x, Sink& sink, int e, int leg, bool valid, const float* act, uint32_t step, int mode,
                         const StepOut& out) {
    const bg_env_cfg& C = E.cfg;
    const int n = E.n;
    const ModelDev& M = *E.model;
    float* const F = E.f;
    bg_half_bits* const HS = E.h;
    int32_t* const II = E.i;
#define FLD(off, comp) field_ref<H16>(F, HS, (off) + (comp), n, e)
    const int j0 = leg * LEG_LINKS;
    const float dt_env = C.sim_dt * (float)C.decimation;
    Phys ph = make_phys(C);
    ph.zmask = E.zmask;
    ContactCfg cc = make_contact_cfg(C);

    // ------------------------------------------------------------ load
    BaseState bs;
    for (int a = 0; a < 3; a++) { bs.pos.e[a] = FLD(F_ROOT, a); bs.vlin.e[a] = FLD(F_ROOT, 7 + a); bs.vang.e[a] = FLD(F_ROOT, 10 + a); }
    for (int a = 0; a < 4; a++) bs.quat[a] = FLD(F_ROOT, 3 + a);
    LegState ls;
    float last_tgt[LEG_LINKS], kp[LEG_LINKS], kd[LEG_LINKS], fric[LEG_LINKS], a6[LEG_LINKS], last_act[LEG_LINKS], last_qd[LEG_LINKS];
    for (int i = 0; i < LEG_LINKS; i++) {
        ls.q[i] = FLD(F_Q, j0 + i); ls.qd[i] = FLD(F_QD, j0 + i);
        last_tgt[i] = FLD(F_LAST_TGT, j0 + i);
        kp[i] = FLD(F_KP, j0 + i); kd[i] = FLD(F_KD, j0 + i); fric[i] = FLD(F_FRIC, j0 + i);
        last_act[i] = FLD(F_LAST_ACT, j0 + i); last_qd[i] = FLD(F_LAST_QD, j0 + i);
    }
    LegParams lp;
    load_leg_params(M, cc, leg, e, n, F + (size_t)F_MASS_SCALE * n, F + (size_t)F_COM_OFF * n, F + (size_t)F_FOOT_MAT * n, lp);
    LinkConst bk = load_base_link(M, e, n, F + (size_t)F_MASS_SCALE * n, F + (size_t)F_COM_OFF * n);
    int ep_len = II[(size_t)I_EP_LEN * n + e], cmd_time = II[(size_t)I_CMD_TIME * n + e], delay = II[(size_t)I_DELAY * n + e];
    float cmd[3] = {FLD(F_CMD, 0), FLD(F_CMD, 1), FLD(F_CMD, 2)};
    float gait_f = FLD(F_GAIT_F, 0), gait_p = FLD(F_GAIT_P, 0);
    V3 filt_lin = v3(FLD(F_FILT_LIN, 0), FLD(F_FILT_LIN, 1), FLD(F_FILT_LIN, 2));
    V3 filt_ang = v3(FLD(F_FILT_ANG, 0), FLD(F_FILT_ANG, 1), FLD(F_FILT_ANG, 2));
    V3 push_f = v3(FLD(F_PUSH, 0), FLD(F_PUSH, 1), FLD(F_PUSH, 2)), push_t = v3(FLD(F_PUSH, 3), FLD(F_PUSH, 4), FLD(F_PUSH, 5));
    V3 last_foot = v3(FLD(F_LAST_FEET, 3 * leg), FLD(F_LAST_FEET, 3 * leg + 1), FLD(F_LAST_FEET, 3 * leg + 2));
    float last_rootvel[6];
    for (int a = 0; a < 6; a++) last_rootvel[a] = FLD(F_LAST_ROOTVEL, a);

    float tmean[LEG_LINKS];
    float body_pen = 0.f, body_term = 0.f;  // penalised / terminating non-foot bodies of this leg in contact (last substep)
    V3 foot_force = v3(0.f, 0.f, 0.f);
    for (int i = 0; i < LEG_LINKS; i++) { tmean[i] = 0.f; a6[i] = FLD(F_ACT, j0 + i); }

    typename Sink::Ctx cx;  // sweep work space; the sink decides where the per-env link constants live (registers or LDS)
    sink.bind(cx, lp);
    if (mode == 0) {
        // ------------------------------------------------------------ pre-physics (t1.py:439-440)
        float target[LEG_LINKS];
        for (int i = 0; i < LEG_LINKS; i++) {
            float a = act[(size_t)e * BG_NUM_DOFS + j0 + i];
            if (((float_bits(a) >> 23) & 0xFFu) == 0xFFu) a = 0.f;  // NaN / Inf action -> 0 (bit test, see the non-finite guard below)
            a = fminf(fmaxf(a, -C.clip_actions), C.clip_actions);
            a6[i] = a;
            target[i] = C.default_dof_pos[j0 + i] + C.action_scale * a;
        }
        // applied at the trunk's centre of mass, trunk frame (apply_rigid_body_force_tensors LOCAL_SPACE, t1.py:522-527)
        SV wrench = local_wrench_at_com(bk, push_f, push_t);
        // ------------------------------------------------------------ physics substeps (t1.py:443-456)
        for (int s = 0; s < C.decimation; s++) {
            float tau[LEG_LINKS];
            for (int i = 0; i < LEG_LINKS; i++) {
                if (delay == s) last_tgt[i] = target[i];
                tau[i] = pd_torque(kp[i], kd[i], fric[i], M.tau_lim[j0 + i], last_tgt[i], ls.q[i], ls.qd[i]);
                tmean[i] += tau[i];
            }
            BodyContactOut bo;
            BaseContribution mine = substep_pre<BODY, Sink::SELF>(ph, E.terrain, M, leg, lp, ls, tau, bs, cx, x, (const SV*)nullptr, &bo);
            // contact forces of the LAST substep are the ones the task logic sees (contact_collection: last substep, T1.yaml:56): how many
            // penalised / terminating non-foot bodies of this leg carry more than 1 N (t1.py:553,629); the trunk is counted once (leg 0)
            body_pen = 0.f; body_term = 0.f;
            if constexpr (BODY) {
                V3 tf = bo.trunk;
                for (int a = 0; a < 3; a++) tf.e[a] += x.swap(tf.e[a]);
                const float trunk_hit = (leg == 0 && dot(tf, tf) > 1.0f) ? 1.f : 0.f;
                body_pen = (C.penalized_body_mask & 1) ? trunk_hit : 0.f;
                body_term = (C.terminate_body_mask & 1) ? trunk_hit : 0.f;
                for (int i = 0; i < LEG_LINKS - 1; i++) {
                    const float hit = dot(bo.link[i], bo.link[i]) > 1.0f ? 1.f : 0.f;
                    const int bit = 1 << (1 + j0 + i);
                    if (C.penalized_body_mask & bit) body_pen += hit;
                    if (C.terminate_body_mask & bit) body_term += hit;
                }
            } else {  // trunk high: the only non-foot contact there can be is the shank against the other leg
                const float hit = dot(bo.link[SELF_SHANK], bo.link[SELF_SHANK]) > 1.0f ? 1.f : 0.f;
                const int bit = 1 << (1 + j0 + SELF_SHANK);
                body_pen = (C.penalized_body_mask & bit) ? hit : 0.f;
                body_term = (C.terminate_body_mask & bit) ? hit : 0.f;
            }
            BaseContribution both;
            for (int k = 0; k < 6; k++) { both.I.A.e[k] = mine.I.A.e[k] + x.swap(mine.I.A.e[k]); both.I.M.e[k] = mine.I.M.e[k] + x.swap(mine.I.M.e[k]); }
            for (int r = 0; r < 3; r++) for (int c2 = 0; c2 < 3; c2++) both.I.H.e[r][c2] = mine.I.H.e[r][c2] + x.swap(mine.I.H.e[r][c2]);
            for (int k = 0; k < 3; k++) { both.p.a.e[k] = mine.p.a.e[k] + x.swap(mine.p.a.e[k]); both.p.l.e[k] = mine.p.l.e[k] + x.swap(mine.p.l.e[k]); }
            float qdd[LEG_LINKS];
            V3 lin_w, ang_w;
            SV wr = wrench;
            if (s != 0) wr = sv_zero();  // applied body forces last for one simulate()
            substep_solve(ph, bk, lp, ls, cx, both, wr, qdd, &lin_w, &ang_w, &foot_force);
            substep_integrate(ph, lp, ls, bs, qdd, lin_w, ang_w);
        }
        for (int i = 0; i < LEG_LINKS; i++) tmean[i] *= 1.0f / (float)C.decimation;
    }

    // ------------------------------------------------------------ non-finite guard (PhysX clamps internally; here a blown-up
    // state is treated as a termination so that one diverged env cannot poison its lane pair forever)
    // (bit tests: the kernels are compiled with -ffinite-math-only, so floating-point comparisons may not be used to detect NaN / Inf)
    uint32_t expo = 0;
    {
        const float st[13] = {bs.pos.e[0], bs.pos.e[1], bs.pos.e[2], bs.quat[0], bs.quat[1], bs.quat[2], bs.quat[3],
                              bs.vlin.e[0], bs.vlin.e[1], bs.vlin.e[2], bs.vang.e[0], bs.vang.e[1], bs.vang.e[2]};
        for (int k = 0; k < 13; k++) expo |= (uint32_t)(((float_bits(st[k]) >> 23) & 0xFFu) >= 0xE6u);  // |x| >= 2^103, Inf or NaN
        for (int i = 0; i < LEG_LINKS; i++) {
            expo |= (uint32_t)(((float_bits(ls.q[i]) >> 23) & 0xFFu) >= 0xE6u);
            expo |= (uint32_t)(((float_bits(ls.qd[i]) >> 23) & 0xFFu) >= 0xE6u);
        }
    }
    float bad = expo ? 1.f : 0.f;
    bad = (bad != 0.f || x.swap(bad) != 0.f) ? 1.f : 0.f;
    if (bad != 0.f) {
        bs.pos = v3(0.f, 0.f, 1.f); bs.quat[0] = bs.quat[1] = bs.quat[2] = 0.f; bs.quat[3] = 1.f;
        bs.vlin = v3(0.f, 0.f, 0.f); bs.vang = v3(0.f, 0.f, 0.f);
        for (int i = 0; i < LEG_LINKS; i++) { ls.q[i] = 0.f; ls.qd[i] = 0.f; tmean[i] = 0.f; }
        foot_force = v3(0.f, 0.f, 0.f);
    }
    // ------------------------------------------------------------ post-physics derived state (t1.py:460-474)
    M3 R0 = quat_to_mat(bs.quat);
    V3 base_lin = mulT(R0, bs.vlin), base_ang = mulT(R0, bs.vang);
    V3 proj_g = mulT(R0, v3(0.f, 0.f, -1.f));
    if (mode == 0) {
        filt_lin = C.filter_weight * base_lin + (1.0f - C.filter_weight) * filt_lin;
        filt_ang = C.filter_weight * base_ang + (1.0f - C.filter_weight) * filt_ang;
    }
    M3 Rf; V3 pf;
    leg_fk<0>(cx.w.st, lp, ls, R0, bs.pos, &Rf, &pf);
    float roll = atan2f(Rf.e[2][1], Rf.e[2][2]), yaw = atan2f(Rf.e[1][0], Rf.e[0][0]);  // get_euler_xyz, wrapped to (-pi, pi]
    float fcontact = 0.f;
    for (int k = 0; k < 4; k++) {
        V3 xw = pf + mul(Rf, lp.corner[k]);
        if (xw.e[2] - terrain_height(E.terrain, xw.e[0], xw.e[1]) < 0.01f) fcontact = 1.f;  // t1.py:545
    }
    // per-leg partial sums of the joint-space reward terms (t1.py:643-694)
    float s_tau2 = 0.f, s_tired = 0.f, s_power = 0.f, s_qd2 = 0.f, s_acc = 0.f, s_arate = 0.f, s_poslim = 0.f, s_vellim = 0.f, s_taulim = 0.f;
    for (int i = 0; i < LEG_LINKS; i++) {
        float lim = M.tau_lim[j0 + i], lo = M.q_lo[j0 + i], hi = M.q_hi[j0 + i];
        s_tau2 += tmean[i] * tmean[i];
        float tr = tmean[i] / lim;
        s_tired += fminf(tr * tr, 1.0f);
        s_power += fmaxf(tmean[i] * ls.qd[i], 0.f);
        s_qd2 += ls.qd[i] * ls.qd[i];
        float acc = (last_qd[i] - ls.qd[i]) / dt_env;
        s_acc += acc * acc;
        float da = last_act[i] - a6[i];
        s_arate += da * da;
        float lower = lo + 0.5f * (1.f - C.soft_dof_pos_limit) * (hi - lo), upper = hi - 0.5f * (1.f - C.soft_dof_pos_limit) * (hi - lo);
        s_poslim += (ls.q[i] < lower || ls.q[i] > upper) ? 1.f : 0.f;
        s_vellim += fminf(fmaxf(fabsf(ls.qd[i]) - M.qd_max[j0 + i] * C.soft_dof_vel_limit, 0.f), 1.f);
        s_taulim += fmaxf(fabsf(tmean[i]) - lim * C.soft_torque_limit, 0.f);
    }
    V3 dfoot = (1.0f / dt_env) * (last_foot - pf);
    float slip_own = dot(dfoot, dfoot) * fcontact;
    float velz_own = dfoot.e[2] * dfoot.e[2];
    // ---- lane-pair exchange: partner foot + partial sums
    V3 pf_o = v3(x.swap(pf.e[0]), x.swap(pf.e[1]), x.swap(pf.e[2]));
    float roll_o = x.swap(roll), yaw_o = x.swap(yaw), fcontact_o = x.swap(fcontact);
    s_tau2 += x.swap(s_tau2); s_tired += x.swap(s_tired); s_power += x.swap(s_power); s_qd2 += x.swap(s_qd2); s_acc += x.swap(s_acc);
    s_arate += x.swap(s_arate); s_poslim += x.swap(s_poslim); s_vellim += x.swap(s_vellim); s_taulim += x.swap(s_taulim);
    float slip = slip_own + x.swap(slip_own), velz = velz_own + x.swap(velz_own);
    V3 foot_force_o = v3(x.swap(foot_force.e[0]), x.swap(foot_force.e[1]), x.swap(foot_force.e[2]));
    // order feet as (left, right)
    V3 pL = leg == 0 ? pf : pf_o, pR = leg == 0 ? pf_o : pf;
    float yawL = leg == 0 ? yaw : yaw_o, yawR = leg == 0 ? yaw_o : yaw, rollL = leg == 0 ? roll : roll_o, rollR = leg == 0 ? roll_o : roll;
    float conL = leg == 0 ? fcontact : fcontact_o, conR = leg == 0 ? fcontact_o : fcontact;

    uint8_t reset_flag = 0, tout_flag = 0;
    float rew_total = 0.f;
    float term[BG_NUM_REWARD_TERMS];
    for (int k = 0; k < BG_NUM_REWARD_TERMS; k++) term[k] = 0.f;
    const uint32_t cnt = step + 1;  // common_step_counter after its increment (t1.py:477)
    const uint32_t so = mode ? 64u : 0u;  // T1.reset() draws from its own streams

    if (mode == 0) {
        ep_len += 1;
        gait_p = fmodf(gait_p + dt_env * gait_f, 1.0f);  // t1.py:478
        // ------------------------------------------------------------ kick (t1.py:499-504)
        if (C.kick_interval > 0 && cnt % (uint32_t)C.kick_interval == 0) {
            Rand4 r0 = rand4(C.seed, (uint32_t)e, step, so + RS_KICK0), r1 = rand4(C.seed, (uint32_t)e, step, so + RS_KICK1);
            for (int a = 0; a < 3; a++) bs.vlin.e[a] = apply_rand(bs.vlin.e[a], C.kick_lin_vel, r0.u[a], r0.n[a]);
            bs.vang.e[0] = apply_rand(bs.vang.e[0], C.kick_ang_vel, r0.u[3], r0.n[3]);
            bs.vang.e[1] = apply_rand(bs.vang.e[1], C.kick_ang_vel, r1.u[0], r1.n[0]);
            bs.vang.e[2] = apply_rand(bs.vang.e[2], C.kick_ang_vel, r1.u[1], r1.n[1]);
        }
        // ------------------------------------------------------------ push schedule (t1.py:506-520)
        if (C.push_interval > 0) {
            uint32_t ph_ = cnt % (uint32_t)C.push_interval;
            if (ph_ == 0) {
                Rand4 r0 = rand4(C.seed, (uint32_t)e, step, so + RS_PUSH0), r1 = rand4(C.seed, (uint32_t)e, step, so + RS_PUSH1);
                for (int a = 0; a < 3; a++) push_f.e[a] = apply_rand(0.f, C.push_force, r0.u[a], r0.n[a]);
                push_t.e[0] = apply_rand(0.f, C.push_torque, r0.u[3], r0.n[3]);
                push_t.e[1] = apply_rand(0.f, C.push_torque, r1.u[0], r1.n[0]);
                push_t.e[2] = apply_rand(0.f, C.push_torque, r1.u[1], r1.n[1]);
            } else if (ph_ == (uint32_t)C.push_duration) {
                push_f = v3(0.f, 0.f, 0.f); push_t = v3(0.f, 0.f, 0.f);
            }
        }
        // ------------------------------------------------------------ termination (t1.py:551-558)
        float h_base = terrain_height(E.terrain, bs.pos.e[0], bs.pos.e[1]);
        float v2 = dot(bs.vlin, bs.vlin) + dot(bs.vang, bs.vang);
        bool to = ep_len > C.max_episode_length;
        {   // the feet rows of the contact tensor (this leg's foot, last substep)
            const float hit = dot(foot_force, foot_force) > 1.0f ? 1.f : 0.f;
            const int bit = 1 << (1 + j0 + LEG_LINKS - 1);
            if (C.penalized_body_mask & bit) body_pen += hit;
            if (C.terminate_body_mask & bit) body_term += hit;
        }
        body_pen += x.swap(body_pen);
        body_term += x.swap(body_term);
        bool rs = (body_term > 0.f) || (v2 > C.terminate_vel) || (bs.pos.e[2] - h_base < C.terminate_height) || to;
        to = to || (ep_len == cmd_time);
        if (bad != 0.f) rs = true;
        reset_flag = rs ? 1 : 0; tout_flag = to ? 1 : 0;
        // ------------------------------------------------------------ rewards (t1.py:560-572, 606-730)
        float base_yaw = quat_yaw(bs.quat);
        float bh = bs.pos.e[2] - h_base - C.base_height_target;
        float ex = cmd[0] - filt_lin.e[0], ey = cmd[1] - filt_lin.e[1], ew = cmd[2] - filt_ang.e[2];
        float racc = 0.f;
        {
            float rv[6] = {bs.vlin.e[0], bs.vlin.e[1], bs.vlin.e[2], bs.vang.e[0], bs.vang.e[1], bs.vang.e[2]};
            for (int a = 0; a < 6; a++) { float d = (last_rootvel[a] - rv[a]) / dt_env; racc += d * d; }
        }
        float ydiff = wrap_pi(yawR - yawL);
        float ymean = 0.5f * (yawL + yawR) + (fabsf(yawR - yawL) > 3.14159265358979f ? 3.14159265358979f : 0.f);
        float ymerr = wrap_pi(base_yaw - ymean);
        float fdist = fabsf(cosf(base_yaw) * (pR.e[1] - pL.e[1]) - sinf(base_yaw) * (pR.e[0] - pL.e[0]));
        bool gait_on = gait_f > 1.0e-8f;
        bool lsw = (fabsf(gait_p - 0.25f) < 0.5f * C.swing_period) && gait_on, rsw = (fabsf(gait_p - 0.75f) < 0.5f * C.swing_period) && gait_on;
        term[BG_REW_SURVIVAL] = 1.0f;
        term[BG_REW_TRACKING_LIN_VEL_X] = expf(-ex * ex / C.tracking_sigma);
        term[BG_REW_TRACKING_LIN_VEL_Y] = expf(-ey * ey / C.tracking_sigma);
        term[BG_REW_TRACKING_ANG_VEL] = expf(-ew * ew / C.tracking_sigma);
        term[BG_REW_BASE_HEIGHT] = bh * bh;
        term[BG_REW_ORIENTATION] = proj_g.e[0] * proj_g.e[0] + proj_g.e[1] * proj_g.e[1];
        term[BG_REW_TORQUES] = s_tau2;
        term[BG_REW_TORQUE_TIREDNESS] = s_tired;
        term[BG_REW_POWER] = s_power;
        term[BG_REW_LIN_VEL_Z] = filt_lin.e[2] * filt_lin.e[2];
        term[BG_REW_ANG_VEL_XY] = base_ang.e[0] * base_ang.e[0] + base_ang.e[1] * base_ang.e[1];
        term[BG_REW_DOF_VEL] = s_qd2;
        term[BG_REW_DOF_ACC] = s_acc;
        term[BG_REW_ROOT_ACC] = racc;
        term[BG_REW_ACTION_RATE] = s_arate;
        term[BG_REW_DOF_POS_LIMITS] = s_poslim;
        term[BG_REW_DOF_VEL_LIMITS] = s_vellim;
        term[BG_REW_TORQUE_LIMITS] = s_taulim;
        term[BG_REW_COLLISION] = body_pen;  // number of penalised bodies with |contact force| > 1 N (t1.py:627-629)
        term[BG_REW_FEET_SLIP] = slip * (ep_len > 1 ? 1.f : 0.f);
        term[BG_REW_FEET_VEL_Z] = velz;
        term[BG_REW_FEET_YAW_DIFF] = ydiff * ydiff;
        term[BG_REW_FEET_YAW_MEAN] = ymerr * ymerr;
        term[BG_REW_FEET_ROLL] = rollL * rollL + rollR * rollR;
        term[BG_REW_FEET_DISTANCE] = fminf(fmaxf(C.feet_distance_ref - fdist, 0.f), 0.1f);
        term[BG_REW_FEET_SWING] = ((lsw && conL == 0.f) ? 1.f : 0.f) + ((rsw && conR == 0.f) ? 1.f : 0.f);
        for (int k = 0; k < BG_NUM_REWARD_TERMS; k++) {
            term[k] = C.reward_scale[k] != 0.f ? term[k] * C.reward_scale[k] : 0.f;
            rew_total += term[k];
        }
        if (C.only_positive_rewards) rew_total = fmaxf(rew_total, 0.f);
        if (bad != 0.f) { rew_total = 0.f; for (int k = 0; k < BG_NUM_REWARD_TERMS; k++) term[k] = 0.f; }
    } else {
        reset_flag = 1;
    }

    // ------------------------------------------------------------ reset (t1.py:301-341)
    V3 pf_store = pf;
    // ------------------------------------------------------------ command curriculum update (t1.py:391-413), pre-reset values
    if (reset_flag && mode == 0 && C.curriculum && leg == 0 && valid && bad == 0.f) {
        const int nx = 2 * C.lin_vel_levels + 1, ny = 2 * C.ang_vel_levels + 1;
        bool success = (float)ep_len > ceilf((float)C.max_episode_length) * (1.0f - C.episode_length_toler);
        success = success && fabsf(filt_lin.e[0] - cmd[0]) < C.lin_vel_x_toler && fabsf(filt_lin.e[1] - cmd[1]) < C.lin_vel_y_toler &&
                  fabsf(filt_ang.e[2] - cmd[2]) < C.ang_vel_yaw_toler;
        if (success) {
            const int cx = II[(size_t)I_CURR_LIN * n + e] + C.lin_vel_levels, cy = II[(size_t)I_CURR_ANG * n + e] + C.ang_vel_levels;
            const float r = C.curriculum_update_rate;
#if defined(__HIP_DEVICE_COMPILE__)
#define BG_ACC(ix, iy) atomicAdd(&E.curr[(ix) * ny + (iy)], r)
#else
#define BG_ACC(ix, iy) (E.curr[(ix) * ny + (iy)] += r)
#endif
            BG_ACC(cx, cy);
            if (cx > 0) BG_ACC(cx - 1, cy);
            if (cx < nx - 1) BG_ACC(cx + 1, cy);
            if (cy > 0) BG_ACC(cx, cy - 1);
            if (cy < ny - 1) BG_ACC(cx, cy + 1);
#undef BG_ACC
        }
    }
    if (reset_flag) {
        uint32_t noise_env = C.shared_reset_noise ? 0xFFFFFFFFu : (uint32_t)e;
        Rand4 d0 = rand4(C.seed, noise_env, step, so + RS_RESETDOF + leg * 2), d1 = rand4(C.seed, noise_env, step, so + RS_RESETDOF + leg * 2 + 1);
        for (int i = 0; i < LEG_LINKS; i++) {
            float u = i < 4 ? d0.u[i] : d1.u[i - 4], nn = i < 4 ? d0.n[i] : d1.n[i - 4];
            ls.q[i] = apply_rand(C.default_dof_pos[j0 + i], C.init_dof_pos, u, nn);
            ls.qd[i] = 0.f;
            last_tgt[i] = ls.q[i];
        }
        Rand4 r0 = rand4(C.seed, (uint32_t)e, step, so + RS_RESET0), r1 = rand4(C.seed, (uint32_t)e, step, so + RS_RESET1);
        float px = apply_rand(C.base_init_state[0] + FLD(F_ORIGIN, 0), C.init_base_pos_xy, r0.u[0], r0.n[0]);
        float py = apply_rand(C.base_init_state[1] + FLD(F_ORIGIN, 1), C.init_base_pos_xy, r0.u[1], r0.n[1]);
        bs.pos = v3(px, py, C.base_init_state[2] + terrain_height(E.terrain, px, py));
        float yw = r0.u[2] * 6.28318530717959f, sy, cy;
        bg_sincos(0.5f * yw, &sy, &cy);
        bs.quat[0] = 0.f; bs.quat[1] = 0.f; bs.quat[2] = sy; bs.quat[3] = cy;  // quat_from_euler_xyz(0, 0, yaw)
        bs.vlin = v3(apply_rand(0.f, C.init_base_lin_vel_xy, r1.u[0], r1.n[0]), apply_rand(0.f, C.init_base_lin_vel_xy, r1.u[1], r1.n[1]), C.base_init_state[9]);
        bs.vang = v3(C.base_init_state[10], C.base_init_state[11], C.base_init_state[12]);
        ep_len = 0; cmd_time = 0;
        filt_lin = v3(0.f, 0.f, 0.f); filt_ang = v3(0.f, 0.f, 0.f);
        int dl = (int)(r0.u[3] * (float)C.decimation);
        delay = dl < C.decimation ? dl : C.decimation - 1;  // randint(0, decimation), t1.py:316
    }
    // ------------------------------------------------------------ teleport (t1.py:343-360)
    if (mode == 0 && C.terrain_type != 0) {
        float sx = 0.f, sy2 = 0.f;
        float wx = C.terrain_env_width + C.terrain_border, wy = C.terrain_env_length + C.terrain_border;
        if (bs.pos.e[0] < -0.75f * C.terrain_border) sx += wx;
        if (bs.pos.e[0] > C.terrain_env_width + 0.75f * C.terrain_border) sx -= wx;
        if (bs.pos.e[1] < -0.75f * C.terrain_border) sy2 += wy;
        if (bs.pos.e[1] > C.terrain_env_length + 0.75f * C.terrain_border) sy2 -= wy;
        bs.pos.e[0] += sx; bs.pos.e[1] += sy2;
        pf_store.e[0] += sx; pf_store.e[1] += sy2;
    }
    // ------------------------------------------------------------ command resample (t1.py:362-389)
    // Reference-exact modes (cfg.exact_still_count / cfg.same_step_curriculum): the parts of _resample_commands that couple envs -- which envs stand
    // still (an exact count over the envs resampling in this step) and the curriculum draw from the grid as updated by THIS step's resets -- are
    // left to resample_apply_kernel, which runs after this launch and patches commands, gait frequency and the command entries of the observation.
    const bool defer_still = C.exact_still_count != 0, defer_curr = C.curriculum && C.same_step_curriculum != 0;
    int resampled = 0;
    if (ep_len == cmd_time) {
        resampled = 1;
        Rand4 c0 = rand4(C.seed, (uint32_t)e, step, so + RS_CMD0), c1 = rand4(C.seed, (uint32_t)e, step, so + RS_CMD1);
        if (C.curriculum && !defer_curr) {
            // t1.py:415-435: draw a grid cell ~ multinomial(curriculum_prob as of the start of this step), then jitter inside the cell.  The reference decodes
            // lin = idx % cols - L and ang = idx // cols - A although it updates prob[lin + L][ang + A] (a transposition); kept.
            Rand4 cr = rand4(C.seed, (uint32_t)e, step, so + RS_CURR);
            int lin_level, ang_level;
            curriculum_draw(C, E.curr_read, cr, cmd, &lin_level, &ang_level);
            if (leg == 0 && valid) { II[(size_t)I_CURR_LIN * n + e] = lin_level; II[(size_t)I_CURR_ANG * n + e] = ang_level; }
        } else if (!C.curriculum) {
            cmd[0] = C.cmd_lin_vel_x[0] + (C.cmd_lin_vel_x[1] - C.cmd_lin_vel_x[0]) * c0.u[0];
            cmd[1] = C.cmd_lin_vel_y[0] + (C.cmd_lin_vel_y[1] - C.cmd_lin_vel_y[0]) * c0.u[1];
            cmd[2] = C.cmd_ang_vel_yaw[0] + (C.cmd_ang_vel_yaw[1] - C.cmd_ang_vel_yaw[0]) * c0.u[2];
        }
        gait_f = C.cmd_gait_frequency[0] + (C.cmd_gait_frequency[1] - C.cmd_gait_frequency[0]) * c0.u[3];
        // per-env Bernoulli (reference: exact count via randperm -> cfg.exact_still_count)
        if (!defer_still && !defer_curr && c1.u[0] < C.still_proportion) { cmd[0] = cmd[1] = cmd[2] = 0.f; gait_f = 0.f; }
        int span = C.resample_steps[1] - C.resample_steps[0];
        int add = C.resample_steps[0] + (span > 0 ? (int)(c1.u[1] * (float)span) : 0);
        if (span > 0 && add >= C.resample_steps[1]) add = C.resample_steps[1] - 1;
        cmd_time += add;
    }
    if ((defer_still || defer_curr) && leg == 0 && valid) II[(size_t)I_RESAMPLED * n + e] = resampled;
    // ------------------------------------------------------------ observations (t1.py:574-603)
    {
        Rand4 o0 = rand4(C.seed, (uint32_t)e, step, so + RS_OBS0), o1 = rand4(C.seed, (uint32_t)e, step, so + RS_OBS1), o2 = rand4(C.seed, (uint32_t)e, step, so + RS_OBS2);
        Rand4 p0 = rand4(C.seed, (uint32_t)e, step, so + RS_DOFPOS + leg * 2), p1 = rand4(C.seed, (uint32_t)e, step, so + RS_DOFPOS + leg * 2 + 1);
        Rand4 v0 = rand4(C.seed, (uint32_t)e, step, so + RS_DOFVEL + leg * 2), v1 = rand4(C.seed, (uint32_t)e, step, so + RS_DOFVEL + leg * 2 + 1);
        if (valid) {
            if (leg == 0) {
                for (int a = 0; a < 3; a++) sink.put_obs(a, apply_rand(proj_g.e[a], C.noise_gravity, o0.u[a], o0.n[a]) * C.norm_gravity);
                sink.put_obs(3, apply_rand(base_ang.e[0], C.noise_ang_vel, o0.u[3], o0.n[3]) * C.norm_ang_vel);
                sink.put_obs(4, apply_rand(base_ang.e[1], C.noise_ang_vel, o1.u[0], o1.n[0]) * C.norm_ang_vel);
                sink.put_obs(5, apply_rand(base_ang.e[2], C.noise_ang_vel, o1.u[1], o1.n[1]) * C.norm_ang_vel);
                sink.put_obs(6, cmd[0] * C.norm_lin_vel); sink.put_obs(7, cmd[1] * C.norm_lin_vel); sink.put_obs(8, cmd[2] * C.norm_ang_vel);
                float sg, cg;
                bg_sincos(6.28318530717959f * gait_p, &sg, &cg);
                float on = gait_f > 1.0e-8f ? 1.f : 0.f;
                sink.put_obs(9, cg * on); sink.put_obs(10, sg * on);
                // privileged (t1.py:593-602)
                for (int a = 0; a < 4; a++) sink.put_priv(a, FLD(F_BMS, a));
                sink.put_priv(4, apply_rand(base_lin.e[0], C.noise_lin_vel, o1.u[2], o1.n[2]) * C.norm_lin_vel);
                sink.put_priv(5, apply_rand(base_lin.e[1], C.noise_lin_vel, o1.u[3], o1.n[3]) * C.norm_lin_vel);
                sink.put_priv(6, apply_rand(base_lin.e[2], C.noise_lin_vel, o2.u[0], o2.n[0]) * C.norm_lin_vel);
                float hh = bs.pos.e[2] - terrain_height(E.terrain, bs.pos.e[0], bs.pos.e[1]);
                sink.put_priv(7, apply_rand(hh, C.noise_height, o2.u[1], o2.n[1]));
                for (int a = 0; a < 3; a++) { sink.put_priv(8 + a, push_f.e[a] * C.norm_push_force); sink.put_priv(11 + a, push_t.e[a] * C.norm_push_torque); }
            }
            for (int i = 0; i < LEG_LINKS; i++) {
                float up = i < 4 ? p0.u[i] : p1.u[i - 4], np_ = i < 4 ? p0.n[i] : p1.n[i - 4];
                float uv = i < 4 ? v0.u[i] : v1.u[i - 4], nv = i < 4 ? v0.n[i] : v1.n[i - 4];
                sink.put_obs(11 + j0 + i, apply_rand(ls.q[i] - C.default_dof_pos[j0 + i], C.noise_dof_pos, up, np_) * C.norm_dof_pos);
                sink.put_obs(23 + j0 + i, apply_rand(ls.qd[i], C.noise_dof_vel, uv, nv) * C.norm_dof_vel);
                sink.put_obs(35 + j0 + i, a6[i]);
            }
        }
    }
    if (!valid) return;
    // ------------------------------------------------------------ store (history update t1.py:492-495)
    for (int i = 0; i < LEG_LINKS; i++) {
        FLD(F_Q, j0 + i) = ls.q[i]; FLD(F_QD, j0 + i) = ls.qd[i];
        FLD(F_LAST_TGT, j0 + i) = last_tgt[i];
        FLD(F_ACT, j0 + i) = a6[i];
        if (mode == 0) { FLD(F_LAST_ACT, j0 + i) = a6[i]; FLD(F_TORQUES, j0 + i) = tmean[i]; }
        FLD(F_LAST_QD, j0 + i) = ls.qd[i];
    }
    for (int a = 0; a < 3; a++) {
        FLD(F_LAST_FEET, 3 * leg + a) = pf_store.e[a];
        FLD(F_FEET_POS, 3 * leg + a) = pf_store.e[a];
        FLD(F_CONTACT, 3 * leg + a) = foot_force.e[a];
    }
    FLD(F_FEET_ROLL, leg) = roll; FLD(F_FEET_YAW, leg) = yaw; FLD(F_FEET_CONTACT, leg) = fcontact;
    (void)foot_force_o;
    if (leg == 0) {
        for (int a = 0; a < 3; a++) {
            FLD(F_ROOT, a) = bs.pos.e[a]; FLD(F_ROOT, 7 + a) = bs.vlin.e[a]; FLD(F_ROOT, 10 + a) = bs.vang.e[a];
            FLD(F_LAST_ROOTVEL, a) = bs.vlin.e[a]; FLD(F_LAST_ROOTVEL, 3 + a) = bs.vang.e[a];
            FLD(F_CMD, a) = cmd[a];
            FLD(F_FILT_LIN, a) = filt_lin.e[a]; FLD(F_FILT_ANG, a) = filt_ang.e[a];
            FLD(F_PUSH, a) = push_f.e[a]; FLD(F_PUSH, 3 + a) = push_t.e[a];
            FLD(F_BASE_LIN, a) = base_lin.e[a]; FLD(F_BASE_ANG, a) = base_ang.e[a]; FLD(F_PROJ_G, a) = proj_g.e[a];
        }
        for (int a = 0; a < 4; a++) FLD(F_ROOT, 3 + a) = bs.quat[a];
        FLD(F_GAIT_F, 0) = gait_f; FLD(F_GAIT_P, 0) = gait_p;
        II[(size_t)I_EP_LEN * n + e] = ep_len; II[(size_t)I_CMD_TIME * n + e] = cmd_time; II[(size_t)I_DELAY * n + e] = delay;
        if (mode == 0) {
            out.rew[e] = rew_total; out.done[e] = reset_flag; out.tout[e] = tout_flag;
            if (out.terms) for (int k = 0; k < BG_NUM_REWARD_TERMS; k++) out.terms[(size_t)k * n + e] = term[k];
            // episode statistics (recorder.py:36-53): running sums, flushed to the global accumulator when the episode ends
            int eps = II[(size_t)I_EP_STEPS * n + e] + 1;
            float es = FLD(F_EP_SUMS, 0) + rew_total;
            if (bad != 0.f) {
#if defined(__HIP_DEVICE_COMPILE__)
                atomicAdd(&E.stats[3 + BG_NUM_REWARD_TERMS], 1.0f);
#else
                E.stats[3 + BG_NUM_REWARD_TERMS] += 1.0f;
#endif
            }
            if (reset_flag) {
#if defined(__HIP_DEVICE_COMPILE__)
                atomicAdd(&E.stats[0], 1.0f); atomicAdd(&E.stats[1], (float)eps); atomicAdd(&E.stats[2], es);
#else
                E.stats[0] += 1.0f; E.stats[1] += (float)eps; E.stats[2] += es;
#endif
                es = 0.f; eps = 0;
            }
            FLD(F_EP_SUMS, 0) = es;
            II[(size_t)I_EP_STEPS * n + e] = eps;
            for (int k = 0; k < BG_NUM_REWARD_TERMS; k++) {
                if (C.reward_scale[k] == 0.f) continue;
                float t = FLD(F_EP_SUMS, 1 + k) + term[k];
                if (reset_flag) {
#if defined(__HIP_DEVICE_COMPILE__)
                    atomicAdd(&E.stats[3 + k], t);
#else
                    E.stats[3 + k] += t;
#endif
                    t = 0.f;
                }
                FLD(F_EP_SUMS, 1 + k) = t;
            }
        }
    }
#undef FLD
}

}  // namespace bg
