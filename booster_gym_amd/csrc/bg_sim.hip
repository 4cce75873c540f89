// HIP kernels + C ABI of the simulator/task half of libbooster_gym_amd.so (gfx950 only).
// Lane mapping: 64-thread workgroups = one wavefront = 32 environments, one leg per lane.
// At 4096 envs that is 128 single-wave workgroups, i.e. one wave per CU on 128 of the 256 CUs:
// the kernel is latency-bound per wave, so each wave gets a whole CU (LDS, scalar unit, I-cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "bg_env.h"
#include "bg_dyn_pk.h"
#include "bg_model.h"

using namespace bg;

// ------------------------------------------------------------------ error plumbing
static int fail(int code, const std::string& msg) { return bg_set_error(code, msg.c_str()); }  // (the message lives in bg_model.cpp, host-only code)
#define HIP_OK(expr)                                                                                 \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) return fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)



struct bg_env {
    bg_env_cfg cfg;
    bg_model_desc model;
    int n;
    float* f = nullptr;
    bg_half_bits* h = nullptr;  // fp16 slab of the dynamic state (cfg.state_fp16), else null
    int* rs_counts = nullptr;     // [blocks of RS_BLOCK envs] envs resampled in this step (exact-resampling modes), else null
    unsigned* lowmask = nullptr;  // [blocks of 32 envs] envs a launch's kernel A leaves to its kernel B (two-kernel schemes); null = never needed
    bool body_two_kernel = false; // the non-foot body contacts can occur (body spheres and body_gate_height > terminate_height)
    unsigned* fd_mask = nullptr;  // [blocks of 32 envs] envs the ABA kernel leaves to its second kernel (legs can meet / trunk low)
    int* fd_list = nullptr;       // [n] the same envs as a compact list (aba_compact_kernel)
    unsigned* fd_count = nullptr; // [2] entries in fd_list, finished workgroups of the second kernel
    int32_t* i = nullptr;
    float* stats = nullptr;
    float* curr = nullptr;
    float* curr_read = nullptr;
    int curr_cells = 0;
    ModelDev* model_dev = nullptr;
    PairModel* pair_dev = nullptr;  // the leg constants with the two legs side by side (packed ABA kernel)
    int16_t* hf = nullptr;
    TerrainDev terrain;
    StepOut bound;
    int64_t step_count = 0;
    int zmask = 0;  // Phys::zmask: leg links whose origin lies on the parent's z axis in both legs
    int num_cus = 256;  // of cfg.device (the packed ABA kernel's persistent grid)
    // granular simulator calls (bg_sim_*): caller-owned Isaac-layout tensors + this library's copies of the last actuation / applied forces
    float *sim_root = nullptr, *sim_dof = nullptr, *sim_contact = nullptr, *sim_body = nullptr;
    float *sim_tau = nullptr, *sim_bforce = nullptr, *sim_btorque = nullptr;
    bool sim_wrench_pending = false;
};

// ------------------------------------------------------------------ lane-pair exchange: DPP quad_perm [1,0,3,2]
struct DppSwap {
    __device__ __forceinline__ float swap(float v) {
        int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true);
        return __int_as_float(r);
    }
};

struct LdsSink {
    // the fused env step runs the leg-against-leg narrow phase lane per leg (SELF_INLINE).  The item-parallel LDS form of the ABA kernel needs the whole
    // wave converged at its call site (a ballot, and worker lanes acting for other envs); the env-step kernels call the lane code under lane-divergent
    // branches (trunk-low gate), and it was slower here anyway (108.3 against 104.7 us, HISTORY.md round 4): not selectable
    static constexpr int SELF = SELF_INLINE;
    float* obs; float* priv; int el;
    lds_f32* self_sc; int lane;
    // sweep work space of the fused step: everything in registers.  (Staging the link constants in LDS as forward_dynamics_kernel does
    // removes all spills here too, but the 10 substeps re-read them and the step got 3-5 % slower at every N: measured, not adopted.)
    using Ctx = SubstepCtx;
    __device__ __forceinline__ void bind(Ctx& cx, const LegParams&) const { cx.w.self_sc = self_sc; cx.w.self_lane = lane; }
    __device__ __forceinline__ void put_obs(int k, float v) { obs[el * BG_NUM_OBS + k] = v; }
    __device__ __forceinline__ void put_priv(int k, float v) { priv[el * BG_NUM_PRIV + k] = v; }
};

constexpr int ENVS_PER_BLOCK = 32;
constexpr int BODY_GRID = 512;  // persistent grid of the body-contact kernels (they walk the per-block masks)

// Non-foot body contacts and the two-kernel scheme.  The contact spheres of the trunk box / hip-yaw / shank cylinders matter only while a robot
// is collapsing (trunk lower than cfg.body_gate_height), and ANY trace of their code inside the main kernel costs the common path dearly (a
// runtime branch in the substep loop +30 %, two instantiations of the lane code in one kernel +100 %: register allocation of a 256 + 256
// register kernel does not survive it).  So the work is split by LAUNCH: kernel A (BODY = false, one 32-env block per workgroup) steps every
// env whose trunk is high at the start of the step and records the others in a 32-bit mask per block; kernel B (BODY = true, a small
// persistent grid) walks the masks and steps exactly the recorded envs with the body contacts evaluated.  B normally finds nothing.
__device__ __forceinline__ void copy_out_rows(const StepOut& out, int e0, int rows, unsigned mask, const float* s_obs, const float* s_priv) {
    const int lane = threadIdx.x;
    for (int r = 0; r < rows; r++) {
        if (!((mask >> r) & 1u)) continue;
        if (lane < BG_NUM_OBS) out.obs[(size_t)(e0 + r) * BG_NUM_OBS + lane] = s_obs[r * BG_NUM_OBS + lane];
        if (lane < BG_NUM_PRIV) out.priv[(size_t)(e0 + r) * BG_NUM_PRIV + lane] = s_priv[r * BG_NUM_PRIV + lane];
    }
}

template <bool H16, bool GATED>
__global__ __launch_bounds__(64) void env_step_kernel(EnvDev E, const float* __restrict__ act, uint32_t step, int mode, StepOut out,
                                                      unsigned* __restrict__ lowmask) {
    __shared__ float s_obs[ENVS_PER_BLOCK * BG_NUM_OBS];
    __shared__ float s_priv[ENVS_PER_BLOCK * BG_NUM_PRIV];
    __shared__ float s_self[SELF_LDS_FLOATS];
    __shared__ unsigned s_low;
    const int lane = threadIdx.x;
    const int e0 = blockIdx.x * ENVS_PER_BLOCK;
    int e = e0 + (lane >> 1);
    const bool valid = e < E.n;
    if (!valid) e = E.n - 1;
    if (lane == 0) s_low = 0u;
    __syncthreads();
    DppSwap x;
    LdsSink sink{s_obs, s_priv, lane >> 1, (lds_f32*)s_self, lane};
    // decided once per env step from the trunk height at its start; both lanes of an env agree
    const V3 p0 = v3(E.f[(size_t)(F_ROOT + 0) * E.n + e], E.f[(size_t)(F_ROOT + 1) * E.n + e], E.f[(size_t)(F_ROOT + 2) * E.n + e]);
    const bool low = GATED && mode == 0 && body_contacts_active(make_phys(E.cfg), E.terrain, *E.model, p0);
    if (low) { if (valid && !(lane & 1)) atomicOr(&s_low, 1u << (lane >> 1)); }
    else env_step_lane<DppSwap, LdsSink, H16, false>(E, x, sink, e, lane & 1, valid, act, step, mode, out);
    __syncthreads();
    const int rows = min(ENVS_PER_BLOCK, E.n - e0);
    const unsigned lowm = s_low;
    if (lowm == 0u) {
        // coalesced copy-out of the block's 32 observation rows (contiguous in the [N][47] / [N][14] outputs)
        float* go = out.obs + (size_t)e0 * BG_NUM_OBS;
        for (int k = lane; k < rows * BG_NUM_OBS; k += 64) go[k] = s_obs[k];
        float* gp = out.priv + (size_t)e0 * BG_NUM_PRIV;
        for (int k = lane; k < rows * BG_NUM_PRIV; k += 64) gp[k] = s_priv[k];
    } else {
        copy_out_rows(out, e0, rows, ~lowm, s_obs, s_priv);
    }
    if (GATED && lane == 0) lowmask[blockIdx.x] = lowm;
}

// kernel B: the envs kernel A left out (trunk low at the start of the step), with the non-foot body contacts
template <bool H16>
__global__ __launch_bounds__(64) void env_step_body_kernel(EnvDev E, const float* __restrict__ act, uint32_t step, StepOut out,
                                                           const unsigned* __restrict__ lowmask, int nblocks) {
    __shared__ float s_obs[ENVS_PER_BLOCK * BG_NUM_OBS];
    __shared__ float s_priv[ENVS_PER_BLOCK * BG_NUM_PRIV];
    __shared__ float s_self[SELF_LDS_FLOATS];
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const unsigned lowm = lowmask[b];
        if (lowm == 0u) continue;
        const int e0 = b * ENVS_PER_BLOCK;
        int e = e0 + (lane >> 1);
        const bool valid = e < E.n;
        if (!valid) e = E.n - 1;
        DppSwap x;
        LdsSink sink{s_obs, s_priv, lane >> 1, (lds_f32*)s_self, lane};
        if ((lowm >> (lane >> 1)) & 1u) env_step_lane<DppSwap, LdsSink, H16, true>(E, x, sink, e, lane & 1, valid, act, step, 0, out);
        __syncthreads();
        copy_out_rows(out, e0, min(ENVS_PER_BLOCK, E.n - e0), lowm, s_obs, s_priv);
        __syncthreads();
    }
}

// ------------------------------------------------------------------ reference-exact command resampling (cfg.exact_still_count / same_step_curriculum)
// The two parts of _resample_commands (t1.py:362-389) that couple envs, as two small launches after the env step:
//   resample_count_kernel   counts, per block of RS_BLOCK envs, the envs whose command was resampled in this step (I_RESAMPLED)
//   resample_apply_kernel   position p of every such env in env order (block offsets from the counts + a scan inside the block), K = their number;
//                           curriculum draw from the grid as it is NOW, i.e. after this step's resets have updated it (t1.py:305 precedes :365);
//                           the env stands still iff keyed_perm(p) < int(still_proportion * K): exactly that many, a random subset (t1.py:381-383);
//                           commands / gait frequency are written to the state and to entries 6..10 of the env's observation row.
constexpr int RS_BLOCK = 256;
__global__ __launch_bounds__(RS_BLOCK) void resample_count_kernel(EnvDev E, int* __restrict__ counts) {
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const int e = blockIdx.x * RS_BLOCK + threadIdx.x;
    const int flag = e < E.n ? E.i[(size_t)I_RESAMPLED * E.n + e] : 0;
    if (flag) atomicAdd(&s_cnt, 1);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_cnt;
}
// counts [nblocks] -> exclusive prefix offsets in place, total in counts[nblocks]; one workgroup (256 threads, each a contiguous chunk), so that
// resample_apply_kernel reads two numbers per block instead of every block summing all counts (O(nblocks^2) loads at 1 M envs)
__global__ __launch_bounds__(RS_BLOCK) void resample_offsets_kernel(int* __restrict__ counts, int nblocks) {
    __shared__ int s_part[RS_BLOCK];
    const int per = (nblocks + RS_BLOCK - 1) / RS_BLOCK, b0 = threadIdx.x * per, b1 = min(nblocks, b0 + per);
    int sum = 0;
    for (int b = b0; b < b1; b++) sum += counts[b];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < RS_BLOCK; d <<= 1) {
        const int v = (int)threadIdx.x >= d ? s_part[threadIdx.x - d] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = s_part[threadIdx.x] - sum;
    for (int b = b0; b < b1; b++) { const int c = counts[b]; counts[b] = run; run += c; }
    if (threadIdx.x == RS_BLOCK - 1) counts[nblocks] = s_part[RS_BLOCK - 1];
}
__global__ __launch_bounds__(RS_BLOCK) void resample_apply_kernel(EnvDev E, const int* __restrict__ counts, int nblocks, uint32_t step, int mode, float* __restrict__ obs) {
    __shared__ int s_scan[RS_BLOCK];
    __shared__ int s_off, s_total;
    const bg_env_cfg& C = E.cfg;
    const int n = E.n, e = blockIdx.x * RS_BLOCK + threadIdx.x;
    const int flag = e < n ? E.i[(size_t)I_RESAMPLED * n + e] : 0;
    if (threadIdx.x == 0) { s_off = counts[blockIdx.x]; s_total = counts[nblocks]; }  // exclusive offsets + total (resample_offsets_kernel)
    s_scan[threadIdx.x] = flag;
    __syncthreads();
    for (int d = 1; d < RS_BLOCK; d <<= 1) {  // inclusive scan of the flags
        const int v = (int)threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    if (!flag) return;
    const uint32_t so = mode ? 64u : 0u;
    const uint32_t K = (uint32_t)s_total, p = (uint32_t)(s_off + s_scan[threadIdx.x] - 1);
    float cmd[3] = {E.f[(size_t)(F_CMD + 0) * n + e], E.f[(size_t)(F_CMD + 1) * n + e], E.f[(size_t)(F_CMD + 2) * n + e]};
    float gait_f = E.f[(size_t)F_GAIT_F * n + e];
    if (C.curriculum && C.same_step_curriculum) {
        Rand4 cr = rand4(C.seed, (uint32_t)e, step, so + RS_CURR);
        int lin_level, ang_level;
        curriculum_draw(C, E.curr, cr, cmd, &lin_level, &ang_level);
        E.i[(size_t)I_CURR_LIN * n + e] = lin_level; E.i[(size_t)I_CURR_ANG * n + e] = ang_level;
    }
    bool still;
    if (C.exact_still_count) {
        const uint32_t m = (uint32_t)((double)C.still_proportion * (double)K);  // int(still_proportion * len(env_ids)), t1.py:381
        const uint32_t key = mix32((uint32_t)C.seed ^ mix32(step + 0x632BE5ABu * (so + 1u)));
        still = keyed_perm(p, K, key) < m;
    } else {
        still = rand4(C.seed, (uint32_t)e, step, so + RS_CMD1).u[0] < C.still_proportion;
    }
    if (still) { cmd[0] = cmd[1] = cmd[2] = 0.f; gait_f = 0.f; }
    for (int a = 0; a < 3; a++) E.f[(size_t)(F_CMD + a) * n + e] = cmd[a];
    E.f[(size_t)F_GAIT_F * n + e] = gait_f;
    float* o = obs + (size_t)e * BG_NUM_OBS;  // t1.py:584-586
    o[6] = cmd[0] * C.norm_lin_vel; o[7] = cmd[1] * C.norm_lin_vel; o[8] = cmd[2] * C.norm_ang_vel;
    float sg, cg;
    bg_sincos(6.28318530717959f * E.f[(size_t)F_GAIT_P * n + e], &sg, &cg);
    const float on = gait_f > 1.0e-8f ? 1.f : 0.f;
    o[9] = cg * on; o[10] = sg * on;
}

// ------------------------------------------------------------------ dynamics only: qacc for N independent states
// One substep's accelerations per launch.  This is the kernel the ABA roofline is quoted on, and the regime that matters for it is a FULL
// chip (>= 65k envs), where throughput = VALU issue slots: a wave64 fp32 instruction holds its SIMD for 4 cycles and the kernel is ~3,400 of
// them per wave.  With everything in registers the kernel needs 282 and one wave fits per SIMD (77 % VALU-busy, the rest is exposed load
// and dependency latency).  Keeping the per-env link constants (13 floats x 6 links per lane, computed once per launch) in LDS brings it to
// 248 registers with no spill, two waves share a SIMD and cover each other's stalls: 88 % VALU-busy, +14 % throughput at 1M envs.
template <bool BODY, int SELF>
__device__ __forceinline__ bool forward_dynamics_lane(const EnvDev& E, int lane, int e, bool valid, float* s_work, const float* __restrict__ root,
                                                      const float* __restrict__ q, const float* __restrict__ qd, const float* __restrict__ tau,
                                                      const float* __restrict__ wrench, float* __restrict__ qacc) {
    const int leg = lane & 1;
    const int n = E.n;
    BG_PHASE("load_state_and_link_constants");
    Phys ph = make_phys(E.cfg);
    ph.zmask = E.zmask;
    ContactCfg cc = make_contact_cfg(E.cfg);
    BaseState bs;
    const float* r = root + (size_t)e * 13;
    bs.pos = v3(r[0], r[1], r[2]);
    for (int a = 0; a < 4; a++) bs.quat[a] = r[3 + a];
    bs.vlin = v3(r[7], r[8], r[9]); bs.vang = v3(r[10], r[11], r[12]);
    LegState ls;
    float t6[LEG_LINKS];
    for (int i = 0; i < LEG_LINKS; i++) {
        ls.q[i] = q[(size_t)e * 12 + leg * 6 + i]; ls.qd[i] = qd[(size_t)e * 12 + leg * 6 + i]; t6[i] = tau[(size_t)e * 12 + leg * 6 + i];
    }
    LegParams lp;
    load_leg_params(*E.model, cc, leg, e, n, E.f + (size_t)F_MASS_SCALE * n, E.f + (size_t)F_COM_OFF * n, E.f + (size_t)F_FOOT_MAT * n, lp);
    LinkConst bk = load_base_link(*E.model, e, n, E.f + (size_t)F_MASS_SCALE * n, E.f + (size_t)F_COM_OFF * n);
    SV wr = sv_zero();
    if (wrench) { const float* w = wrench + (size_t)e * 6; wr.l = v3(w[0], w[1], w[2]); wr.a = v3(w[3], w[4], w[5]); }
    DppSwap x;
    SubstepCtxLdsLink cx;
    cx.w.st.bind((lds_f32*)s_work, lane);
    cx.w.st.stash(lp);
    cx.w.self_sc = LdsLinkStore::self_scratch((lds_f32*)s_work); cx.w.self_lane = lane;
    BaseContribution mine = substep_pre<BODY, SELF>(ph, E.terrain, *E.model, leg, lp, ls, t6, bs, cx, x), both;
    if constexpr (SELF == SELF_DEFER) { if (cx.w.self_deferred) return true; }  // the legs can meet: the second kernel's env (both lanes agree)
    bg_pin(mine.I); bg_pin(mine.p);
    BG_PHASE("pair_exchange");
    for (int k = 0; k < 6; k++) { both.I.A.e[k] = mine.I.A.e[k] + x.swap(mine.I.A.e[k]); both.I.M.e[k] = mine.I.M.e[k] + x.swap(mine.I.M.e[k]); }
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) both.I.H.e[a][b] = mine.I.H.e[a][b] + x.swap(mine.I.H.e[a][b]);
    for (int k = 0; k < 3; k++) { both.p.a.e[k] = mine.p.a.e[k] + x.swap(mine.p.a.e[k]); both.p.l.e[k] = mine.p.l.e[k] + x.swap(mine.p.l.e[k]); }
    float qdd[LEG_LINKS];
    V3 lin_w, ang_w, fw;
    substep_solve(ph, bk, lp, ls, cx, both, wr, qdd, &lin_w, &ang_w, &fw);
    bg_pin(lin_w); bg_pin(ang_w); bg_pin(fw);
    BG_PHASE("store");
    if (!valid) return false;
    float* o = qacc + (size_t)e * 18;
    if (leg == 0) for (int a = 0; a < 3; a++) { o[a] = lin_w.e[a]; o[3 + a] = ang_w.e[a]; }
    for (int i = 0; i < LEG_LINKS; i++) o[6 + leg * 6 + i] = qdd[i];
    for (int a = 0; a < 3; a++) E.f[(size_t)(F_CONTACT + 3 * leg + a) * n + e] = fw.e[a];
    return false;
}
// The ABA kernel.  Leg-against-leg contacts: the clearance test per lane, the narrow phase item-parallel through LDS inside this kernel
// (SELF_LDS, bg_dyn.h:self_narrow_phase_lds) -- one launch, no second pass over the envs whose legs are close.  (Rounds 2-3 left those envs to a
// second kernel that re-read their scattered state: 79 us and 3.2 x their bytes for 9 % of the envs.)
// GATED (only when the non-foot body contacts can occur, i.e. contact.body_gate_height > rewards.terminate_height): an env whose trunk is low is
// LEFT to kernel B, which carries the body-contact code; there kernel A also leaves the envs whose legs can meet (SELF_DEFER), records both in a
// 32-bit mask per block, aba_compact_kernel turns the masks into a compact env list and kernel B walks it.  left_mask is null otherwise.
template <bool GATED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void forward_dynamics_kernel(EnvDev E, const float* __restrict__ root, const float* __restrict__ q,
                                                              const float* __restrict__ qd, const float* __restrict__ tau,
                                                              const float* __restrict__ wrench, float* __restrict__ qacc,
                                                              unsigned* __restrict__ left_mask) {
    __shared__ float s_work[LdsLinkStore::FLOATS];
    const int lane = threadIdx.x;
    LdsLinkStore::write_origins((lds_f32*)s_work, *E.model, lane);
    // (A persistent form -- 2,048 workgroups walking the blocks, per-lane constants kept from being hoisted -- was measured on one box against this
    // one-workgroup-per-block form: 263 / 243 us against 257 / 243-248 us for the two bench states.  The wave slots' turnover is not what the launch waits for.)
    int e = blockIdx.x * ENVS_PER_BLOCK + (lane >> 1);
    const bool valid = e < E.n;
    if (!valid) e = E.n - 1;
    bool left = false;
    if constexpr (GATED) {
        const float* r = root + (size_t)e * 13;
        left = body_contacts_active(make_phys(E.cfg), E.terrain, *E.model, v3(r[0], r[1], r[2]));
    }
    if constexpr (GATED) { if (!left) left = forward_dynamics_lane<false, SELF_DEFER>(E, lane, e, valid, s_work, root, q, qd, tau, wrench, qacc); }
    else forward_dynamics_lane<false, SELF_LDS>(E, lane, e, valid, s_work, root, q, qd, tau, wrench, qacc);
    if (GATED && left_mask) {
        const unsigned long long m = __ballot(left && valid && !(lane & 1));  // bit 2k = env k of the block
        if (lane == 0) {
            unsigned packed = 0u;
            for (int k = 0; k < ENVS_PER_BLOCK; k++) packed |= (unsigned)((m >> (2 * k)) & 1ull) << k;
            left_mask[blockIdx.x] = packed;
        }
    }
}
// The packed ABA kernel (round 5): one ENV per lane, its two legs in the halves of 64-bit register pairs (bg_dyn_pk.h), 64 envs per one-wave
// workgroup.  The sweeps issue v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 for both legs at once, the trunk is computed once per env, the legs'
// contributions meet without a lane exchange.  ~2 x the registers of the lane-per-leg form, so ONE wave per SIMD.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void forward_dynamics_pk_kernel(
    EnvDev E, const PairModel* __restrict__ pm, const float* __restrict__ root, const float* __restrict__ q, const float* __restrict__ qd,
    const float* __restrict__ tau, const float* __restrict__ wrench, float* __restrict__ qacc) {
    __shared__ float s_self[SelfPk::END];
    const int lane = threadIdx.x, n = E.n;
    int e = blockIdx.x * 64 + lane;
    const bool valid = e < n;
    if (!valid) e = n - 1;
    BG_PHASE("load_inputs");
    Phys ph = make_phys(E.cfg);
    ph.zmask = E.zmask;
    const ContactCfg cc = make_contact_cfg(E.cfg);
    PkCtx cx;
    PkInputs& in = cx.w.st.in;
    {
        const float* r = root + (size_t)e * 13;
#pragma unroll
        for (int k = 0; k < 13; k++) in.v[PkSlots::ROOT + k] = r[k];
        // [N][12] rows are 48 bytes: three 16-byte loads per array
        const float4* q4 = (const float4*)(q + (size_t)e * 12);
        const float4* d4 = (const float4*)(qd + (size_t)e * 12);
        const float4* t4 = (const float4*)(tau + (size_t)e * 12);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float4 a = q4[k], b = d4[k], c = t4[k];
            in.v[PkSlots::Q + 4 * k] = a.x; in.v[PkSlots::Q + 4 * k + 1] = a.y; in.v[PkSlots::Q + 4 * k + 2] = a.z; in.v[PkSlots::Q + 4 * k + 3] = a.w;
            in.v[PkSlots::QD + 4 * k] = b.x; in.v[PkSlots::QD + 4 * k + 1] = b.y; in.v[PkSlots::QD + 4 * k + 2] = b.z; in.v[PkSlots::QD + 4 * k + 3] = b.w;
            in.v[PkSlots::TAU + 4 * k] = c.x; in.v[PkSlots::TAU + 4 * k + 1] = c.y; in.v[PkSlots::TAU + 4 * k + 2] = c.z; in.v[PkSlots::TAU + 4 * k + 3] = c.w;
        }
        in.has_wrench = wrench != nullptr;
#pragma unroll
        for (int k = 0; k < 6; k++) in.v[PkSlots::WRENCH + k] = wrench ? wrench[(size_t)e * 6 + k] : 0.f;
        const float* par = E.f + (size_t)F_MASS_SCALE * n + e;   // the 58 per-env parameter fields, n floats apart
#pragma unroll
        for (int k = 0; k < 58; k++) in.v[PkSlots::MS + k] = par[(size_t)k * n];
    }
    cx.w.st.pm = pm;
    cx.w.self_sc = (lds_f32*)s_self; cx.w.self_lane = lane;
    f2 qdd[LEG_LINKS];
    V3 lin_w, ang_w;
    V3T<f2> fw;
    pk_forward_env(ph, cc, E.terrain, *E.model, cx, qdd, &lin_w, &ang_w, &fw);
    bg_pin(lin_w); bg_pin(ang_w); bg_pin(fw);
    BG_PHASE("store");
    if (!valid) return;
    float* o = qacc + (size_t)e * 18;
#pragma unroll
    for (int a = 0; a < 3; a++) { o[a] = lin_w.e[a]; o[3 + a] = ang_w.e[a]; }
#pragma unroll
    for (int i = 0; i < LEG_LINKS; i++) { o[6 + i] = qdd[i][0]; o[12 + i] = qdd[i][1]; }
#pragma unroll
    for (int a = 0; a < 3; a++) { E.f[(size_t)(F_CONTACT + a) * n + e] = fw.e[a][0]; E.f[(size_t)(F_CONTACT + 3 + a) * n + e] = fw.e[a][1]; }
}
// masks of kernel A -> compact list of env indices (count in left_count[0]); one atomic per 256 blocks
__global__ __launch_bounds__(256) void aba_compact_kernel(const unsigned* __restrict__ left_mask, int nblocks, int* __restrict__ left_list, unsigned* __restrict__ left_count) {
    __shared__ unsigned s_scan[256];
    __shared__ unsigned s_base;
    const int b = blockIdx.x * 256 + threadIdx.x;
    const unsigned m = b < nblocks ? left_mask[b] : 0u;
    const unsigned c = (unsigned)__popc(m);
    s_scan[threadIdx.x] = c;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned v = (int)threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0u;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    if (threadIdx.x == 255) s_base = s_scan[255] ? atomicAdd(left_count, s_scan[255]) : 0u;
    __syncthreads();
    unsigned o = s_base + s_scan[threadIdx.x] - c, mm = m;
    while (mm) {
        const int k = __ffs((int)mm) - 1;
        mm &= mm - 1u;
        left_list[o++] = b * ENVS_PER_BLOCK + k;
    }
}
// kernel B of the ABA launch: the envs kernel A left, 32 per wave from the compact list, with the leg-against-leg narrow phase and (trunk low)
// the non-foot body contacts.  The last workgroup to finish empties the list for the next launch.  left_count: [0] entries, [1] finished workgroups.
template <bool GATED>  // GATED: some of the listed envs may have their trunk low (non-foot body contacts); otherwise all of them are there for their legs
__global__ __launch_bounds__(64) void forward_dynamics_body_kernel(EnvDev E, const float* __restrict__ root, const float* __restrict__ q,
                                                                   const float* __restrict__ qd, const float* __restrict__ tau,
                                                                   const float* __restrict__ wrench, float* __restrict__ qacc,
                                                                   const int* __restrict__ left_list, unsigned* __restrict__ left_count) {
    __shared__ float s_work[LdsLinkStore::FLOATS];
    const int lane = threadIdx.x;
    LdsLinkStore::write_origins((lds_f32*)s_work, *E.model, lane);
    const int cnt = (int)__builtin_nontemporal_load(left_count);
    for (int w = blockIdx.x; w * ENVS_PER_BLOCK < cnt; w += gridDim.x) {
        const int idx = w * ENVS_PER_BLOCK + (lane >> 1);
        const bool valid = idx < cnt;
        const int e = left_list[valid ? idx : cnt - 1];
        // the body spheres only for an env whose trunk is low (the same rule as kernel A's and the oracle's)
        const float* r = root + (size_t)e * 13;
        bool low = false;
        if constexpr (GATED) low = body_contacts_active(make_phys(E.cfg), E.terrain, *E.model, v3(r[0], r[1], r[2]));
        if (low) { if constexpr (GATED) forward_dynamics_lane<true, SELF_INLINE>(E, lane, e, valid, s_work, root, q, qd, tau, wrench, qacc); }
        else forward_dynamics_lane<false, SELF_INLINE>(E, lane, e, valid, s_work, root, q, qd, tau, wrench, qacc);
    }
    __syncthreads();
    if (lane == 0 && atomicAdd(left_count + 1, 1u) == gridDim.x - 1u) { left_count[0] = 0u; left_count[1] = 0u; }
}

// ------------------------------------------------------------------ granular simulator calls on caller-owned Isaac-layout tensors
struct SimLane {
    BaseState bs;
    LegState ls;
    LegParams lp;
    LinkConst bk;
};
__device__ __forceinline__ void sim_load(const EnvDev& E, int e, int leg, const float* __restrict__ root, const float* __restrict__ dof, SimLane& L) {
    const int n = E.n;
    const float* r = root + (size_t)e * 13;
    L.bs.pos = v3(r[0], r[1], r[2]);
    for (int a = 0; a < 4; a++) L.bs.quat[a] = r[3 + a];
    L.bs.vlin = v3(r[7], r[8], r[9]); L.bs.vang = v3(r[10], r[11], r[12]);
    for (int i = 0; i < LEG_LINKS; i++) {
        const float* d = dof + ((size_t)e * 12 + leg * 6 + i) * 2;
        L.ls.q[i] = d[0]; L.ls.qd[i] = d[1];
    }
    ContactCfg cc = make_contact_cfg(E.cfg);
    load_leg_params(*E.model, cc, leg, e, n, E.f + (size_t)F_MASS_SCALE * n, E.f + (size_t)F_COM_OFF * n, E.f + (size_t)F_FOOT_MAT * n, L.lp);
    L.bk = load_base_link(*E.model, e, n, E.f + (size_t)F_MASS_SCALE * n, E.f + (size_t)F_COM_OFF * n);
}
// rigid-body state rows [13][13] of env e: the trunk row (leg-0 lane) and this lane's six links
__device__ __forceinline__ void sim_write_body(const SimLane& L, int e, int leg, float* __restrict__ body) {
    float* b = body + (size_t)e * 13 * 13;
    if (leg == 0) {
        for (int a = 0; a < 3; a++) { b[a] = L.bs.pos.e[a]; b[7 + a] = L.bs.vlin.e[a]; b[10 + a] = L.bs.vang.e[a]; }
        const float sg = L.bs.quat[3] < 0.f ? -1.f : 1.f;
        for (int a = 0; a < 4; a++) b[3 + a] = sg * L.bs.quat[a];
    }
    M3 R0 = quat_to_mat(L.bs.quat);
    leg_body_states<0>(L.lp, L.ls, base_body_velocity(R0, L.bs), R0, L.bs.pos, b + 13 * (1 + leg * LEG_LINKS));
}

// one gym.simulate (t1.py:451): a sim.dt step of every env, in place on root [N][13] / dof [N][12][2]
template <bool BODY>
__device__ __forceinline__ void sim_substep_lane(const EnvDev& E, int e, bool valid, float* __restrict__ root, float* __restrict__ dof,
                                                 const float* __restrict__ tau, const float* __restrict__ bforce, const float* __restrict__ btorque,
                                                 float* __restrict__ contact, float* __restrict__ body) {
    const int lane = threadIdx.x, leg = lane & 1;
    Phys ph = make_phys(E.cfg);
    ph.zmask = E.zmask;
    SimLane L;
    sim_load(E, e, leg, root, dof, L);
    float t6[LEG_LINKS];
    for (int i = 0; i < LEG_LINKS; i++) t6[i] = tau[(size_t)e * 12 + leg * 6 + i];
    SV wr = sv_zero();
    SV fext[LEG_LINKS];
    const bool has_w = bforce != nullptr;
    if (has_w) {
        const float* f = bforce + (size_t)e * 39;
        const float* t = btorque + (size_t)e * 39;
        wr = local_wrench_at_com(L.bk, v3(f[0], f[1], f[2]), v3(t[0], t[1], t[2]));
        for (int i = 0; i < LEG_LINKS; i++) {
            const int b = 3 * (1 + leg * LEG_LINKS + i);
            fext[i] = local_wrench_at_com(L.lp.lk[i], v3(f[b], f[b + 1], f[b + 2]), v3(t[b], t[b + 1], t[b + 2]));
        }
    }
    DppSwap x;
    SubstepCtx cx;
    BodyContactOut bo;
    BaseContribution mine = substep_pre<BODY>(ph, E.terrain, *E.model, leg, L.lp, L.ls, t6, L.bs, cx, x, has_w ? fext : nullptr, &bo), both;
    for (int k = 0; k < 6; k++) { both.I.A.e[k] = mine.I.A.e[k] + x.swap(mine.I.A.e[k]); both.I.M.e[k] = mine.I.M.e[k] + x.swap(mine.I.M.e[k]); }
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) both.I.H.e[a][b] = mine.I.H.e[a][b] + x.swap(mine.I.H.e[a][b]);
    for (int k = 0; k < 3; k++) { both.p.a.e[k] = mine.p.a.e[k] + x.swap(mine.p.a.e[k]); both.p.l.e[k] = mine.p.l.e[k] + x.swap(mine.p.l.e[k]); }
    float qdd[LEG_LINKS];
    V3 lin_w, ang_w, fw;
    substep_solve(ph, L.bk, L.lp, L.ls, cx, both, wr, qdd, &lin_w, &ang_w, &fw);
    substep_integrate(ph, L.lp, L.ls, L.bs, qdd, lin_w, ang_w);
    if (!valid) return;
    if (leg == 0) {
        float* r = root + (size_t)e * 13;
        for (int a = 0; a < 3; a++) { r[a] = L.bs.pos.e[a]; r[7 + a] = L.bs.vlin.e[a]; r[10 + a] = L.bs.vang.e[a]; }
        for (int a = 0; a < 4; a++) r[3 + a] = L.bs.quat[a];
    }
    for (int i = 0; i < LEG_LINKS; i++) {
        float* d = dof + ((size_t)e * 12 + leg * 6 + i) * 2;
        d[0] = L.ls.q[i]; d[1] = L.ls.qd[i];
    }
    if (contact) {  // net contact force per body, world frame (t1.py:219): sole corners of the feet + contact spheres of the other shapes
        float* c = contact + (size_t)e * 39;
        V3 tf = bo.trunk;
        for (int a = 0; a < 3; a++) tf.e[a] += x.swap(tf.e[a]);
        if (leg == 0) for (int a = 0; a < 3; a++) c[a] = tf.e[a];
        for (int i = 0; i < LEG_LINKS - 1; i++)
            for (int a = 0; a < 3; a++) c[3 * (1 + leg * LEG_LINKS + i) + a] = bo.link[i].e[a];
        for (int a = 0; a < 3; a++) c[3 * (LEG_LINKS + leg * LEG_LINKS) + a] = fw.e[a];
    }
    if (body) sim_write_body(L, e, leg, body);
}
__global__ __launch_bounds__(64) void sim_substep_kernel(EnvDev E, float* __restrict__ root, float* __restrict__ dof, const float* __restrict__ tau,
                                                         const float* __restrict__ bforce, const float* __restrict__ btorque,
                                                         float* __restrict__ contact, float* __restrict__ body, unsigned* __restrict__ lowmask) {
    __shared__ unsigned s_low;
    const int lane = threadIdx.x;
    int e = blockIdx.x * ENVS_PER_BLOCK + (lane >> 1);
    const bool valid = e < E.n;
    if (!valid) e = E.n - 1;
    if (lowmask) {
        if (lane == 0) s_low = 0u;
        __syncthreads();
    }
    const float* r = root + (size_t)e * 13;
    const bool low = lowmask && body_contacts_active(make_phys(E.cfg), E.terrain, *E.model, v3(r[0], r[1], r[2]));
    if (low) { if (valid && !(lane & 1)) atomicOr(&s_low, 1u << (lane >> 1)); }
    else sim_substep_lane<false>(E, e, valid, root, dof, tau, bforce, btorque, contact, body);
    if (lowmask) {
        __syncthreads();
        if (lane == 0) lowmask[blockIdx.x] = s_low;
    }
}
__global__ __launch_bounds__(64) void sim_substep_body_kernel(EnvDev E, float* __restrict__ root, float* __restrict__ dof, const float* __restrict__ tau,
                                                              const float* __restrict__ bforce, const float* __restrict__ btorque,
                                                              float* __restrict__ contact, float* __restrict__ body,
                                                              const unsigned* __restrict__ lowmask, int nblocks) {
    const int lane = threadIdx.x;
    for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const unsigned lowm = lowmask[b];
        if (lowm == 0u) continue;
        int e = b * ENVS_PER_BLOCK + (lane >> 1);
        const bool valid = e < E.n;
        if (!valid) e = E.n - 1;
        if ((lowm >> (lane >> 1)) & 1u) sim_substep_lane<true>(E, e, valid, root, dof, tau, bforce, btorque, contact, body);
    }
}

// gym.refresh_rigid_body_state_tensor (t1.py:462): body rows from the current root / dof tensors, no dynamics
__global__ __launch_bounds__(64) void sim_body_state_kernel(EnvDev E, const float* __restrict__ root, const float* __restrict__ dof, float* __restrict__ body) {
    const int lane = threadIdx.x, leg = lane & 1;
    const int e = blockIdx.x * ENVS_PER_BLOCK + (lane >> 1);
    if (e >= E.n) return;
    SimLane L;
    sim_load(E, e, leg, root, dof, L);
    sim_write_body(L, e, leg, body);
}

// ------------------------------------------------------------------ layout conversion helpers
// per-env float field `field` (+ component c) of env e, wherever it is stored (fp32 slab, or the fp16 slab of a state_fp16 env)
__device__ __forceinline__ float ld_field(const EnvDev& E, int field, int e) {
    const size_t o = (size_t)field * E.n + e;
    return (E.h && field < FP16_SLAB_FIELDS && fp16_state_field(field)) ? half_bits_to_float(E.h[o]) : E.f[o];
}
__device__ __forceinline__ void st_field(const EnvDev& E, int field, int e, float v) {
    const size_t o = (size_t)field * E.n + e;
    if (E.h && field < FP16_SLAB_FIELDS && fp16_state_field(field)) E.h[o] = float_to_half_bits(v);
    else E.f[o] = v;
}
__global__ void soa_to_aos_kernel(EnvDev E, int field, int is_int, float* __restrict__ aos, int comps) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E.n * comps) return;
    int e = idx / comps, c = idx % comps;
    if (is_int) reinterpret_cast<int32_t*>(aos)[idx] = E.i[(size_t)(field + c) * E.n + e];
    else aos[idx] = ld_field(E, field + c, e);
}
__global__ void aos_to_soa_kernel(EnvDev E, int field, int is_int, const float* __restrict__ aos, int comps) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= E.n * comps) return;
    int e = idx / comps, c = idx % comps;
    if (is_int) E.i[(size_t)(field + c) * E.n + e] = reinterpret_cast<const int32_t*>(aos)[idx];
    else st_field(E, field + c, e, aos[idx]);
}
__global__ void get_state_kernel(EnvDev E, float* root, float* dof, float* contact) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E.n) return;
    const int n = E.n;
    if (root) for (int c = 0; c < 13; c++) root[(size_t)e * 13 + c] = ld_field(E, F_ROOT + c, e);
    if (dof) for (int j = 0; j < 12; j++) {
        dof[((size_t)e * 12 + j) * 2] = ld_field(E, F_Q + j, e);
        dof[((size_t)e * 12 + j) * 2 + 1] = ld_field(E, F_QD + j, e);
    }
    if (contact) {
        for (int c = 0; c < 39; c++) contact[(size_t)e * 39 + c] = 0.f;
        for (int a = 0; a < 3; a++) {
            contact[(size_t)e * 39 + 6 * 3 + a] = E.f[(size_t)(F_CONTACT + a) * n + e];
            contact[(size_t)e * 39 + 12 * 3 + a] = E.f[(size_t)(F_CONTACT + 3 + a) * n + e];
        }
    }
}
__global__ void set_state_kernel(EnvDev E, const float* root, const float* dof) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E.n) return;
    if (root) for (int c = 0; c < 13; c++) st_field(E, F_ROOT + c, e, root[(size_t)e * 13 + c]);
    if (dof) for (int j = 0; j < 12; j++) {
        st_field(E, F_Q + j, e, dof[((size_t)e * 12 + j) * 2]);
        st_field(E, F_QD + j, e, dof[((size_t)e * 12 + j) * 2 + 1]);
    }
}

// ------------------------------------------------------------------ ABI: model


// ------------------------------------------------------------------ ABI: env
static EnvDev env_dev(const bg_env* e) {
    EnvDev E;
    E.zmask = e->zmask;
    E.f = e->f; E.h = e->h; E.i = e->i; E.stats = e->stats; E.curr = e->curr; E.curr_read = e->curr_read; E.model = e->model_dev; E.terrain = e->terrain; E.cfg = e->cfg; E.n = e->n;
    return E;
}

extern "C" void bg_env_destroy(bg_env* e);
// allocations and uploads of bg_env_create; on failure the caller destroys whatever was allocated so far
static int env_create_fill(bg_env* e, const bg_env_cfg* cfg, const bg_model* model) {
    e->cfg = *cfg; e->model = model->desc; e->n = cfg->num_envs;
    const size_t n = (size_t)e->n;
    HIP_OK(hipMalloc(&e->f, sizeof(float) * n * F_COUNT));
    HIP_OK(hipMalloc(&e->i, sizeof(int32_t) * n * I_COUNT));
    HIP_OK(hipMalloc(&e->stats, sizeof(float) * STATS_COUNT));
    HIP_OK(hipMalloc(&e->model_dev, sizeof(ModelDev)));
    e->curr_cells = (2 * cfg->lin_vel_levels + 1) * (2 * cfg->ang_vel_levels + 1);
    HIP_OK(hipMalloc(&e->curr, sizeof(float) * e->curr_cells));
    HIP_OK(hipMalloc(&e->curr_read, sizeof(float) * e->curr_cells));
    if (cfg->exact_still_count || (cfg->curriculum && cfg->same_step_curriculum)) {
        if (cfg->state_fp16) return fail(-4, "bg_env_create: exact_still_count / same_step_curriculum are not available with state_fp16");
        HIP_OK(hipMalloc(&e->rs_counts, sizeof(int) * ((n + 255) / 256 + 1)));
    }
    {   // t1.py:249-255: all mass on the centre cell
        std::vector<float> c0(e->curr_cells, 0.f);
        c0[cfg->lin_vel_levels * (2 * cfg->ang_vel_levels + 1) + cfg->ang_vel_levels] = 1.f;
        HIP_OK(hipMemcpy(e->curr, c0.data(), sizeof(float) * e->curr_cells, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(e->curr_read, c0.data(), sizeof(float) * e->curr_cells, hipMemcpyHostToDevice));
    }
    HIP_OK(hipMemset(e->f, 0, sizeof(float) * n * F_COUNT));
    HIP_OK(hipMemset(e->i, 0, sizeof(int32_t) * n * I_COUNT));
    HIP_OK(hipMemset(e->stats, 0, sizeof(float) * STATS_COUNT));
    ModelDev md;
    memset(&md, 0, sizeof(md));
    for (int b = 0; b < BG_NUM_BODIES; b++) {
        md.mass[b] = model->desc.mass[b];
        for (int a = 0; a < 3; a++) { md.pos[b][a] = model->desc.body_pos[b][a]; md.com[b][a] = model->desc.com[b][a]; }
        for (int a = 0; a < 6; a++) md.inertia[b][a] = model->desc.inertia[b][a];
    }
    for (int j = 0; j < BG_NUM_DOFS; j++) {
        md.q_lo[j] = model->desc.dof_lower[j]; md.q_hi[j] = model->desc.dof_upper[j];
        md.qd_max[j] = model->desc.dof_velocity[j]; md.tau_lim[j] = model->desc.dof_effort[j];
    }
    for (int k = 0; k < 4; k++) for (int a = 0; a < 3; a++) md.corner[k][a] = model->desc.feet_edge_pos[k][a];
    md.sph_n = model->desc.num_body_spheres;
    for (int k = 0; k < md.sph_n; k++) {
        const int b = model->desc.sphere_body[k];
        if (md.sph_cnt[b]++ == 0) md.sph_first[b] = k;
        md.sph_r[k] = model->desc.sphere_radius[k];
        for (int a = 0; a < 3; a++) md.sph_pos[k][a] = model->desc.sphere_pos[k][a];
    }
    e->zmask = 0;
    for (int i = 0; i < 6; i++) {
        bool z = true;
        for (int leg = 0; leg < 2; leg++) z = z && model->desc.body_pos[1 + 6 * leg + i][0] == 0.f && model->desc.body_pos[1 + 6 * leg + i][1] == 0.f;
        if (z) e->zmask |= 1 << i;
    }
    bool caps = true;
    for (int leg = 0; leg < 2; leg++)
        for (int k = 0; k < 2; k++) {
            const int ax = k == 0 ? 2 : 0;
            for (int a = 0; a < 3; a++) md.cap_c[leg][k][a] = 0.5f * (model->desc.self_capsule_a[leg][k][a] + model->desc.self_capsule_b[leg][k][a]);
            md.cap_h[leg][k] = 0.5f * (model->desc.self_capsule_b[leg][k][ax] - model->desc.self_capsule_a[leg][k][ax]);
            md.cap_r[leg][k] = model->desc.self_capsule_r[leg][k];
            caps = caps && md.cap_r[leg][k] > 0.f;
        }
    if (!caps) e->cfg.self_collisions = 0;  // a model without self-collision geometry
    HIP_OK(hipMemcpy(e->model_dev, &md, sizeof(md), hipMemcpyHostToDevice));
    {
        hipDeviceProp_t prop;
        HIP_OK(hipGetDeviceProperties(&prop, cfg->device));
        e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const PairModel pmh = make_pair_model(md);
    HIP_OK(hipMalloc(&e->pair_dev, sizeof(PairModel)));
    HIP_OK(hipMemcpy(e->pair_dev, &pmh, sizeof(pmh), hipMemcpyHostToDevice));
    e->terrain.type = 0; e->terrain.rows = e->terrain.cols = e->terrain.border_px = 0; e->terrain.inv_hscale = 1.f; e->terrain.vscale = 0.f; e->terrain.hf = nullptr;
    memset(&e->bound, 0, sizeof(e->bound));
    // defaults: unit mass scale / compliance, identity orientation, nominal friction
    std::vector<float> host(n * F_COUNT, 0.f);
    for (size_t k = 0; k < n; k++) {
        host[(size_t)(F_ROOT + 6) * n + k] = 1.f;
        for (int b = 0; b < 13; b++) host[(size_t)(F_MASS_SCALE + b) * n + k] = 1.f;
        for (int f = 0; f < 2; f++) { host[(size_t)(F_FOOT_MAT + 3 * f) * n + k] = 1.f; host[(size_t)(F_FOOT_MAT + 3 * f + 1) * n + k] = 1.f; }
    }
    HIP_OK(hipMemcpy(e->f, host.data(), sizeof(float) * n * F_COUNT, hipMemcpyHostToDevice));
    // Non-foot body contacts are evaluated for envs whose trunk starts a step below body_gate_height.  When the task resets every env that ends
    // a step below terminate_height >= body_gate_height (the shipped T1.yaml: 0.45 / 0.45) no env can ever start a step that low, and the whole
    // two-kernel scheme (mask bookkeeping + the second launch) is left out: the launch sequence is then exactly that of a model without spheres.
    e->body_two_kernel = model->desc.num_body_spheres > 0 && cfg->body_gate_height > cfg->terminate_height;
    if (e->body_two_kernel) {  // the ABA launch's second kernel (trunk-low envs; there also the envs whose legs can meet)
        HIP_OK(hipMalloc(&e->fd_mask, sizeof(unsigned) * ((n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK)));
        HIP_OK(hipMalloc(&e->fd_list, sizeof(int) * n));
        HIP_OK(hipMalloc(&e->fd_count, sizeof(unsigned) * 2));
        HIP_OK(hipMemset(e->fd_count, 0, sizeof(unsigned) * 2));
    }
    if (e->body_two_kernel) {
        const size_t nb = (n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
        HIP_OK(hipMalloc(&e->lowmask, sizeof(unsigned) * nb));
        HIP_OK(hipMemset(e->lowmask, 0, sizeof(unsigned) * nb));
    }
    if (cfg->state_fp16) {  // fp16 slab of the dynamic state: zeros, identity orientation
        HIP_OK(hipMalloc(&e->h, sizeof(bg_half_bits) * n * FP16_SLAB_FIELDS));
        std::vector<bg_half_bits> hh(n * FP16_SLAB_FIELDS, 0);
        for (size_t k = 0; k < n; k++) hh[(size_t)(F_ROOT + 6) * n + k] = 0x3C00;  // 1.0
        HIP_OK(hipMemcpy(e->h, hh.data(), sizeof(bg_half_bits) * hh.size(), hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" int bg_env_create(const bg_env_cfg* cfg, const bg_model* model, bg_env** out) {
    if (!cfg || !model || !out) return fail(-1, "bg_env_create: null argument");
    if (cfg->num_envs <= 0) return fail(-1, "bg_env_create: num_envs must be positive");
    if (cfg->decimation <= 0 || !(cfg->sim_dt > 0.f)) return fail(-1, "bg_env_create: bad sim dt / decimation");
    if (cfg->lin_vel_levels < 0 || cfg->ang_vel_levels < 0 || cfg->lin_vel_levels > 64 || cfg->ang_vel_levels > 64)
        return fail(-1, "bg_env_create: curriculum levels out of range");
    int ndev = 0;
    hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev == 0) return fail(-3, "bg_env_create: no HIP device available (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(-1, "bg_env_create: device ordinal out of range");
    HIP_OK(hipSetDevice(cfg->device));
    bg_env* e = new bg_env;
    const int rc = env_create_fill(e, cfg, model);
    if (rc != 0) {  // the message of the failing call is already set
        bg_env_destroy(e);
        return rc;
    }
    *out = e;
    return 0;
}

extern "C" void bg_env_destroy(bg_env* e) {
    if (!e) return;
    (void)hipFree(e->sim_tau); (void)hipFree(e->sim_bforce); (void)hipFree(e->sim_btorque);
    (void)hipFree(e->f); (void)hipFree(e->h); (void)hipFree(e->lowmask); (void)hipFree(e->fd_mask); (void)hipFree(e->fd_list); (void)hipFree(e->fd_count); (void)hipFree(e->rs_counts); (void)hipFree(e->i); (void)hipFree(e->stats); (void)hipFree(e->model_dev); (void)hipFree(e->pair_dev); (void)hipFree(e->hf); (void)hipFree(e->curr); (void)hipFree(e->curr_read);
    delete e;
}

extern "C" int bg_env_set_heightfield(bg_env* e, const int16_t* hf, int32_t rows, int32_t cols, int32_t border_px, float hscale, float vscale) {
    if (!e || !hf) return fail(-1, "bg_env_set_heightfield: null argument");
    if (rows < 2 || cols < 2 || !(hscale > 0.f)) return fail(-1, "bg_env_set_heightfield: bad dimensions");
    if (e->hf) { HIP_OK(hipFree(e->hf)); e->hf = nullptr; }
    HIP_OK(hipMalloc(&e->hf, sizeof(int16_t) * (size_t)rows * cols));
    HIP_OK(hipMemcpy(e->hf, hf, sizeof(int16_t) * (size_t)rows * cols, hipMemcpyHostToDevice));
    e->terrain.type = 1; e->terrain.rows = rows; e->terrain.cols = cols; e->terrain.border_px = border_px;
    e->terrain.inv_hscale = 1.0f / hscale; e->terrain.vscale = vscale; e->terrain.hf = e->hf;
    return 0;
}

static int upload_field(bg_env* e, int field, int comps, const float* src_env_major) {
    if (!src_env_major) return 0;
    const size_t n = (size_t)e->n;
    std::vector<float> t(n * comps);
    for (size_t k = 0; k < n; k++) for (int c = 0; c < comps; c++) t[(size_t)c * n + k] = src_env_major[k * comps + c];
    hipError_t r = hipMemcpy(e->f + (size_t)field * n, t.data(), sizeof(float) * n * comps, hipMemcpyHostToDevice);
    if (r != hipSuccess) return fail(-2, std::string("upload_field: ") + hipGetErrorString(r));
    return 0;
}

extern "C" int bg_env_set_params(bg_env* e, const float* kp, const float* kd, const float* friction, const float* mass_scale,
                                 const float* com_offset, const float* foot_material, const float* base_mass_scaled, const float* env_origins) {
    if (!e) return fail(-1, "bg_env_set_params: null env");
    int r = 0;
    if ((r = upload_field(e, F_KP, 12, kp))) return r;
    if ((r = upload_field(e, F_KD, 12, kd))) return r;
    if ((r = upload_field(e, F_FRIC, 12, friction))) return r;
    if ((r = upload_field(e, F_MASS_SCALE, 13, mass_scale))) return r;
    if ((r = upload_field(e, F_COM_OFF, 39, com_offset))) return r;
    if ((r = upload_field(e, F_FOOT_MAT, 6, foot_material))) return r;
    if ((r = upload_field(e, F_BMS, 4, base_mass_scaled))) return r;
    if ((r = upload_field(e, F_ORIGIN, 3, env_origins))) return r;
    return 0;
}

extern "C" int bg_env_bind_outputs(bg_env* e, float* obs, float* priv, float* rew, uint8_t* done, uint8_t* tout, float* terms) {
    if (!e || !obs || !priv || !rew || !done || !tout) return fail(-1, "bg_env_bind_outputs: null argument");
    e->bound.obs = obs; e->bound.priv = priv; e->bound.rew = rew; e->bound.done = done; e->bound.tout = tout; e->bound.terms = terms;
    return 0;
}

static int launch_step(bg_env* e, const float* actions, int mode, const StepOut& out, void* stream) {
    if (!out.obs || !out.priv || !out.rew || !out.done || !out.tout) return fail(-1, "bg_env_step: outputs are not bound");
    dim3 grid((e->n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK), block(64);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t cnt = (uint32_t)e->step_count;
    if (e->body_two_kernel) {
        if (e->h) hipLaunchKernelGGL((env_step_kernel<true, true>), grid, block, 0, st, env_dev(e), actions, cnt, mode, out, e->lowmask);
        else hipLaunchKernelGGL((env_step_kernel<false, true>), grid, block, 0, st, env_dev(e), actions, cnt, mode, out, e->lowmask);
    } else {
        if (e->h) hipLaunchKernelGGL((env_step_kernel<true, false>), grid, block, 0, st, env_dev(e), actions, cnt, mode, out, (unsigned*)nullptr);
        else hipLaunchKernelGGL((env_step_kernel<false, false>), grid, block, 0, st, env_dev(e), actions, cnt, mode, out, (unsigned*)nullptr);
    }
    if (e->body_two_kernel && mode == 0) {  // kernel B: the envs whose trunk was low at the start of the step (usually none)
        const int nb = (int)grid.x;
        dim3 gb(nb < BODY_GRID ? nb : BODY_GRID);
        if (e->h) hipLaunchKernelGGL(env_step_body_kernel<true>, gb, block, 0, st, env_dev(e), actions, cnt, out, (const unsigned*)e->lowmask, nb);
        else hipLaunchKernelGGL(env_step_body_kernel<false>, gb, block, 0, st, env_dev(e), actions, cnt, out, (const unsigned*)e->lowmask, nb);
    }
    if (e->rs_counts) {  // reference-exact resampling: the cross-env part of _resample_commands (see resample_apply_kernel)
        const int nb = (e->n + RS_BLOCK - 1) / RS_BLOCK;
        hipLaunchKernelGGL(resample_count_kernel, dim3(nb), dim3(RS_BLOCK), 0, st, env_dev(e), e->rs_counts);
        hipLaunchKernelGGL(resample_offsets_kernel, dim3(1), dim3(RS_BLOCK), 0, st, e->rs_counts, nb);
        hipLaunchKernelGGL(resample_apply_kernel, dim3(nb), dim3(RS_BLOCK), 0, st, env_dev(e), (const int*)e->rs_counts, nb, cnt, mode, out.obs);
    }
    HIP_OK(hipGetLastError());
    if (e->cfg.curriculum && mode == 0)  // publish this step's curriculum increments to the next step's samplers
        HIP_OK(hipMemcpyAsync(e->curr_read, e->curr, sizeof(float) * e->curr_cells, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int bg_env_reset(bg_env* e, void* stream) {
    if (!e) return fail(-1, "bg_env_reset: null env");
    return launch_step(e, nullptr, 1, e->bound, stream);
}
extern "C" int bg_env_step(bg_env* e, const float* actions, void* stream) {
    if (!e || !actions) return fail(-1, "bg_env_step: null argument");
    int r = launch_step(e, actions, 0, e->bound, stream);
    if (r == 0) e->step_count++;
    return r;
}
extern "C" int bg_env_step_to(bg_env* e, const float* actions, float* obs, float* priv, float* rew, uint8_t* done, uint8_t* tout, void* stream) {
    if (!e || !actions) return fail(-1, "bg_env_step_to: null argument");
    StepOut o = e->bound;
    o.obs = obs; o.priv = priv; o.rew = rew; o.done = done; o.tout = tout;
    int r = launch_step(e, actions, 0, o, stream);
    if (r == 0) e->step_count++;
    return r;
}

extern "C" int bg_env_get_state(bg_env* e, float* root, float* dof, float* contact, void* stream) {
    if (!e) return fail(-1, "bg_env_get_state: null env");
    hipLaunchKernelGGL(get_state_kernel, dim3((e->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_dev(e), root, dof, contact);
    HIP_OK(hipGetLastError());
    return 0;
}
extern "C" int bg_env_set_state(bg_env* e, const float* root, const float* dof, void* stream) {
    if (!e) return fail(-1, "bg_env_set_state: null env");
    hipLaunchKernelGGL(set_state_kernel, dim3((e->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_dev(e), root, dof);
    HIP_OK(hipGetLastError());
    return 0;
}

struct FieldInfo { const char* name; int off; int comps; int is_int; };
static const FieldInfo kFields[] = {
    {"root_states", F_ROOT, 13, 0}, {"dof_pos", F_Q, 12, 0}, {"dof_vel", F_QD, 12, 0}, {"last_dof_targets", F_LAST_TGT, 12, 0},
    {"actions", F_ACT, 12, 0}, {"last_actions", F_LAST_ACT, 12, 0}, {"last_dof_vel", F_LAST_QD, 12, 0}, {"last_root_vel", F_LAST_ROOTVEL, 6, 0},
    {"commands", F_CMD, 3, 0}, {"gait_frequency", F_GAIT_F, 1, 0}, {"gait_process", F_GAIT_P, 1, 0}, {"filtered_lin_vel", F_FILT_LIN, 3, 0},
    {"filtered_ang_vel", F_FILT_ANG, 3, 0}, {"last_feet_pos", F_LAST_FEET, 6, 0}, {"pushing", F_PUSH, 6, 0}, {"feet_contact_forces", F_CONTACT, 6, 0},
    {"dof_stiffness", F_KP, 12, 0}, {"dof_damping", F_KD, 12, 0}, {"dof_friction", F_FRIC, 12, 0}, {"mass_scale", F_MASS_SCALE, 13, 0},
    {"com_offset", F_COM_OFF, 39, 0}, {"foot_material", F_FOOT_MAT, 6, 0}, {"base_mass_scaled", F_BMS, 4, 0}, {"env_origins", F_ORIGIN, 3, 0},
    {"feet_pos", F_FEET_POS, 6, 0}, {"feet_roll", F_FEET_ROLL, 2, 0}, {"feet_yaw", F_FEET_YAW, 2, 0}, {"feet_contact", F_FEET_CONTACT, 2, 0},
    {"torques", F_TORQUES, 12, 0}, {"base_lin_vel", F_BASE_LIN, 3, 0}, {"base_ang_vel", F_BASE_ANG, 3, 0}, {"projected_gravity", F_PROJ_G, 3, 0},
    {"episode_sums", F_EP_SUMS, 27, 0},
    {"env_curriculum_level_lin", I_CURR_LIN, 1, 1}, {"env_curriculum_level_ang", I_CURR_ANG, 1, 1},
    {"episode_length_buf", I_EP_LEN, 1, 1}, {"cmd_resample_time", I_CMD_TIME, 1, 1}, {"delay_steps", I_DELAY, 1, 1}, {"episode_steps", I_EP_STEPS, 1, 1},
};
static const FieldInfo* find_field(const char* name) {
    for (const auto& f : kFields) if (strcmp(f.name, name) == 0) return &f;
    return nullptr;
}
extern "C" int bg_env_field_info(bg_env* e, const char* name, int32_t* comps, int32_t* is_int) {
    (void)e;
    const FieldInfo* f = name ? find_field(name) : nullptr;
    if (!f) return fail(-1, std::string("unknown field: ") + (name ? name : "(null)"));
    if (comps) *comps = f->comps;
    if (is_int) *is_int = f->is_int;
    return 0;
}
extern "C" int bg_env_get_field(bg_env* e, const char* name, void* dst, void* stream) {
    if (!e || !dst) return fail(-1, "bg_env_get_field: null argument");
    if (name && strcmp(name, "episode_stats") == 0) {  // global accumulator, STATS_COUNT floats
        HIP_OK(hipMemcpyAsync(dst, e->stats, sizeof(float) * STATS_COUNT, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return 0;
    }
    const FieldInfo* f = name ? find_field(name) : nullptr;
    if (!f) return fail(-1, std::string("unknown field: ") + (name ? name : "(null)"));
    int total = e->n * f->comps;
    hipLaunchKernelGGL(soa_to_aos_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_dev(e), f->off, f->is_int, (float*)dst, f->comps);
    HIP_OK(hipGetLastError());
    return 0;
}
extern "C" int bg_env_set_field(bg_env* e, const char* name, const void* src, void* stream) {
    if (!e || !src) return fail(-1, "bg_env_set_field: null argument");
    if (name && strcmp(name, "episode_stats") == 0) {
        HIP_OK(hipMemcpyAsync(e->stats, src, sizeof(float) * STATS_COUNT, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return 0;
    }
    const FieldInfo* f = name ? find_field(name) : nullptr;
    if (!f) return fail(-1, std::string("unknown field: ") + (name ? name : "(null)"));
    int total = e->n * f->comps;
    hipLaunchKernelGGL(aos_to_soa_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, env_dev(e), f->off, f->is_int, (const float*)src, f->comps);
    HIP_OK(hipGetLastError());
    return 0;
}
__global__ void clamp_copy_kernel(const float* src, float* dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = fminf(src[i], 1.0f);
}
extern "C" int bg_env_get_curriculum(bg_env* e, float* prob, void* stream) {
    if (!e || !prob) return fail(-1, "bg_env_get_curriculum: null argument");
    hipLaunchKernelGGL(clamp_copy_kernel, dim3((e->curr_cells + 255) / 256), dim3(256), 0, (hipStream_t)stream, e->curr, prob, e->curr_cells);
    HIP_OK(hipGetLastError());
    return 0;
}
extern "C" int bg_env_set_curriculum(bg_env* e, const float* prob, void* stream) {
    if (!e || !prob) return fail(-1, "bg_env_set_curriculum: null argument");
    HIP_OK(hipMemcpyAsync(e->curr, prob, sizeof(float) * e->curr_cells, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_OK(hipMemcpyAsync(e->curr_read, prob, sizeof(float) * e->curr_cells, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
extern "C" int64_t bg_env_step_count(const bg_env* e) { return e ? e->step_count : -1; }
extern "C" int bg_env_set_step_count(bg_env* e, int64_t c) {
    if (!e) return fail(-1, "bg_env_set_step_count: null env");
    e->step_count = c;
    return 0;
}

extern "C" int bg_env_forward_dynamics(bg_env* e, const float* root, const float* q, const float* qd, const float* tau, const float* wrench,
                                       float* qacc, void* stream) {
    if (!e || !root || !q || !qd || !tau || !qacc) return fail(-1, "bg_env_forward_dynamics: null argument");
    const int nb = (e->n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
    dim3 grid(nb), block(64);
    if (!e->body_two_kernel) hipLaunchKernelGGL(forward_dynamics_kernel<false>, grid, block, 0, (hipStream_t)stream, env_dev(e), root, q, qd, tau, wrench, qacc, (unsigned*)nullptr);
    else {
        hipLaunchKernelGGL(forward_dynamics_kernel<true>, grid, block, 0, (hipStream_t)stream, env_dev(e), root, q, qd, tau, wrench, qacc, e->fd_mask);
        // kernel B: one wave per SIMD, 1024 resident workgroups walk the list
        hipLaunchKernelGGL(aba_compact_kernel, dim3((nb + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const unsigned*)e->fd_mask, nb, e->fd_list, e->fd_count);
        hipLaunchKernelGGL(forward_dynamics_body_kernel<true>, dim3(nb < 1024 ? nb : 1024), block, 0, (hipStream_t)stream, env_dev(e), root, q, qd, tau,
                           wrench, qacc, (const int*)e->fd_list, e->fd_count);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// The same accelerations from the packed kernel (one env per lane, bg_dyn_pk.h).  43 % fewer VALU instructions per env than the lane-per-leg
// kernel and 3-5 % SLOWER at 1 M envs (215-218 against 208 us on one box): ~420 registers = one wave per SIMD, and a lone wave keeps the vector
// ALU busy 55 % of its life (profiles/r05_aba_pk_*.json, HISTORY.md).  Not available with the trunk-low gate (non-foot body contacts).
extern "C" int bg_env_forward_dynamics_packed(bg_env* e, const float* root, const float* q, const float* qd, const float* tau, const float* wrench,
                                              float* qacc, void* stream) {
    if (!e || !root || !q || !qd || !tau || !qacc) return fail(-1, "bg_env_forward_dynamics_packed: null argument");
    if (e->body_two_kernel) return fail(-4, "bg_env_forward_dynamics_packed: not available when the non-foot body contacts can occur (body_gate_height > terminate_height)");
    hipLaunchKernelGGL(forward_dynamics_pk_kernel, dim3((e->n + 63) / 64), dim3(64), 0, (hipStream_t)stream, env_dev(e), e->pair_dev, root, q, qd, tau, wrench, qacc);
    HIP_OK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ ABI: granular simulator calls (SURVEY.md 8(b) lower seam)
extern "C" int bg_sim_bind_state(bg_env* e, float* root, float* dof, float* contact, float* body) {
    if (!e || !root || !dof) return fail(-1, "bg_sim_bind_state: root and dof tensors are required");
    const size_t n = (size_t)e->n;
    if (!e->sim_tau) {
        HIP_OK(hipMalloc(&e->sim_tau, sizeof(float) * n * 12));
        HIP_OK(hipMalloc(&e->sim_bforce, sizeof(float) * n * 39));
        HIP_OK(hipMalloc(&e->sim_btorque, sizeof(float) * n * 39));
        HIP_OK(hipMemset(e->sim_tau, 0, sizeof(float) * n * 12));
        HIP_OK(hipMemset(e->sim_bforce, 0, sizeof(float) * n * 39));
        HIP_OK(hipMemset(e->sim_btorque, 0, sizeof(float) * n * 39));
    }
    e->sim_root = root; e->sim_dof = dof; e->sim_contact = contact; e->sim_body = body;
    e->sim_wrench_pending = false;
    return 0;
}
extern "C" int bg_sim_set_actuation(bg_env* e, const float* tau, void* stream) {
    if (!e || !tau) return fail(-1, "bg_sim_set_actuation: null argument");
    if (!e->sim_tau) return fail(-1, "bg_sim_set_actuation: call bg_sim_bind_state first");
    HIP_OK(hipMemcpyAsync(e->sim_tau, tau, sizeof(float) * (size_t)e->n * 12, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
extern "C" int bg_sim_apply_body_wrench_local(bg_env* e, const float* force, const float* torque, void* stream) {
    if (!e) return fail(-1, "bg_sim_apply_body_wrench_local: null env");
    if (!e->sim_tau) return fail(-1, "bg_sim_apply_body_wrench_local: call bg_sim_bind_state first");
    const size_t bytes = sizeof(float) * (size_t)e->n * 39;
    if (force) HIP_OK(hipMemcpyAsync(e->sim_bforce, force, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    else HIP_OK(hipMemsetAsync(e->sim_bforce, 0, bytes, (hipStream_t)stream));
    if (torque) HIP_OK(hipMemcpyAsync(e->sim_btorque, torque, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    else HIP_OK(hipMemsetAsync(e->sim_btorque, 0, bytes, (hipStream_t)stream));
    e->sim_wrench_pending = force || torque;
    return 0;
}
extern "C" int bg_sim_simulate(bg_env* e, void* stream) {
    if (!e) return fail(-1, "bg_sim_simulate: null env");
    if (!e->sim_root) return fail(-1, "bg_sim_simulate: call bg_sim_bind_state first");
    dim3 grid((e->n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK), block(64);
    const bool w = e->sim_wrench_pending;
    hipLaunchKernelGGL(sim_substep_kernel, grid, block, 0, (hipStream_t)stream, env_dev(e), e->sim_root, e->sim_dof, (const float*)e->sim_tau,
                       w ? (const float*)e->sim_bforce : nullptr, w ? (const float*)e->sim_btorque : nullptr, e->sim_contact, e->sim_body,
                       e->body_two_kernel ? e->lowmask : (unsigned*)nullptr);
    if (e->body_two_kernel) {
        const int nb = (int)grid.x;
        hipLaunchKernelGGL(sim_substep_body_kernel, dim3(nb < BODY_GRID ? nb : BODY_GRID), block, 0, (hipStream_t)stream, env_dev(e), e->sim_root, e->sim_dof,
                           (const float*)e->sim_tau, w ? (const float*)e->sim_bforce : nullptr, w ? (const float*)e->sim_btorque : nullptr, e->sim_contact,
                           e->sim_body, (const unsigned*)e->lowmask, nb);
    }
    HIP_OK(hipGetLastError());
    e->sim_wrench_pending = false;  // applied forces last for one simulate
    return 0;
}
extern "C" int bg_sim_refresh_body_state(bg_env* e, void* stream) {
    if (!e) return fail(-1, "bg_sim_refresh_body_state: null env");
    if (!e->sim_root) return fail(-1, "bg_sim_refresh_body_state: call bg_sim_bind_state first");
    if (!e->sim_body) return 0;
    dim3 grid((e->n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK), block(64);
    hipLaunchKernelGGL(sim_body_state_kernel, grid, block, 0, (hipStream_t)stream, env_dev(e), (const float*)e->sim_root, (const float*)e->sim_dof, e->sim_body);
    HIP_OK(hipGetLastError());
    return 0;
}
// The bound tensors ARE the simulator state, so an indexed write-back has nothing to copy; the rigid-body rows are re-derived so that
// they agree with what the caller just wrote.
extern "C" int bg_sim_write_root_state(bg_env* e, const int32_t* env_ids, int32_t count, void* stream) {
    (void)env_ids;
    if (!e) return fail(-1, "bg_sim_write_root_state: null env");
    if (count < 0 || count > e->n) return fail(-1, "bg_sim_write_root_state: count out of range");
    return bg_sim_refresh_body_state(e, stream);
}
extern "C" int bg_sim_write_dof_state(bg_env* e, const int32_t* env_ids, int32_t count, void* stream) {
    (void)env_ids;
    if (!e) return fail(-1, "bg_sim_write_dof_state: null env");
    if (count < 0 || count > e->n) return fail(-1, "bg_sim_write_dof_state: count out of range");
    return bg_sim_refresh_body_state(e, stream);
}
