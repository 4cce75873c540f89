// The packed form of the per-lane dynamics: one ENV per wavefront lane, its two legs in the halves of 64-bit register pairs (T = f2, bg_math.h).
//
// The sweeps, the sole contact and the joint limits are the SAME lane code as the fused env step's (bg_dyn.h, generic over the scalar type): with
// T = f2 their adds / multiplies / FMAs issue as v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, one instruction for both legs.  What this header
// adds is what differs when a lane owns the whole env:
//   * the nominal leg constants with the two legs side by side (PairModel: one 8-byte scalar load feeds both halves of a packed operand);
//   * the per-env link constants kept RAW (mass scale + centre-of-mass offset per link, 4 values) and cooked where the inward sweep uses
//     them, each exactly once (PkStore);
//   * the trunk as plain float code, once per env (the lane-per-leg form computes it in both lanes of an env), the two legs' contributions
//     at the trunk added as the two halves of a register pair (no lane exchange);
//   * the leg-against-leg contacts: clearance test within the lane, narrow phase item-parallel through LDS as in round 4, with lane = env
//     (64 envs per wave, up to SelfPk::PE close envs per pass).
// Replaces `gym.simulate` (reference envs/t1.py:451) for the dynamics-only launch `bg_env_forward_dynamics`; the oracle is oracle/dyn_ref.c.
#pragma once
#include "bg_dyn.h"

namespace bg {

struct PairModel {   // [.][0] = left leg (bodies 1..6), [.][1] = right leg (bodies 7..12)
    f2 pos[LEG_LINKS][3], mass[LEG_LINKS], com[LEG_LINKS][3], inertia[LEG_LINKS][6];
    f2 q_lo[LEG_LINKS], q_hi[LEG_LINKS], qd_max[LEG_LINKS];
    f2 cap_c[2][3], cap_h[2], cap_r[2];  // self-collision capsules [0 = shank, 1 = foot]: centre, half length, radius
};
inline PairModel make_pair_model(const ModelDev& m) {
    PairModel p;
    for (int i = 0; i < LEG_LINKS; i++) {
        const int l = 1 + i, r = 1 + LEG_LINKS + i;
        p.mass[i] = mk2(m.mass[l], m.mass[r]);
        for (int a = 0; a < 3; a++) { p.pos[i][a] = mk2(m.pos[l][a], m.pos[r][a]); p.com[i][a] = mk2(m.com[l][a], m.com[r][a]); }
        for (int a = 0; a < 6; a++) p.inertia[i][a] = mk2(m.inertia[l][a], m.inertia[r][a]);
        p.q_lo[i] = mk2(m.q_lo[i], m.q_lo[LEG_LINKS + i]); p.q_hi[i] = mk2(m.q_hi[i], m.q_hi[LEG_LINKS + i]);
        p.qd_max[i] = mk2(m.qd_max[i], m.qd_max[LEG_LINKS + i]);
    }
    for (int k = 0; k < 2; k++) {
        for (int a = 0; a < 3; a++) p.cap_c[k][a] = mk2(m.cap_c[0][k][a], m.cap_c[1][k][a]);
        p.cap_h[k] = mk2(m.cap_h[0][k], m.cap_h[1][k]); p.cap_r[k] = mk2(m.cap_r[0][k], m.cap_r[1][k]);
    }
    return p;
}

// The inputs of one env, one 4-byte slot each (constant slot indices: every entry is a register on the GPU; the host harness fills the same array)
struct PkSlots { enum { ROOT = 0, Q = 13, QD = 25, TAU = 37, WRENCH = 49, MS = 55, CO = 68, FM = 107, COUNT = 113 }; };
struct PkInputs {
    float v[PkSlots::COUNT];
    bool has_wrench;
    BG_HD float get(int slot) const { return v[slot]; }
    BG_HD f2 pair(int slot_l, int slot_r) const { return mk2(v[slot_l], v[slot_r]); }
    // a 12-float row as (left, right) pairs: element i of the left leg with element 6 + i of the right
    BG_HD void row(int slot0, f2* out) const { for (int i = 0; i < LEG_LINKS; i++) out[i] = mk2(v[slot0 + i], v[slot0 + LEG_LINKS + i]); }
    BG_HD f2 mass_scale(int i) const { return pair(PkSlots::MS + 1 + i, PkSlots::MS + 1 + LEG_LINKS + i); }
    BG_HD V3T<f2> com_off(int i) const {
        const int l = PkSlots::CO + 3 * (1 + i), r = PkSlots::CO + 3 * (1 + LEG_LINKS + i);
        return v3t<f2>(pair(l, r), pair(l + 1, r + 1), pair(l + 2, r + 2));
    }
};

// Sweep work space of the packed kernel: v / cb / U in registers (two per value); link constants cooked from the RAW per-env parameters (mass
// scale + centre-of-mass offset per link, 4 values) where the inward sweep uses them, each exactly once; joint ranges and link origins straight
// from the PairModel (scalar loads).  (Round 5 also measured U and cb handed over through LDS as 64-bit lane-contiguous values -- 115 fewer VALU
// instructions per wave, 1 % slower -- and the inputs copied into LDS a block ahead by a persistent grid -- 20 % slower: HISTORY.md.)
struct PkStore : RegStoreT<f2> {
    static constexpr bool ZSPEC = true, PLANE_SPEC = true;  // throughput-bound like round 4's ABA kernel: fewer issued instructions pay directly
    const PairModel* pm;
    PkInputs in;
    template <int I, class LP> BG_HD V3T<f2> link_pos(const LP&) const { return v3t<f2>(pm->pos[I][0], pm->pos[I][1], pm->pos[I][2]); }
    template <int I, class LP> BG_HD LinkConstT<f2> link(const LP& lp) const {
        return make_link_t<f2>(link_pos<I>(lp), pm->mass[I], v3t<f2>(pm->com[I][0], pm->com[I][1], pm->com[I][2]), pm->inertia[I], in.mass_scale(I), in.com_off(I));
    }
    template <int I, class LP> BG_HD f2 q_lo(const LP&) const { return pm->q_lo[I]; }
    template <int I, class LP> BG_HD f2 q_hi(const LP&) const { return pm->q_hi[I]; }
};
struct PkCtx { M3 R0; SV v0; LegWorkT<PkStore> w; };

// ---------------------------------------------------------------- leg against leg, one env per lane
constexpr int SELF_PK = 3;  // (continues the SELF_* modes of bg_dyn.h)
struct NoSwap {};           // both legs live in one lane: nothing to exchange

// One capsule segment in the world frame (relative to the trunk origin) from its link's pose and velocity.  ax = 2: along the link's z axis
// (shank), 0: along x (foot).
BG_HD SelfSeg self_item_segment(const M3& R, V3 p, SV v, V3 c, float h, float rad, int ax) {
    const V3 axis = ax == 0 ? v3(R.e[0][0], R.e[1][0], R.e[2][0]) : v3(R.e[0][2], R.e[1][2], R.e[2][2]);
    const V3 mid = p + mul(R, c);
    SelfSeg s;
    s.A = mid - h * axis; s.B = mid + h * axis;
    V3 a_loc = c;
    if (ax == 0) a_loc.e[0] -= h; else a_loc.e[2] -= h;
    s.w = mul(R, v.a);
    s.vA = mul(R, v.l + cross(v.a, a_loc));
    s.r = rad;
    return s;
}
// One pair: the left leg's segment o against the right leg's segment q.  F = world-frame force on the LEFT link (the right one takes -F at
// the same point: momentum is conserved to the bit), To / Tq = its torque about the left / right link's origin.
BG_HD void self_item_pair(const Phys& ph, const SelfSeg& o, const SelfSeg& q, V3 org_o, V3 org_q, V3* F, V3* To, V3* Tq) {
    V3 f = v3(0.f, 0.f, 0.f), xc = org_o;
    if (!self_pair(ph, o, q, &f, &xc)) { f = v3(0.f, 0.f, 0.f); xc = org_o; }
    *F = f;
    *To = cross(xc - org_o, f);
    *Tq = cross(org_q - xc, f);
}

// lateral clearance between the capsules of the two legs along the trunk's y axis (bg_dyn.h:self_clearance with both legs in the lane): the
// left leg's lowest y minus the right leg's highest, i.e. the sum of the two halves of sgn ym - yh - r with sgn = (+1, -1)
template <int AX>
BG_HD f2 self_inner_extent_pk(const M3T<f2>& R, V3T<f2> p, const f2* c, f2 h, f2 r, V3 ey) {
    const V3T<f2> e2 = splat_v3<f2>(ey);
    const f2 ym = dot(e2, p + mul(R, v3t<f2>(c[0], c[1], c[2])));
    const f2 yh = bg_abs(h * (e2.e[0] * R.e[0][AX] + e2.e[1] * R.e[1][AX] + e2.e[2] * R.e[2][AX]));
    return mk2(1.0f, -1.0f) * ym - yh - r;
}

// LDS scratch of the item-parallel narrow phase (below): poses, segments and pair results of up to PE envs per pass
struct SelfPk {
    static constexpr int PE = 12, NL = 2 * PE, NS = 4 * PE;
    static constexpr int RAW = 0, SEG = 36 * NL, RES = SEG + 13 * NS, END = RES + 9 * NS;  // floats; the P3 results reuse the segments' space
    static_assert(NS <= 64, "one worker lane per segment / pair");
};
#if defined(__HIP_DEVICE_COMPILE__)
// The narrow phase of a whole wave (64 envs), item-parallel through LDS: the close envs are ranked from the vote mask and, up to PE per pass,
//   P0  their lanes put shank and foot poses / velocities of both legs into LDS (64-bit stores: the two legs of a value are neighbours);
//   P1  lane 4 s + 2 leg + k builds ONE capsule segment (env slot s, link k of that leg);
//   P2  lane 4 s + 2 k0 + k1 evaluates ONE pair (link k0 of the left leg against link k1 of the right leg) and its torques about both origins;
//   P3  lane 4 s + 2 leg + k adds its link's two pairs and rotates force and torque into link coordinates;
//   P4  the owning lanes read their four wrenches back.
// One wave per workgroup: the LDS operations of a wave execute in order, the hand-overs need compiler fences only.
template <class W>
__device__ __forceinline__ void self_narrow_phase_pk(const Phys& ph, const PairModel& P, bool need, W& w, V3T<f2> pfoot_rel, SVT<f2> vfoot) {
    constexpr int PE = SelfPk::PE, NL = SelfPk::NL, NS = SelfPk::NS;
    const unsigned long long vote = __ballot(need);
    if (vote == 0ull) return;  // wave-uniform
    const int cnt = __popcll(vote);
    const int lane = w.self_lane;
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(vote >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vote, 0u));  // close envs in lower lanes
    lds_f32* raw = w.self_sc + SelfPk::RAW;
    lds_f32* seg = w.self_sc + SelfPk::SEG;
    lds_f32* res = w.self_sc + SelfPk::RES;
    const int gs = lane >> 2, gl = (lane >> 1) & 1, gk = lane & 1;  // this lane as worker of P1 / P3 (slot, leg, link) and of P2 (slot, k0, k1)
    const float* capf = (const float*)P.cap_c;  // [link][axis][leg]
    for (int base = 0; base < cnt; base += PE) {  // wave-uniform trip count (one pass unless more than PE envs of the wave are close)
        const int slot = rank - base;
        const bool mine = need && slot >= 0 && slot < PE;
        const bool worker = lane < NS && base + gs < cnt;
        BG_PHASE("self_p0_poses");
        if (mine) {
            typedef __attribute__((address_space(3))) f2 lds_f2;
            lds_f2* r2 = (lds_f2*)(raw + 2 * slot);  // entry idx of legs (0, 1) of this env: floats raw[idx * NL + 2 slot + leg]
            const SVT<f2> vsh = w.st.template get_v<SELF_SHANK>();
            for (int r = 0; r < 3; r++) for (int cidx = 0; cidx < 3; cidx++) { r2[(3 * r + cidx) * (NL / 2)] = w.Rsh.e[r][cidx]; r2[(18 + 3 * r + cidx) * (NL / 2)] = w.Rfoot.e[r][cidx]; }
            for (int a = 0; a < 3; a++) {
                r2[(9 + a) * (NL / 2)] = w.psh.e[a]; r2[(27 + a) * (NL / 2)] = pfoot_rel.e[a];
                r2[(12 + a) * (NL / 2)] = vsh.a.e[a]; r2[(15 + a) * (NL / 2)] = vsh.l.e[a];
                r2[(30 + a) * (NL / 2)] = vfoot.a.e[a]; r2[(33 + a) * (NL / 2)] = vfoot.l.e[a];
            }
        }
        self_lds_fence();
        BG_PHASE("self_p1_segments");
        if (worker) {
            const volatile lds_f32* vr = raw + 18 * gk * NL + 2 * gs + gl;
            M3 R; V3 p; SV v;
            for (int r = 0; r < 3; r++) for (int cidx = 0; cidx < 3; cidx++) R.e[r][cidx] = vr[(3 * r + cidx) * NL];
            for (int a = 0; a < 3; a++) { p.e[a] = vr[(9 + a) * NL]; v.a.e[a] = vr[(12 + a) * NL]; v.l.e[a] = vr[(15 + a) * NL]; }
            const V3 c = v3(capf[(gk * 3 + 0) * 2 + gl], capf[(gk * 3 + 1) * 2 + gl], capf[(gk * 3 + 2) * 2 + gl]);
            const float h = ((const float*)P.cap_h)[gk * 2 + gl], rad = ((const float*)P.cap_r)[gk * 2 + gl];
            const SelfSeg s = self_item_segment(R, p, v, c, h, rad, gk ? 0 : 2);
            for (int a = 0; a < 3; a++) { seg[a * NS + lane] = s.A.e[a]; seg[(3 + a) * NS + lane] = s.B.e[a]; seg[(6 + a) * NS + lane] = s.w.e[a]; seg[(9 + a) * NS + lane] = s.vA.e[a]; }
            seg[12 * NS + lane] = s.r;
        }
        self_lds_fence();
        BG_PHASE("self_p2_pairs");
        if (worker) {
            const int k0 = gl, k1 = gk;  // (the same bit positions, read as left-leg link / right-leg link)
            SelfSeg o, q;
            const volatile lds_f32* so = seg + 4 * gs + k0;
            const volatile lds_f32* sq = seg + 4 * gs + 2 + k1;
            for (int a = 0; a < 3; a++) {
                o.A.e[a] = so[a * NS]; o.B.e[a] = so[(3 + a) * NS]; o.w.e[a] = so[(6 + a) * NS]; o.vA.e[a] = so[(9 + a) * NS];
                q.A.e[a] = sq[a * NS]; q.B.e[a] = sq[(3 + a) * NS]; q.w.e[a] = sq[(6 + a) * NS]; q.vA.e[a] = sq[(9 + a) * NS];
            }
            o.r = so[12 * NS]; q.r = sq[12 * NS];
            const volatile lds_f32* po = raw + (18 * k0 + 9) * NL + 2 * gs;
            const volatile lds_f32* pq = raw + (18 * k1 + 9) * NL + 2 * gs + 1;
            V3 F, To, Tq;
            self_item_pair(ph, o, q, v3(po[0], po[NL], po[2 * NL]), v3(pq[0], pq[NL], pq[2 * NL]), &F, &To, &Tq);
            for (int a = 0; a < 3; a++) { res[a * NS + lane] = F.e[a]; res[(3 + a) * NS + lane] = To.e[a]; res[(6 + a) * NS + lane] = Tq.e[a]; }
        }
        self_lds_fence();
        BG_PHASE("self_p3_wrenches");
        if (worker) {
            // link gk of leg gl: the left leg's link k takes part in pairs (k, 0) and (k, 1), the right leg's in (0, k) and (1, k)
            const int i0 = 4 * gs + (gl == 0 ? 2 * gk : gk), i1 = i0 + (gl == 0 ? 1 : 2);
            const volatile lds_f32* r0 = res + i0;
            const volatile lds_f32* r1 = res + i1;
            const int to = gl == 0 ? 3 : 6;
            const float sgn = gl == 0 ? 1.0f : -1.0f;
            const V3 Fs = sgn * (v3(r0[0], r0[NS], r0[2 * NS]) + v3(r1[0], r1[NS], r1[2 * NS]));
            const V3 Ts = v3(r0[to * NS], r0[(to + 1) * NS], r0[(to + 2) * NS]) + v3(r1[to * NS], r1[(to + 1) * NS], r1[(to + 2) * NS]);
            const volatile lds_f32* vr = raw + 18 * gk * NL + 2 * gs + gl;
            M3 R;
            for (int r = 0; r < 3; r++) for (int cidx = 0; cidx < 3; cidx++) R.e[r][cidx] = vr[(3 * r + cidx) * NL];
            const V3 fl = mulT(R, Fs), fa = mulT(R, Ts);
            for (int a = 0; a < 3; a++) { seg[a * NS + lane] = fl.e[a]; seg[(3 + a) * NS + lane] = fa.e[a]; seg[(6 + a) * NS + lane] = Fs.e[a]; }
        }
        self_lds_fence();
        BG_PHASE("self_p4_read_back");
        if (mine) {
            const volatile lds_f32* f0 = seg + 4 * slot;  // + 2 leg + k
            for (int k = 0; k < 2; k++) {
                for (int a = 0; a < 3; a++) {
                    w.self_fx[k].l.e[a] = mk2(f0[a * NS + k], f0[a * NS + 2 + k]);
                    w.self_fx[k].a.e[a] = mk2(f0[(3 + a) * NS + k], f0[(3 + a) * NS + 2 + k]);
                }
                const V3T<f2> Fs = v3t<f2>(mk2(f0[6 * NS + k], f0[6 * NS + 2 + k]), mk2(f0[7 * NS + k], f0[7 * NS + 2 + k]), mk2(f0[8 * NS + k], f0[8 * NS + 2 + k]));
                if (k == 0) w.self_shank = Fs; else w.self_foot = Fs;
            }
        }
        self_lds_fence();  // the next pass overwrites the scratch
    }
    BG_PHASE("self_done");
}
#else
// host (tests/host_harness): the same items, one env, no LDS
template <class W>
inline void self_narrow_phase_pk(const Phys& ph, const PairModel& P, bool need, W& w, V3T<f2> pfoot_rel, SVT<f2> vfoot) {
    if (!need) return;
    const SVT<f2> vsh = w.st.template get_v<SELF_SHANK>();
    SelfSeg seg[2][2];  // [leg][link]
    V3 org[2][2];
    M3 R[2][2];
    for (int l = 0; l < 2; l++)
        for (int k = 0; k < 2; k++) {
            R[l][k] = k ? half(w.Rfoot, l) : half(w.Rsh, l);
            org[l][k] = k ? half(pfoot_rel, l) : half(w.psh, l);
            const V3 c = v3(P.cap_c[k][0][l], P.cap_c[k][1][l], P.cap_c[k][2][l]);
            seg[l][k] = self_item_segment(R[l][k], org[l][k], k ? half(vfoot, l) : half(vsh, l), c, P.cap_h[k][l], P.cap_r[k][l], k ? 0 : 2);
        }
    V3 Fs[2][2], Ts[2][2];
    for (int l = 0; l < 2; l++) for (int k = 0; k < 2; k++) { Fs[l][k] = v3(0.f, 0.f, 0.f); Ts[l][k] = v3(0.f, 0.f, 0.f); }
    for (int k0 = 0; k0 < 2; k0++)
        for (int k1 = 0; k1 < 2; k1++) {
            V3 F, To, Tq;
            self_item_pair(ph, seg[0][k0], seg[1][k1], org[0][k0], org[1][k1], &F, &To, &Tq);
            Fs[0][k0] = Fs[0][k0] + F; Ts[0][k0] = Ts[0][k0] + To;
            Fs[1][k1] = Fs[1][k1] - F; Ts[1][k1] = Ts[1][k1] + Tq;
        }
    for (int k = 0; k < 2; k++) {
        const V3 fl0 = mulT(R[0][k], Fs[0][k]), fl1 = mulT(R[1][k], Fs[1][k]), fa0 = mulT(R[0][k], Ts[0][k]), fa1 = mulT(R[1][k], Ts[1][k]);
        for (int a = 0; a < 3; a++) { w.self_fx[k].l.e[a] = mk2(fl0.e[a], fl1.e[a]); w.self_fx[k].a.e[a] = mk2(fa0.e[a], fa1.e[a]); }
        const V3T<f2> F2 = v3t<f2>(mk2(Fs[0][k].e[0], Fs[1][k].e[0]), mk2(Fs[0][k].e[1], Fs[1][k].e[1]), mk2(Fs[0][k].e[2], Fs[1][k].e[2]));
        if (k == 0) w.self_shank = F2; else w.self_foot = F2;
    }
}
#endif

// the SELF_PK overload of bg_dyn.h:self_contacts (found by leg_phase1 through the f2 argument types)
template <int MODE, class W>
BG_HD void self_contacts(const Phys& ph, const ModelDev&, int, W& w, const M3& R0, V3T<f2> pfoot_rel, SVT<f2> vfoot, NoSwap&) {
    static_assert(MODE == SELF_PK, "two legs per lane: SELF_PK");
    w.self_fx[0] = svt_zero<f2>(); w.self_fx[1] = svt_zero<f2>();
    w.self_shank = v3_zero<f2>(); w.self_foot = v3_zero<f2>();
    w.self_deferred = false;
    w.self_gap = 1.0f;
    if (!ph.self_on) return;
    const PairModel& P = *w.st.pm;
    const V3 ey = v3(R0.e[0][1], R0.e[1][1], R0.e[2][1]);
    const f2 inner = bg_min(self_inner_extent_pk<2>(w.Rsh, w.psh, P.cap_c[0], P.cap_h[0], P.cap_r[0], ey),
                            self_inner_extent_pk<0>(w.Rfoot, pfoot_rel, P.cap_c[1], P.cap_h[1], P.cap_r[1], ey));
    w.self_gap = inner[0] + inner[1];
    self_narrow_phase_pk(ph, P, w.self_gap < 0.f, w, pfoot_rel, vfoot);
}

// ---------------------------------------------------------------- one env: accelerations of the trunk and of both legs
// per-env contact parameters of both feet from the raw foot materials (bg_dyn.h:load_leg_params)
BG_HD void pk_foot_params(const ContactCfg& cc, f2 mu_f, f2 compl_f, f2 rest_f, LegParamsT<f2>& lp) {
    lp.mu = 0.5f * (mu_f + cc.terrain_mu);  // PhysX default material combine: average
    lp.kn = cc.k * bg_rcp(compl_f);
    lp.dn = cc.d * (1.0f - 0.5f * (rest_f + cc.terrain_restitution));
}
// cx.w.st.pm / .in and (GPU) cx.w.self_sc / .self_lane are set by the caller.  qdd[i] = (left, right) joint i; foot_force = world frame, per foot.
BG_HD void pk_forward_env(const Phys& ph, const ContactCfg& cc, const TerrainDev& tr, const ModelDev& M, PkCtx& cx, f2* qdd, V3* lin_w, V3* ang_w, V3T<f2>* foot_force) {
    const PkInputs& in = cx.w.st.in;
    BG_PHASE("load_state");
    BaseState bs;
    bs.pos = v3(in.get(PkSlots::ROOT), in.get(PkSlots::ROOT + 1), in.get(PkSlots::ROOT + 2));
    for (int a = 0; a < 4; a++) bs.quat[a] = in.get(PkSlots::ROOT + 3 + a);
    bs.vlin = v3(in.get(PkSlots::ROOT + 7), in.get(PkSlots::ROOT + 8), in.get(PkSlots::ROOT + 9));
    bs.vang = v3(in.get(PkSlots::ROOT + 10), in.get(PkSlots::ROOT + 11), in.get(PkSlots::ROOT + 12));
    LegStateT<f2> ls;
    in.row(PkSlots::Q, ls.q); in.row(PkSlots::QD, ls.qd);
    BG_PHASE("base_kinematics");
    cx.R0 = quat_to_mat(bs.quat);
    cx.v0 = base_body_velocity(cx.R0, bs);
    LegWorkT<PkStore>& w = cx.w;
    w.zmask = ph.zmask;
#ifdef BG_CENSUS_ZMASK
    w.zmask = BG_CENSUS_ZMASK;
#endif
    LegParamsT<f2> lp;   // sole corners and foot materials (the store serves link constants and joint ranges)
    for (int k = 0; k < 4; k++) lp.corner[k] = v3(M.corner[k][0], M.corner[k][1], M.corner[k][2]);
    SVT<f2> vfoot;
    V3T<f2> pfoot_rel;
    leg_outward<0>(lp, ls, w, splat_sv<f2>(cx.v0), splat_m3<f2>(cx.R0), v3_zero<f2>(), &vfoot, &pfoot_rel);
    BG_PHASE("self_clearance");
    NoSwap x;
    self_contacts<SELF_PK>(ph, M, 0, w, cx.R0, pfoot_rel, vfoot, x);
    pk_foot_params(cc, in.pair(PkSlots::FM, PkSlots::FM + 3), in.pair(PkSlots::FM + 1, PkSlots::FM + 4), in.pair(PkSlots::FM + 2, PkSlots::FM + 5), lp);
    f2 tau[LEG_LINKS];
    in.row(PkSlots::TAU, tau);
    BaseContributionT<f2> legs = leg_phase1_inward(ph, tr, lp, ls, tau, bs, w, vfoot, pfoot_rel);
    bg_pin(legs.I); bg_pin(legs.p);
    BG_PHASE("pair_sum");
    const LinkConst bk = make_link(M, 0, in.get(PkSlots::MS), v3(in.get(PkSlots::CO), in.get(PkSlots::CO + 1), in.get(PkSlots::CO + 2)));
    SV wrench = sv_zero();
    if (in.has_wrench) {
        wrench.l = v3(in.get(PkSlots::WRENCH), in.get(PkSlots::WRENCH + 1), in.get(PkSlots::WRENCH + 2));
        wrench.a = v3(in.get(PkSlots::WRENCH + 3), in.get(PkSlots::WRENCH + 4), in.get(PkSlots::WRENCH + 5));
    }
    BaseContribution own = base_own(bk, cx.v0, wrench);
    SI I;
    for (int k = 0; k < 6; k++) { I.A.e[k] = own.I.A.e[k] + (legs.I.A.e[k][0] + legs.I.A.e[k][1]); I.M.e[k] = own.I.M.e[k] + (legs.I.M.e[k][0] + legs.I.M.e[k][1]); }
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) I.H.e[a][b] = own.I.H.e[a][b] + (legs.I.H.e[a][b][0] + legs.I.H.e[a][b][1]);
    SV p;
    p.a = own.p.a + sum_halves(legs.p.a); p.l = own.p.l + sum_halves(legs.p.l);
    BG_PHASE("base_solve");
    const SV a0p = base_solve(I, p);
    SVT<f2> afoot;
    leg_accel<0>(ph, lp, ls, w, splat_sv<f2>(a0p), qdd, &afoot);
    bg_pin(afoot);
    BG_PHASE("rates_and_foot_force");
    base_world_rates(cx.R0, cx.v0, a0p, ph.g, lin_w, ang_w);
    *foot_force = foot_force_over_step(ph, w, afoot);
}

}  // namespace bg
