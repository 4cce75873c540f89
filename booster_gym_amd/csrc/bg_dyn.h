// Per-lane articulated-body dynamics of the T1 legs (one LEG per wavefront lane,
// two neighbouring lanes = one environment).
//
// Replaces `gym.simulate` (reference envs/t1.py:451, PhysX inside Isaac Gym) for the
// 13-body / 12-DoF collapsed T1 (resources/T1/T1_locomotion.xml:36-139).  Algorithm:
// floating-base articulated-body algorithm (Featherstone, RBDA Table 9.4) in body
// coordinates, specialised to this robot's structure:
//   * two identical 6-link serial chains hanging off the trunk -> each lane owns one
//     chain; the only exchange is the chain's articulated inertia / bias force at the
//     trunk (27 floats, one DPP lane swap per value)
//   * every joint is revolute about a coordinate axis of its own frame and no body
//     frame is rotated against its parent at q=0 -> joint transforms are a 2-D rotation
//     (compile-time axis) plus a translation
//   * spatial inertias are kept as 3x3 blocks [A H; H^T M] with A, M symmetric
// Contact and joint limits are linearly-implicit penalty forces folded into the
// articulated inertia; the oracle in oracle/dyn_ref.c states the
// same model with dense 6x6 algebra in double precision.
#pragma once
#include "bg_math.h"

// Instruction census (tools/isa_census.py builds with -DBG_ISA_PHASES): named markers that the scheduler may not move code across, so that
// the VALU instructions of the emitted ISA can be attributed to the phases of the algorithm.  No effect on the product build.
#if defined(BG_ISA_PHASES) && defined(__HIP_DEVICE_COMPILE__)
#define BG_PHASE(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; BG_PHASE " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#define BG_PIN1(x) asm volatile("" : "+v"(x))   // the value exists HERE: what produces it cannot sink below, what uses it cannot rise above
#else
#define BG_PHASE(name) do { } while (0)
#define BG_PIN1(x) do { } while (0)
#endif

namespace bg {

constexpr int LEG_LINKS = 6;
// Hip_Pitch y, Hip_Roll x, Hip_Yaw z, Knee_Pitch y, Ankle_Pitch y, Ankle_Roll x  (T1_locomotion.xml:56-79)
constexpr int LEG_AXIS[LEG_LINKS] = {2, 1, 3, 2, 2, 1};
constexpr int SELF_SHANK = 3, SELF_FOOT = 5;  // the leg links that carry a self-collision capsule

// The lane code below is generic over the scalar type T (bg_math.h): T = float, one LEG per lane (LinkConst, LegParams, ... are the float
// instances: the fused env step, the granular simulator calls); T = f2, both legs of an env in the halves of 64-bit register pairs, one ENV
// per lane (bg_dyn_pk.h: the packed ABA kernel).  What cannot be a branch per half is written as a select (sel / both / any_of).
template <class T> struct LinkConstT {   // one rigid body, per-env randomisation already applied
    V3T<T> pos;      // origin in the parent frame
    T m;             // mass
    V3T<T> mc;       // mass * centre of mass
    S3T<T> Io;       // rotational inertia about the body origin
};
using LinkConst = LinkConstT<float>;

struct Phys {
    float dt;
    V3 g;
    float contact_ramp, friction_visc, limit_k, limit_d;
    int clamp_qd;
    // non-foot body contacts (explicit penalty on the contact spheres of the trunk box / hip-yaw and shank cylinders)
    float body_gate, body_kn, body_dn, body_mu;
    // bit i: the origin of leg link i lies on the z axis of its parent's frame (pos = (0, 0, z)) in BOTH legs -- true for the T1's hip-roll,
    // hip-yaw and ankle-pitch links.  Wave-uniform, so the sweeps branch on it at no cost and run their r-dependent parts (cross products with
    // the link position, the shift of the articulated inertia) with the two zero components folded away: 46 VALU instructions per such link.
    // 0 = the general code for every link.
    int zmask;
    // leg against leg (explicit penalty between the shank / foot capsules of the two legs)
    int self_on;
    float self_k, self_d, self_mu, self_visc;
};

struct TerrainDev {
    int type;  // 0 plane, 1 heightfield
    int rows, cols, border_px;
    float inv_hscale, vscale;
    const int16_t* hf;
};

struct ModelDev {  // nominal (un-randomised) model, shared by all envs; filled by bg_model_create
    float pos[13][3];
    float mass[13];
    float com[13][3];
    float inertia[13][6];  // about the centre of mass: xx yy zz xy xz yz
    float q_lo[12], q_hi[12], qd_max[12], tau_lim[12];
    float corner[4][3];
    // contact spheres of the non-foot collision shapes, sorted by body: body b owns sph_first[b] .. + sph_cnt[b]
    int sph_n;
    int sph_first[13], sph_cnt[13];
    float sph_pos[16][3], sph_r[16];
    // self-collision capsules [leg][0 = shank (leg link 3, axis z), 1 = foot (leg link 5, axis x)]: centre, half length along the axis, radius
    float cap_c[2][2][3], cap_h[2][2], cap_r[2][2];
};

template <class T> struct LegParamsT {
    LinkConstT<T> lk[LEG_LINKS];
    T q_lo[LEG_LINKS], q_hi[LEG_LINKS], qd_max[LEG_LINKS];
    V3 corner[4];  // (the same four sole corners under both feet)
    T mu, kn, dn;  // combined friction, normal stiffness, normal damping of this foot
};
using LegParams = LegParamsT<float>;

template <class T> struct LegStateT { T q[LEG_LINKS], qd[LEG_LINKS]; };
using LegState = LegStateT<float>;

struct BaseState {
    V3 pos;
    float quat[4];  // xyzw
    V3 vlin, vang;  // world frame (Isaac Gym root-state layout, t1.py:221-222)
};

// Where the sweeps keep what they hand to each other: the per-link vectors v (outward -> inward), cb (outward -> inward, accel) and
// U = IA S (inward -> accel), and the per-env link constants.  RegStore: everything in registers (the fused env step: one wave per CU at
// 4096 envs, latency is all that matters).  LdsLinkStore: link constants in LDS, lane-contiguous (slot k of lane l at p[64 k],
// conflict-free), which is what lets two waves share a SIMD in forward_dynamics_kernel once the chip is full.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) float lds_f32;  // explicit LDS pointers: volatile accesses are not address-space-inferred
#else
typedef float lds_f32;
#endif
// LDS scratch of the item-parallel leg-against-leg narrow phase (SELF_LDS, self_narrow_phase_lds): poses, segments and pair results of up to 7 envs per pass
constexpr int SELF_PASS_ENVS = 7, SELF_LDS_FLOATS = (36 * 2 + 13 * 4 + 9 * 4) * SELF_PASS_ENVS;
#ifndef BG_ENV_PLANE_SPEC
#define BG_ENV_PLANE_SPEC 1
#endif
template <class T> struct RegStoreT {
    using Scalar = T;
    // the plane-specialised sole contact (foot_contact) also in the fused env step: 104.95 -> 103.1 us per env step on flat ground, no change on the
    // height field (tools/ab_sim.py, three alternating pairs on one box) -- unlike the z-axis specialisation below, this branch replaces a loop body
    static constexpr bool PLANE_SPEC = BG_ENV_PLANE_SPEC;
    // the z-axis specialisation of the sweeps (Phys::zmask) is off here: this store serves the fused env step, one wave per SIMD and latency-bound,
    // where the extra (wave-uniform) branches split the blocks the scheduler overlaps work in: 107.7 -> 109.3 us per env step with it on (measured)
    static constexpr bool ZSPEC = false;
    SVT<T> v[LEG_LINKS], cb[LEG_LINKS], U[LEG_LINKS];
    template <int I> BG_HD void put_v(SVT<T> x) { v[I] = x; }
    template <int I> BG_HD SVT<T> get_v() const { return v[I]; }
    template <int I> BG_HD void put_cb(SVT<T> x) { cb[I] = x; }
    template <int I> BG_HD SVT<T> get_cb() const { return cb[I]; }
    template <int I> BG_HD void put_U(SVT<T> x) { U[I] = x; }
    template <int I> BG_HD SVT<T> get_U() const { return U[I]; }
    template <int I, class LP> BG_HD V3T<T> link_pos(const LP& lp) const { return lp.lk[I].pos; }
    template <int I, class LP> BG_HD LinkConstT<T> link(const LP& lp) const { return lp.lk[I]; }
    // joint range and applied torque of link I: plain members here; the packed kernel's store fetches them where the inward sweep asks
    template <int I, class LP> BG_HD T q_lo(const LP& lp) const { return lp.q_lo[I]; }
    template <int I, class LP> BG_HD T q_hi(const LP& lp) const { return lp.q_hi[I]; }
    template <int I> BG_HD T tau_at(const T* tau) const { return tau[I]; }
};
using RegStore = RegStoreT<float>;
// v / cb / U in registers, the per-env link constants (10 floats per link: mass, m c, inertia about the origin; computed once per launch) in LDS.
// The link ORIGINS are not per-env (nominal model constants, one set per leg): they sit in a 2 x 6 x 4-float table of the workgroup that every
// lane of a leg reads at the same address (a broadcast read) -- 18 fewer lane-wide slots, which is what pays for the leg-against-leg scratch
// below within the 20 KB a workgroup may use when eight of them share a CU.
struct LdsLinkStore : RegStore {
    static constexpr bool ZSPEC = true;  // the ABA kernel: throughput-bound (two waves per SIMD), fewer issued instructions pay directly
    static constexpr bool PLANE_SPEC = true;
    static constexpr int PER_LINK = 10, SLOTS = PER_LINK * LEG_LINKS, STRIDE = 64;
    static constexpr int POS_FLOATS = 2 * LEG_LINKS * 4;  // [leg][link][x y z pad]
    static constexpr int FLOATS = SLOTS * STRIDE + POS_FLOATS + SELF_LDS_FLOATS;
    static_assert(FLOATS * 4 <= 20480, "eight workgroups of the ABA kernel share a CU's 160 KB of LDS");
    lds_f32* p;     // + lane: this lane's slots, STRIDE apart
    lds_f32* ppos;  // this leg's row of the origin table
    BG_HD void bind(lds_f32* base, int lane) { p = base + lane; ppos = base + SLOTS * STRIDE + (lane & 1) * (LEG_LINKS * 4); }
    static BG_HD lds_f32* self_scratch(lds_f32* base) { return base + SLOTS * STRIDE + POS_FLOATS; }
    // the origin table: written once per workgroup, by ALL its lanes together, before any lane code runs (the lane code may be called from
    // divergent branches, where the lanes that would write it need not be the first to read it)
    template <class MD> static BG_HD void write_origins(lds_f32* base, const MD& m, int lane) {
        if (lane < 2 * LEG_LINKS * 3) {
            const int leg = lane / (3 * LEG_LINKS), i = (lane % (3 * LEG_LINKS)) / 3, a = lane % 3;
            base[SLOTS * STRIDE + leg * (LEG_LINKS * 4) + 4 * i + a] = m.pos[1 + leg * LEG_LINKS + i][a];
        }
    }
    template <class LP> BG_HD void stash(const LP& lp) const {
        for (int i = 0; i < LEG_LINKS; i++) {
            const LinkConst& k = lp.lk[i];
            const float f[PER_LINK] = {k.m, k.mc.e[0], k.mc.e[1], k.mc.e[2], k.Io.e[0], k.Io.e[1], k.Io.e[2], k.Io.e[3], k.Io.e[4], k.Io.e[5]};
            for (int j = 0; j < PER_LINK; j++) p[(PER_LINK * i + j) * STRIDE] = f[j];
        }
    }
    // volatile reads: a plain load is forwarded from the stash stores above and the constants stay in registers after all
    template <int I, class LP> BG_HD V3 link_pos(const LP&) const {
        const volatile lds_f32* q = ppos;
        return v3(q[4 * I], q[4 * I + 1], q[4 * I + 2]);
    }
    template <int I, class LP> BG_HD LinkConst link(const LP& lp) const {
        const volatile lds_f32* q = p;
        LinkConst k;
        k.pos = link_pos<I>(lp);
        k.m = q[(PER_LINK * I) * STRIDE];
        k.mc = v3(q[(PER_LINK * I + 1) * STRIDE], q[(PER_LINK * I + 2) * STRIDE], q[(PER_LINK * I + 3) * STRIDE]);
        for (int j = 0; j < 6; j++) k.Io.e[j] = q[(PER_LINK * I + 4 + j) * STRIDE];
        return k;
    }
};

template <class Store>
struct LegWorkT {  // what the inward sweep leaves behind for the outward sweep
    using T = typename Store::Scalar;
    using Scalar = T;
    T c[LEG_LINKS], s[LEG_LINKS];
    int zmask;         // Phys::zmask, for the sweeps that do not see Phys
    Store st;
    T dinv[LEG_LINKS], u[LEG_LINKS];
    M3T<T> Rfoot;             // foot -> world
    M3T<T> Rsh; V3T<T> psh;   // shank -> world, shank origin relative to the trunk origin (for the leg-against-leg contacts; dead after them)
    // leg-against-leg contacts: wrench on this leg's shank / foot about the link origin in link coordinates, their world-frame forces,
    // the lateral clearance between the two legs (negative = capsules can meet), and "left to the second kernel" (SELF_DEFER)
    SVT<T> self_fx[2];
    V3T<T> self_shank, self_foot;
    float self_gap;
    bool self_deferred;
    lds_f32* self_sc;  // SELF_LDS / SELF_PK: the workgroup's scratch and this lane's index, set by the kernel
    int self_lane;
    SIT<T> Bc;         // contact impedance on the foot
    SVT<T> f0c;        // contact wrench at the current state (foot coords)
    bool contact;      // (T = f2: of either foot; the impedance and the wrench of a foot in the air are zero)
};
using LegWork = LegWorkT<RegStore>;

template <class T> struct BaseContributionT { SIT<T> I; SVT<T> p; };  // articulated inertia / bias force seen at the trunk
using BaseContribution = BaseContributionT<float>;
template <class T> BG_HD void bg_pin(V3T<T>& v) { for (int a = 0; a < 3; a++) BG_PIN1(v.e[a]); }
template <class T> BG_HD void bg_pin(SVT<T>& v) { bg_pin(v.a); bg_pin(v.l); }
template <class T> BG_HD void bg_pin(M3T<T>& m) { for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) BG_PIN1(m.e[a][b]); }
template <class T> BG_HD void bg_pin(S3T<T>& m) { for (int a = 0; a < 6; a++) BG_PIN1(m.e[a]); }
template <class T> BG_HD void bg_pin(SIT<T>& i) { bg_pin(i.A); bg_pin(i.H); bg_pin(i.M); }

template <class T> BG_HD SIT<T> rigid_inertia(const LinkConstT<T>& k) {
    SIT<T> I;
    I.A = k.Io;
    I.H = skew(k.mc);
    I.M = s3t_zero<T>();
    I.M.e[0] = I.M.e[1] = I.M.e[2] = k.m;
    return I;
}
template <class T> BG_HD SVT<T> mul(const SIT<T>& I, SVT<T> v) {
    SVT<T> f;
    f.a = mul(I.A, v.a) + mul(I.H, v.l);
    f.l = mulT(I.H, v.a) + mul(I.M, v.l);
    return f;
}
// rigid-body shortcut of  I v  (H = skew(mc), M = m 1)
template <class T> BG_HD SVT<T> mul_rigid(const LinkConstT<T>& k, SVT<T> v) {
    SVT<T> f;
    f.a = mul(k.Io, v.a) + cross(k.mc, v.l);
    f.l = k.m * v.l - cross(k.mc, v.a);
    return f;
}
// v x* f  (spatial force cross product, RBDA eq. 2.32)
template <class T> BG_HD SVT<T> crf(SVT<T> v, SVT<T> f) {
    SVT<T> o;
    o.a = cross(v.a, f.a) + cross(v.l, f.l);
    o.l = cross(v.a, f.l);
    return o;
}

// bilinear terrain height and surface normal (reference utils/terrain.py:101-121; indices clamped)
BG_HD void terrain_query(const TerrainDev& t, float x, float y, float* h, V3* n) {
#ifdef BG_CENSUS_PLANE   // tools/isa_census.py --plane: count the instructions a launch on flat ground executes (the height-field branch is never taken there)
    *h = 0.f; *n = v3(0.f, 0.f, 1.f); return;
#endif
    if (t.type == 0) { *h = 0.f; *n = v3(0.f, 0.f, 1.f); return; }
    float px = (float)t.border_px + x * t.inv_hscale, py = (float)t.border_px + y * t.inv_hscale;
    int x1 = (int)floorf(px), y1 = (int)floorf(py);
    x1 = x1 < 0 ? 0 : (x1 > t.rows - 2 ? t.rows - 2 : x1);
    y1 = y1 < 0 ? 0 : (y1 > t.cols - 2 ? t.cols - 2 : y1);
    float fx = px - (float)x1, fy = py - (float)y1;
    const int16_t* p = t.hf + (size_t)x1 * t.cols + y1;
    float h00 = (float)p[0], h01 = (float)p[1], h10 = (float)p[t.cols], h11 = (float)p[t.cols + 1];
    *h = ((1.f - fx) * (1.f - fy) * h00 + fx * (1.f - fy) * h10 + (1.f - fx) * fy * h01 + fx * fy * h11) * t.vscale;
    float hx = ((1.f - fy) * (h10 - h00) + fy * (h11 - h01)) * t.vscale * t.inv_hscale;
    float hy = ((1.f - fx) * (h01 - h00) + fx * (h11 - h10)) * t.vscale * t.inv_hscale;
    float inv = bg_rsqrt(hx * hx + hy * hy + 1.0f);
    *n = v3(-hx * inv, -hy * inv, inv);
}
BG_HD void terrain_query(const TerrainDev& t, f2 x, f2 y, f2* h, V3T<f2>* n) {  // both feet of a lane, one after the other
    float h0, h1; V3 n0, n1;
    terrain_query(t, x[0], y[0], &h0, &n0);
    terrain_query(t, x[1], y[1], &h1, &n1);
    *h = mk2(h0, h1);
    *n = v3t<f2>(mk2(n0.e[0], n1.e[0]), mk2(n0.e[1], n1.e[1]), mk2(n0.e[2], n1.e[2]));
}
BG_HD float terrain_height(const TerrainDev& t, float x, float y) {
    float h; V3 n;
    terrain_query(t, x, y, &h, &n);
    return h;
}

// ---------------------------------------------------------------- non-foot body contacts (trunk box, hip-yaw / shank cylinders)
// Explicit penalty contact of a body's contact spheres against the terrain: same normal / friction law as the sole corners, evaluated at
// the current state only (these links are heavy enough for an explicit force at dt = 2 ms), default material averaged with the terrain's.
// R, p = body pose in the world, v = body velocity in body coordinates.  sel < 0: all spheres of body b; sel = 0 / 1: those whose index has
// that parity (the two lanes of an env share the trunk's spheres).  Returns the wrench about the body origin in body coordinates
// (a = torque, l = force) and adds the world-frame force to *fw.
BG_HD SV body_contact_wrench(const Phys& ph, const TerrainDev& tr, const ModelDev& M, int b, int sel, const M3& R, V3 p, SV v, V3* fw) {
    SV w = sv_zero();
    const int first = M.sph_first[b], cnt = M.sph_cnt[b];
    for (int k = first; k < first + cnt; k++) {
        if (sel >= 0 && (k & 1) != sel) continue;
        const V3 c = v3(M.sph_pos[k][0], M.sph_pos[k][1], M.sph_pos[k][2]);
        const float r = M.sph_r[k];
        const V3 xw = p + mul(R, c);
        float h; V3 n;
        terrain_query(tr, xw.e[0], xw.e[1], &h, &n);
        const float pen = (h - xw.e[2]) * n.e[2] + r;
        if (!(pen > 0.f)) continue;
        const V3 nb = mulT(R, n);
        const V3 rc = c - r * nb;  // contact point = sphere centre - r n
        const V3 vw = mul(R, v.l + cross(v.a, rc));
        const float vn = dot(vw, n);
        const float ramp = pen < ph.contact_ramp ? pen * bg_rcp(ph.contact_ramp) : 1.0f;
        const float fn0 = ph.body_kn * pen - ph.body_dn * ramp * vn;
        if (!(fn0 > 0.f)) continue;
        const V3 vt = vw - vn * n;
        const float c_t = fminf(ph.friction_visc, ph.body_mu * fn0 * bg_rcp(bg_sqrt(dot(vt, vt)) + 1e-6f));
        const V3 f_w = fn0 * n - c_t * vt;
        const V3 fb = mulT(R, f_w);
        w.l = w.l + fb;
        w.a = w.a + cross(rc, fb);
        *fw = *fw + f_w;
    }
    return w;
}
// Kinematics-only walk down one leg that evaluates the contact spheres of its links: run BEFORE the sweeps, and only while the trunk is low
// (substep_pre), so that the sweeps themselves exist once and carry no contact code for these bodies.  fext[I] += contact wrench of link I
// about its origin (link coordinates), fw[I] = its world-frame force.
template <int I, class St>
BG_HD void leg_contact_prepass(const Phys& ph, const TerrainDev& tr, const ModelDev& M, int body0, const St& st, const LegParams& lp, const LegState& ls,
                               SV vpar, M3 Rpar, V3 ppar, SV* fext, V3* fw) {
    constexpr int AX = LEG_AXIS[I], A = AX - 1;
    float s, c;
    bg_sincos(ls.q[I], &s, &c);
    const V3 lpos = st.template link_pos<I>(lp);
    SV v;
    v.a = rotT<AX>(c, s, vpar.a);
    v.l = rotT<AX>(c, s, vpar.l + cross(vpar.a, lpos));
    v.a.e[A] += ls.qd[I];
    const V3 p = ppar + mul(Rpar, lpos);
    M3 R = Rpar;
    {
        constexpr int J = Plane<AX>::J, K = Plane<AX>::K;
        for (int r = 0; r < 3; r++) {
            R.e[r][J] = c * Rpar.e[r][J] + s * Rpar.e[r][K];
            R.e[r][K] = -s * Rpar.e[r][J] + c * Rpar.e[r][K];
        }
    }
    if (M.sph_cnt[body0 + I] > 0) {
        V3 f = v3(0.f, 0.f, 0.f);
        fext[I] = fext[I] + body_contact_wrench(ph, tr, M, body0 + I, -1, R, p, v, &f);
        fw[I] = f;
    }
    if constexpr (I + 1 < LEG_LINKS - 1) leg_contact_prepass<I + 1>(ph, tr, M, body0, st, lp, ls, v, R, p, fext, fw);  // the foot has its own contacts
}

// ---------------------------------------------------------------- outward sweep, link I
template <int I, class W, class T = typename W::Scalar>
BG_HD void leg_outward(const LegParamsT<T>& lp, const LegStateT<T>& ls, W& w, SVT<T> vpar, M3T<T> Rpar, V3T<T> ppar, SVT<T>* vfoot, V3T<T>* pfoot) {
    constexpr int AX = LEG_AXIS[I], A = AX - 1;
    bg_pin(vpar); bg_pin(Rpar); bg_pin(ppar);
    BG_PHASE("outward_link");
    T s, c;
    bg_sincos(ls.q[I], &s, &c);
    w.c[I] = c; w.s[I] = s;
    // v_i = X v_parent + S qd :  w' = E w ; v' = E (v + w x r)
    SVT<T> v;
    v.a = rotT<AX>(c, s, vpar.a);
    const V3T<T> lpos = w.st.template link_pos<I>(lp);
    V3T<T> vlin, p;  // linear velocity of the link origin in parent coordinates, link origin relative to the trunk origin
    if ((w.zmask >> I) & 1) {  // r = (0, 0, z): wave-uniform branch (Phys::zmask)
        const V3T<T> rz = v3t<T>(splat<T>(0.f), splat<T>(0.f), lpos.e[2]);
        vlin = vpar.l + cross(vpar.a, rz);
        p = ppar + mul(Rpar, rz);
    } else {
        vlin = vpar.l + cross(vpar.a, lpos);
        p = ppar + mul(Rpar, lpos);
    }
    v.l = rotT<AX>(c, s, vlin);
    // c_i = v_i x (S qd)   (S = unit angular axis A)
    V3T<T> sq = v3_zero<T>(); sq.e[A] = ls.qd[I];
    SVT<T> cb;
    cb.a = cross(v.a, sq);
    cb.l = cross(v.l, sq);
    w.st.template put_cb<I>(cb);
    v.a.e[A] += ls.qd[I];
    w.st.template put_v<I>(v);
    // world-aligned pose of the link (for the foot and the self-collision capsules, carried down the chain)
    M3T<T> R;  // R_world_child = R_world_parent * R(axis,q): rotate the columns J,K
    {
        constexpr int J = Plane<AX>::J, K = Plane<AX>::K;
        R = Rpar;
        for (int r = 0; r < 3; r++) {
            R.e[r][J] = c * Rpar.e[r][J] + s * Rpar.e[r][K];
            R.e[r][K] = -s * Rpar.e[r][J] + c * Rpar.e[r][K];
        }
    }
    if constexpr (I == SELF_SHANK) { w.Rsh = R; w.psh = p; }
    if constexpr (I + 1 < LEG_LINKS) {
        leg_outward<I + 1>(lp, ls, w, v, R, p, vfoot, pfoot);
    } else {
        w.Rfoot = R;
        *pfoot = p;
        *vfoot = v;
    }
}

// ---------------------------------------------------------------- leg against leg (self-collision)
// The reference runs PhysX with self-collision enabled (envs/T1.yaml:69 `self_collisions: 0`, passed to create_actor at envs/t1.py:128).
// Here: the shank cylinder and the foot box of each leg are capsules (ModelDev::cap_*), every capsule of the left leg can meet every capsule
// of the right leg, and a pair that overlaps repels with an explicit penalty force along the line between the closest points of the two
// segments, with regularised Coulomb friction, applied with opposite signs to the two links at ONE point (the middle of the overlap), so
// momentum is conserved.  oracle/dyn_ref.c:self_contacts states the same model in float64.
//
// Lane structure: a lane owns one leg.  Cheap part, every substep: the lateral extent (trunk y axis) of the own two capsules, one value
// swapped with the partner lane; the legs are on opposite sides of the trunk, so capsules can only meet when the left leg's lowest y is below
// the right leg's highest y.  Only then (both lanes of the env take the branch together: the test is symmetric) the lanes exchange their
// segments (26 values) and each evaluates the four pairs for the forces on ITS links.
constexpr float SELF_REG = 1e-2f;   // Tikhonov term of the closest-point problem: unique, continuous answer for parallel segments
struct SelfSeg { V3 A, B, w, vA; float r; };  // world frame: end points, angular velocity, velocity of the point A, radius

template <int AX>
BG_HD SelfSeg self_segment(const M3& R, V3 p, SV v, V3 c, float h, float r) {
    const V3 ax = v3(R.e[0][AX], R.e[1][AX], R.e[2][AX]);
    const V3 mid = p + mul(R, c);
    SelfSeg s;
    s.A = mid - h * ax; s.B = mid + h * ax;
    V3 a_loc = c; a_loc.e[AX] -= h;
    s.w = mul(R, v.a);
    s.vA = mul(R, v.l + cross(v.a, a_loc));
    s.r = r;
    return s;
}
// lateral extent of a capsule towards the other leg, as a signed margin: left leg (leg 0, +y side) = its lowest y, right leg = minus its highest y
template <int AX>
BG_HD float self_inner_extent(int leg, const M3& R, V3 p, V3 c, float h, float r, V3 ey, V3 p0) {
    const float ym = dot(ey, p + mul(R, c) - p0);
    const float yh = fabsf(h * (ey.e[0] * R.e[0][AX] + ey.e[1] * R.e[1][AX] + ey.e[2] * R.e[2][AX]));
    return leg == 0 ? ym - yh - r : -(ym + yh + r);
}
BG_HD float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
// own segment o against the partner leg's segment q: world-frame force on the own link and its point of application
BG_HD bool self_pair(const Phys& ph, const SelfSeg& o, const SelfSeg& q, V3* F, V3* x) {
    const V3 d1 = o.B - o.A, d2 = q.B - q.A, r = o.A - q.A;
    const float a = dot(d1, d1), e = dot(d2, d2), b = dot(d1, d2), c = dot(d1, r), f = dot(d2, r);
    const float ap = a * (1.0f + SELF_REG) + 1e-30f, ep = e * (1.0f + SELF_REG) + 1e-30f, cp = c - 0.5f * SELF_REG * a, fp = f + 0.5f * SELF_REG * e;
    float s = clamp01((b * fp - cp * ep) * bg_rcp(ap * ep - b * b));
    float t = (b * s + fp) * bg_rcp(ep);
    if (t < 0.f) { t = 0.f; s = clamp01(-cp * bg_rcp(ap)); }
    else if (t > 1.f) { t = 1.f; s = clamp01((b - cp) * bg_rcp(ap)); }
    const V3 cj = q.A + t * d2, dv = o.A + s * d1 - cj;
    const float dd = dot(dv, dv), rs = o.r + q.r;
    if (!(dd < rs * rs)) return false;
    const float dist = bg_sqrt(dd), pen = rs - dist;
    const V3 n = bg_rcp(dist + 1e-9f) * dv;  // from the partner's link to the own one
    const V3 xc = cj + (q.r - 0.5f * pen) * n;
    const V3 vrel = (o.vA + cross(o.w, xc - o.A)) - (q.vA + cross(q.w, xc - q.A));
    const float vn = dot(vrel, n);
    const float ramp = pen < ph.contact_ramp ? pen * bg_rcp(ph.contact_ramp) : 1.0f;
    const float fn = ph.self_k * pen - ph.self_d * ramp * vn;
    if (!(fn > 0.f)) return false;
    const V3 vt = vrel - vn * n;
    const float c_t = fminf(ph.self_visc, ph.self_mu * fn * bg_rcp(bg_sqrt(dot(vt, vt)) + 1e-6f));
    *F = fn * n - c_t * vt;
    *x = xc;
    return true;
}
BG_HD float sel2(int leg, float left, float right) { return leg == 0 ? left : right; }
struct SelfCaps { V3 cs, cf; float hs, hf, rs, rf; };  // this leg's shank / foot capsule: centre, half length, radius (link coordinates)
BG_HD SelfCaps self_caps(const ModelDev& M, int leg) {
    SelfCaps c;
    for (int a = 0; a < 3; a++) { c.cs.e[a] = sel2(leg, M.cap_c[0][0][a], M.cap_c[1][0][a]); c.cf.e[a] = sel2(leg, M.cap_c[0][1][a], M.cap_c[1][1][a]); }
    c.hs = sel2(leg, M.cap_h[0][0], M.cap_h[1][0]); c.hf = sel2(leg, M.cap_h[0][1], M.cap_h[1][1]);
    c.rs = sel2(leg, M.cap_r[0][0], M.cap_r[1][0]); c.rf = sel2(leg, M.cap_r[0][1], M.cap_r[1][1]);
    return c;
}
// Lateral clearance between the capsules of the two legs along the trunk's y axis (the legs hang on opposite sides of the trunk: capsules can
// only meet when it is negative).  Positions are relative to the trunk origin.  The sum is the same bits on both lanes of the env.
template <class X>
BG_HD float self_clearance(const SelfCaps& c, int leg, const M3& Rsh, V3 psh, const M3& Rft, V3 pft, const M3& R0, X& x) {
    const V3 ey = v3(R0.e[0][1], R0.e[1][1], R0.e[2][1]), o = v3(0.f, 0.f, 0.f);
    const float mine = fminf(self_inner_extent<2>(leg, Rsh, psh, c.cs, c.hs, c.rs, ey, o), self_inner_extent<0>(leg, Rft, pft, c.cf, c.hf, c.rf, ey, o));
    return mine + x.swap(mine);
}
// The narrow phase: both lanes of the env exchange their two segments (26 values) and each evaluates the four pairs for the forces on ITS
// links.  Poses relative to the trunk origin, velocities in link coordinates.  fx[0 / 1] = wrench on the shank / foot about the link origin,
// link coordinates; fw_shank / fw_foot = world-frame forces (contact-force tensor rows, t1.py:219).  Both lanes of an env must call it together.
template <class X>
BG_HD void self_narrow_phase(const Phys& ph, const SelfCaps& c, const M3& Rsh, V3 psh, SV vsh, const M3& Rft, V3 pft, SV vft, X& x, SV* fx,
                             V3* fw_shank, V3* fw_foot) {
    SelfSeg own[2], oth[2];
    own[0] = self_segment<2>(Rsh, psh, vsh, c.cs, c.hs, c.rs);
    own[1] = self_segment<0>(Rft, pft, vft, c.cf, c.hf, c.rf);
    for (int k = 0; k < 2; k++) {
        for (int a = 0; a < 3; a++) {
            oth[k].A.e[a] = x.swap(own[k].A.e[a]); oth[k].B.e[a] = x.swap(own[k].B.e[a]);
            oth[k].w.e[a] = x.swap(own[k].w.e[a]); oth[k].vA.e[a] = x.swap(own[k].vA.e[a]);
        }
        oth[k].r = x.swap(own[k].r);
    }
    for (int k = 0; k < 2; k++) {
        V3 Fs = v3(0.f, 0.f, 0.f), Ts = v3(0.f, 0.f, 0.f);
        const V3 org = k == 0 ? psh : pft;
        for (int l = 0; l < 2; l++) {
            V3 F, xc;
            if (self_pair(ph, own[k], oth[l], &F, &xc)) { Fs = Fs + F; Ts = Ts + cross(xc - org, F); }
        }
        const M3& R = k == 0 ? Rsh : Rft;
        fx[k].l = mulT(R, Fs); fx[k].a = mulT(R, Ts);
        if (k == 0) *fw_shank = Fs; else *fw_foot = Fs;
    }
}
// How a kernel runs the leg-against-leg contacts.  The narrow phase is ~500 instructions that almost no env needs in a given substep; the
// clearance test alone costs the fused env step 1.7 %, the branch around the narrow phase another 7.5 % although it is hardly ever taken
// (register allocation around a large block in the middle of the sweeps), and it makes the two-waves-per-SIMD ABA kernel spill:
//   SELF_INLINE  clearance test and narrow phase between the outward and the inward sweep (fused env step, granular simulate, second kernels)
//   SELF_DEFER   clearance test only; an env whose legs can meet is marked (w.self_deferred) and left to the launch's second kernel (ABA kernel
//                with the trunk-low gate, where a second kernel exists anyway)
// (Also measured for the fused env step: looking for the contacts at the TOP of the substep loop instead, from a kinematics-only walk, only when
// the clearance of the substep before was small: 111.8 us per env step against 107.8 for SELF_INLINE and 98.5 without the contacts.  Not kept.)
//   SELF_LDS     clearance test per lane; the narrow phase ITEM-parallel through LDS (below): what a wave spends on it follows the number of its
//                envs whose legs can meet, not the number of its lanes (ABA kernel)
enum { SELF_INLINE = 0, SELF_DEFER = 1, SELF_LDS = 2 };

#if defined(__HIP_DEVICE_COMPILE__)
// bit 2k of a wave mask -> bit k (the two lanes of an env vote alike).  Wave-uniform: scalar ALU.
__device__ __forceinline__ unsigned even_bits(unsigned long long m) {
    m &= 0x5555555555555555ull;
    m = (m | (m >> 1)) & 0x3333333333333333ull;
    m = (m | (m >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    m = (m | (m >> 4)) & 0x00FF00FF00FF00FFull;
    m = (m | (m >> 8)) & 0x0000FFFF0000FFFFull;
    m = (m | (m >> 16)) & 0x00000000FFFFFFFFull;
    return (unsigned)m;
}
// The narrow phase of a whole wave (one 64-lane workgroup = 32 envs), item-parallel.  In the lane-per-leg form every lane of a wave walks through
// the four pair evaluations as soon as ONE of its 32 envs has its legs close (with 9 % of the envs close that is 93 % of the waves), which is why the
// ABA kernel used to leave such envs to a second kernel that gathered their scattered state again (3.2 x their bytes in fabric traffic).  Here the
// wave compacts instead, in five steps that hand over through LDS, up to PE = 7 close envs per pass (rank among the close envs from the vote mask):
//   P0  the lanes of the close envs put the poses and velocities of their shank and foot into LDS (no arithmetic);
//   P1  lane 4 s + 2 leg + k builds ONE capsule segment (env slot s, link k of that leg);
//   P2  lane 4 s + 2 k0 + k1 evaluates ONE pair (link k0 of the left leg against link k1 of the right leg): one evaluation serves both legs --
//       the force on the right leg's link is minus the one on the left's, at the same point, so momentum is conserved to the bit -- and gives the
//       torques about both links' origins;
//   P3  lane 4 s + 2 leg + k adds the two pairs its link takes part in and rotates force and torque into link coordinates;
//   P4  the owning lanes read their two wrenches (and world-frame forces) back.
// What a wave spends follows the number of its close envs (one instruction stream of ~45 + 145 + 35 VALU per pass) and not its 64 lanes' (the
// lane-per-leg narrow phase is ~500).  One wave per workgroup: the LDS operations of a wave execute in order, so the hand-overs need no barrier,
// only compiler fences.
struct SelfLds {
    static constexpr int PE = SELF_PASS_ENVS, NL = 2 * PE, NS = 4 * PE;
    static constexpr int RAW = 0, SEG = 36 * NL, RES = SEG + 13 * NS, END = RES + 9 * NS;  // the P3 results reuse the segments' space
    static_assert(END <= SELF_LDS_FLOATS, "leg-against-leg scratch");
};
__device__ __forceinline__ void self_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class W>
__device__ __forceinline__ void self_narrow_phase_lds(const Phys& ph, const ModelDev& M, int leg, bool need, W& w, V3 pfoot_rel, SV vfoot) {
    constexpr int PE = SelfLds::PE, NL = SelfLds::NL, NS = SelfLds::NS;
    const unsigned long long vote = __ballot(need);
    if (vote == 0ull) return;  // wave-uniform
    const unsigned envs = even_bits(vote | (vote >> 1));
    const int cnt = __popc(envs);
    const int lane = w.self_lane;
    const int rank = __popc(envs & ((1u << (lane >> 1)) - 1u));
    lds_f32* raw = w.self_sc + SelfLds::RAW;
    lds_f32* seg = w.self_sc + SelfLds::SEG;
    lds_f32* res = w.self_sc + SelfLds::RES;
    const int gs = lane >> 2, gl = (lane >> 1) & 1, gk = lane & 1;  // this lane as worker of P1 / P3 (slot, leg, link) and of P2 (slot, k0, k1)
    for (int base = 0; base < cnt; base += PE) {  // wave-uniform trip count (one pass unless more than PE envs of the wave are close)
        const int slot = rank - base;
        const bool mine = need && slot >= 0 && slot < PE;
        const bool worker = lane < NS && base + gs < cnt;
        BG_PHASE("self_p0_poses");
        if (mine) {
            const int j = 2 * slot + leg;
            const SV vsh = w.st.template get_v<SELF_SHANK>();
            for (int r = 0; r < 3; r++) for (int cidx = 0; cidx < 3; cidx++) { raw[(3 * r + cidx) * NL + j] = w.Rsh.e[r][cidx]; raw[(18 + 3 * r + cidx) * NL + j] = w.Rfoot.e[r][cidx]; }
            for (int a = 0; a < 3; a++) {
                raw[(9 + a) * NL + j] = w.psh.e[a]; raw[(27 + a) * NL + j] = pfoot_rel.e[a];
                raw[(12 + a) * NL + j] = vsh.a.e[a]; raw[(15 + a) * NL + j] = vsh.l.e[a];
                raw[(30 + a) * NL + j] = vfoot.a.e[a]; raw[(33 + a) * NL + j] = vfoot.l.e[a];
            }
        }
        self_lds_fence();
        BG_PHASE("self_p1_segments");
        if (worker) {
            const volatile lds_f32* vr = raw + 18 * gk * NL + 2 * gs + gl;
            M3 R; V3 p; SV v;
            for (int r = 0; r < 3; r++) for (int cidx = 0; cidx < 3; cidx++) R.e[r][cidx] = vr[(3 * r + cidx) * NL];
            for (int a = 0; a < 3; a++) { p.e[a] = vr[(9 + a) * NL]; v.a.e[a] = vr[(12 + a) * NL]; v.l.e[a] = vr[(15 + a) * NL]; }
            const V3 c = v3(M.cap_c[gl][gk][0], M.cap_c[gl][gk][1], M.cap_c[gl][gk][2]);
            const float h = M.cap_h[gl][gk], rad = M.cap_r[gl][gk];
            // the shank's capsule lies along its z axis, the foot's along its x axis
            const V3 ax = gk ? v3(R.e[0][0], R.e[1][0], R.e[2][0]) : v3(R.e[0][2], R.e[1][2], R.e[2][2]);
            const V3 mid = p + mul(R, c);
            const V3 A = mid - h * ax, B = mid + h * ax;
            V3 a_loc = c;
            if (gk) a_loc.e[0] -= h; else a_loc.e[2] -= h;
            const V3 wv = mul(R, v.a), vA = mul(R, v.l + cross(v.a, a_loc));
            for (int a = 0; a < 3; a++) { seg[a * NS + lane] = A.e[a]; seg[(3 + a) * NS + lane] = B.e[a]; seg[(6 + a) * NS + lane] = wv.e[a]; seg[(9 + a) * NS + lane] = vA.e[a]; }
            seg[12 * NS + lane] = rad;
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
            const V3 org_o = v3(po[0], po[NL], po[2 * NL]), org_q = v3(pq[0], pq[NL], pq[2 * NL]);
            V3 F = v3(0.f, 0.f, 0.f), xc = v3(0.f, 0.f, 0.f);
            if (!self_pair(ph, o, q, &F, &xc)) { F = v3(0.f, 0.f, 0.f); xc = org_o; }
            const V3 To = cross(xc - org_o, F), Tq = cross(org_q - xc, F);  // torque of +F about the left link's origin, of -F about the right link's
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
            const volatile lds_f32* f0 = seg + 4 * slot + 2 * leg;
            for (int k = 0; k < 2; k++) {
                w.self_fx[k].l = v3(f0[k], f0[NS + k], f0[2 * NS + k]);
                w.self_fx[k].a = v3(f0[3 * NS + k], f0[4 * NS + k], f0[5 * NS + k]);
                const V3 Fs = v3(f0[6 * NS + k], f0[7 * NS + k], f0[8 * NS + k]);
                if (k == 0) w.self_shank = Fs; else w.self_foot = Fs;
            }
        }
        self_lds_fence();  // the next pass overwrites the scratch
    }
    BG_PHASE("self_done");
}
#endif

template <int MODE, class W, class X>
BG_HD void self_contacts(const Phys& ph, const ModelDev& M, int leg, W& w, const M3& R0, V3 pfoot_rel, SV vfoot, X& x) {
    w.self_fx[0] = sv_zero(); w.self_fx[1] = sv_zero();
    w.self_shank = v3(0.f, 0.f, 0.f); w.self_foot = v3(0.f, 0.f, 0.f);
    w.self_deferred = false;
    w.self_gap = 1.0f;
    if (!ph.self_on) return;
    const SelfCaps c = self_caps(M, leg);
    w.self_gap = self_clearance(c, leg, w.Rsh, w.psh, w.Rfoot, pfoot_rel, R0, x);
    if constexpr (MODE == SELF_DEFER) {
        w.self_deferred = w.self_gap < 0.f;
    } else if constexpr (MODE == SELF_LDS) {
#if defined(__HIP_DEVICE_COMPILE__)
        self_narrow_phase_lds(ph, M, leg, w.self_gap < 0.f, w, pfoot_rel, vfoot);
#endif
    } else if constexpr (MODE == SELF_INLINE) {
#ifndef BG_SELF_NORARE   // (timing experiment: the per-substep part only)
        if (w.self_gap < 0.f)
            self_narrow_phase(ph, c, w.Rsh, w.psh, w.st.template get_v<SELF_SHANK>(), w.Rfoot, pfoot_rel, vfoot, x, w.self_fx, &w.self_shank, &w.self_foot);
#endif
    }
}
// ---------------------------------------------------------------- foot contact (4 sole corners)
// the 0 / 1 factor that removes the half of a two-wide corner evaluation whose foot does not touch; under `if (hit)` a lane-per-leg corner is a hit
BG_HD float hit_factor(bool) { return 1.0f; }
BG_HD f2 hit_factor(B2 h) { return mk2(h.x ? 1.0f : 0.0f, h.y ? 1.0f : 0.0f); }
template <class W, class T = typename W::Scalar>
BG_HD void foot_contact(const Phys& ph, const TerrainDev& tr, const LegParamsT<T>& lp, W& w, SVT<T> vfoot, V3T<T> pfoot, V3T<T>* force_w0) {
    bg_pin(vfoot); bg_pin(pfoot); bg_pin(w.Rfoot);
    BG_PHASE("foot_contact");
    // Everything after the terrain query is done in FOOT coordinates (normal and velocities rotated once; the friction law only needs norms
    // and the normal component, which do not depend on the frame).  With C = alpha nb nb^T + beta 1 (alpha = cn - ctt, beta = ctt) and the
    // corner at r, the impedance B = J^T C J, J = [-rx 1], is
    //     M += alpha nb nb^T + beta 1,   H += alpha m nb^T + beta rx,   A += alpha m m^T + beta (|r|^2 1 - r r^T),   m = r x nb,
    // so a corner costs three rank-1 updates (two of them symmetric) instead of two 3x3 cross-column products; the beta terms are summed
    // over the corners first (sb = sum beta, sr = sum beta r, P = sum beta r r^T) and added once.
    // T = f2: a corner is evaluated when EITHER foot's corner touches; `on` (0 / 1 per half) removes the other foot's share.
    S3T<T> BM = s3t_zero<T>(), BA = s3t_zero<T>(), P = s3t_zero<T>();
    M3T<T> BH = m3t_zero<T>();
    SVT<T> f0 = svt_zero<T>();
    V3T<T> sr = v3_zero<T>();
    T sb = splat<T>(0.f);
    const T zero = splat<T>(0.f), one = splat<T>(1.0f);
    bool any = false;
    if (decltype(w.st)::PLANE_SPEC && tr.type == 0) {
        // Flat ground (wave-uniform; the throughput-bound ABA kernel only, like the z-axis link specialisation): the normal is the world's z axis for
        // every corner, so nb is ONE vector (the third row of Rfoot), a corner's penetration needs its height only (one dot product instead of the
        // world position), and the rank-1 terms that carry nb factor out of the sum over the corners: M += (sum alpha) nb nb^T, H += (sum alpha m) nb^T.
        const V3T<T> nb = v3t<T>(w.Rfoot.e[2][0], w.Rfoot.e[2][1], w.Rfoot.e[2][2]);
        T sa = zero;
        V3T<T> sam = v3_zero<T>();
        for (int k = 0; k < 4; k++) {
            const V3T<T> r = splat_v3<T>(lp.corner[k]);
            const T pen = -(pfoot.e[2] + dot(nb, r));
            const V3T<T> vb = vfoot.l + cross(vfoot.a, r);
            const T vn = dot(vb, nb);
            const T ramp = sel(lt(pen, splat<T>(ph.contact_ramp)), pen * bg_rcp(ph.contact_ramp), one);
            const T d_eff = lp.dn * ramp;
            const T fn0 = lp.kn * pen - d_eff * vn;
            const auto hit = both(gt(pen, zero), gt(fn0, zero));
            if (any_of(hit)) {
                any = true;
                const T on = hit_factor(hit);
                const V3T<T> vt = vb - vn * nb;
                const T fn = on * fn0;
                const T c_t = on * bg_min(splat<T>(ph.friction_visc), lp.mu * fn0 * bg_rcp(bg_sqrt(dot(vt, vt)) + 1e-6f));
                const V3T<T> fb = fn * nb - c_t * vt;
                const T beta = ph.dt * c_t, alpha = on * (ph.dt * (d_eff + ph.dt * lp.kn)) - beta;
                f0.l = f0.l + fb;
                f0.a = f0.a + cross(r, fb);
                const V3T<T> m = cross(r, nb), am = alpha * m, br = beta * r;
                sa += alpha;
                sam = sam + am;
                BA.e[0] += am.e[0] * m.e[0]; BA.e[1] += am.e[1] * m.e[1]; BA.e[2] += am.e[2] * m.e[2];
                BA.e[3] += am.e[0] * m.e[1]; BA.e[4] += am.e[0] * m.e[2]; BA.e[5] += am.e[1] * m.e[2];
                P.e[0] += br.e[0] * r.e[0]; P.e[1] += br.e[1] * r.e[1]; P.e[2] += br.e[2] * r.e[2];
                P.e[3] += br.e[0] * r.e[1]; P.e[4] += br.e[0] * r.e[2]; P.e[5] += br.e[1] * r.e[2];
                sr = sr + br;
                sb += beta;
            }
        }
        const V3T<T> an = sa * nb;
        BM.e[0] = an.e[0] * nb.e[0]; BM.e[1] = an.e[1] * nb.e[1]; BM.e[2] = an.e[2] * nb.e[2];
        BM.e[3] = an.e[0] * nb.e[1]; BM.e[4] = an.e[0] * nb.e[2]; BM.e[5] = an.e[1] * nb.e[2];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) BH.e[i][j] = sam.e[i] * nb.e[j];
    } else
    for (int k = 0; k < 4; k++) {
        const V3T<T> r = splat_v3<T>(lp.corner[k]);
        const V3T<T> xw = pfoot + mul(w.Rfoot, r);
        const V3T<T> vb = vfoot.l + cross(vfoot.a, r);
        T h; V3T<T> n;
        terrain_query(tr, xw.e[0], xw.e[1], &h, &n);
        const V3T<T> nb = mulT(w.Rfoot, n);
        const T pen = (h - xw.e[2]) * n.e[2];
        const T vn = dot(vb, nb);
        const T ramp = sel(lt(pen, splat<T>(ph.contact_ramp)), pen * bg_rcp(ph.contact_ramp), one);
        const T d_eff = lp.dn * ramp;
        const T fn0 = lp.kn * pen - d_eff * vn;
        const auto hit = both(gt(pen, zero), gt(fn0, zero));
        if (any_of(hit)) {
            any = true;
            const T on = hit_factor(hit);
            const V3T<T> vt = vb - vn * nb;
            const T fn = on * fn0;
            const T c_t = on * bg_min(splat<T>(ph.friction_visc), lp.mu * fn0 * bg_rcp(bg_sqrt(dot(vt, vt)) + 1e-6f));
            const V3T<T> fb = fn * nb - c_t * vt;
            const T beta = ph.dt * c_t, alpha = on * (ph.dt * (d_eff + ph.dt * lp.kn)) - beta;
            f0.l = f0.l + fb;
            f0.a = f0.a + cross(r, fb);
            const V3T<T> m = cross(r, nb), an = alpha * nb, am = alpha * m, br = beta * r;
            BM.e[0] += an.e[0] * nb.e[0]; BM.e[1] += an.e[1] * nb.e[1]; BM.e[2] += an.e[2] * nb.e[2];
            BM.e[3] += an.e[0] * nb.e[1]; BM.e[4] += an.e[0] * nb.e[2]; BM.e[5] += an.e[1] * nb.e[2];
            BA.e[0] += am.e[0] * m.e[0]; BA.e[1] += am.e[1] * m.e[1]; BA.e[2] += am.e[2] * m.e[2];
            BA.e[3] += am.e[0] * m.e[1]; BA.e[4] += am.e[0] * m.e[2]; BA.e[5] += am.e[1] * m.e[2];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) BH.e[i][j] += am.e[i] * nb.e[j];
            P.e[0] += br.e[0] * r.e[0]; P.e[1] += br.e[1] * r.e[1]; P.e[2] += br.e[2] * r.e[2];
            P.e[3] += br.e[0] * r.e[1]; P.e[4] += br.e[0] * r.e[2]; P.e[5] += br.e[1] * r.e[2];
            sr = sr + br;
            sb += beta;
        }
    }
    const T trP = P.e[0] + P.e[1] + P.e[2];
    SIT<T> B;
    B.M = BM; B.M.e[0] += sb; B.M.e[1] += sb; B.M.e[2] += sb;
    B.H = BH + skew(sr);
    B.A.e[0] = BA.e[0] + trP - P.e[0]; B.A.e[1] = BA.e[1] + trP - P.e[1]; B.A.e[2] = BA.e[2] + trP - P.e[2];
    B.A.e[3] = BA.e[3] - P.e[3]; B.A.e[4] = BA.e[4] - P.e[4]; B.A.e[5] = BA.e[5] - P.e[5];
    w.Bc = B; w.f0c = f0; w.contact = any;
    *force_w0 = v3_zero<T>();  // (the force that acts over the step comes out of substep_solve: f0 - B a)
}

// ---------------------------------------------------------------- inward sweep, link I (child -> parent)
template <int I, class W, class T = typename W::Scalar>
BG_HD void leg_inward(const Phys& ph, const LegParamsT<T>& lp, const LegStateT<T>& ls, const T* tau, W& w, SIT<T> IA, SVT<T> pA,
                      BaseContributionT<T>* out, const SVT<T>* fext) {
    constexpr int AX = LEG_AXIS[I], A = AX - 1;
    bg_pin(IA); bg_pin(pA);
    BG_PHASE("inward_link");
    // joint limit spring/damper, implicit in the joint velocity: tau_lim = t0 - bl * qdd
    const T zero = splat<T>(0.f);
    const T lo = w.st.template q_lo<I>(lp), hi = w.st.template q_hi<I>(lp);
    const T viol = sel(lt(ls.q[I], lo), ls.q[I] - lo, sel(gt(ls.q[I], hi), ls.q[I] - hi, zero));
    const auto outside = nonzero(viol);
    const T t0 = sel(outside, -ph.limit_k * viol - ph.limit_d * ls.qd[I], zero), bl = sel(outside, splat<T>(ph.dt * (ph.limit_d + ph.dt * ph.limit_k)), zero);
    // U = IA S ; d = S.U + bl ; u = tau - S.pA     (S = the unit angular axis A: U is a column of [A; H^T])
    const M3T<T> Af = full(IA.A);
    SVT<T> U;
    U.a = v3t<T>(Af.e[0][A], Af.e[1][A], Af.e[2][A]);
    U.l = v3t<T>(IA.H.e[A][0], IA.H.e[A][1], IA.H.e[A][2]);
    T d = U.a.e[A] + bl;
    T dinv = bg_rcp(d);
    T u = w.st.template tau_at<I>(tau) + t0 - pA.a.e[A];
    w.dinv[I] = dinv; w.u[I] = u;
    // Ia = IA - U U^T / d ;  pa = pA + Ia c + U u / d.   A and M stay SYMMETRIC through the update, the rotation and the shift: only their six
    // unique entries are computed (the compiler cannot see the symmetry of a full 3x3 and computed all nine)
    const V3T<T> Uad = dinv * U.a, Uld = dinv * U.l;
    S3T<T> A1, M1;
    A1.e[0] = IA.A.e[0] - Uad.e[0] * U.a.e[0]; A1.e[1] = IA.A.e[1] - Uad.e[1] * U.a.e[1]; A1.e[2] = IA.A.e[2] - Uad.e[2] * U.a.e[2];
    A1.e[3] = IA.A.e[3] - Uad.e[0] * U.a.e[1]; A1.e[4] = IA.A.e[4] - Uad.e[0] * U.a.e[2]; A1.e[5] = IA.A.e[5] - Uad.e[1] * U.a.e[2];
    M1.e[0] = IA.M.e[0] - Uld.e[0] * U.l.e[0]; M1.e[1] = IA.M.e[1] - Uld.e[1] * U.l.e[1]; M1.e[2] = IA.M.e[2] - Uld.e[2] * U.l.e[2];
    M1.e[3] = IA.M.e[3] - Uld.e[0] * U.l.e[1]; M1.e[4] = IA.M.e[4] - Uld.e[0] * U.l.e[2]; M1.e[5] = IA.M.e[5] - Uld.e[1] * U.l.e[2];
    M3T<T> H1 = IA.H - outer(Uad, U.l);
    SVT<T> cb = w.st.template get_cb<I>();
    SVT<T> pa;
    pa.a = pA.a + mul(A1, cb.a) + mul(H1, cb.l) + u * Uad;
    pa.l = pA.l + mulT(H1, cb.a) + mul(M1, cb.l) + u * Uld;
    // to parent coordinates: rotate by R(axis,q) then shift the origin by r = pos
    T c = w.c[I], s = w.s[I];
    const T c2 = c * c - s * s, s2 = 2.0f * c * s;
    A1 = rot_conj_sym<AX>(c, s, c2, s2, A1); M1 = rot_conj_sym<AX>(c, s, c2, s2, M1);
    H1 = rot_conj<AX>(c, s, H1);
    const V3T<T> r0 = w.st.template link_pos<I>(lp);
    const M3T<T> M1f = full(M1);
    M3T<T> H2, A2;
    SVT<T> pp;
    pp.l = rot<AX>(c, s, pa.l);
    if ((w.zmask >> I) & 1) {  // r = (0, 0, z): wave-uniform branch (Phys::zmask)
        const V3T<T> r = v3t<T>(zero, zero, r0.e[2]);
        H2 = H1 + cross_cols(r, M1f);
        A2 = full(A1) + cross_cols(r, transpose(H1)) - mul_skew(H2, r);
        pp.a = rot<AX>(c, s, pa.a) + cross(r, pp.l);
    } else {
        const V3T<T> r = r0;
        H2 = H1 + cross_cols(r, M1f);
        A2 = full(A1) + cross_cols(r, transpose(H1)) - mul_skew(H2, r);
        pp.a = rot<AX>(c, s, pa.a) + cross(r, pp.l);
    }
    if constexpr (I > 0) {
        // parent's own rigid inertia and velocity-dependent bias
        const LinkConstT<T> pk = w.st.template link<I - 1>(lp);
        SIT<T> IP = rigid_inertia(pk);
        SVT<T> vp = w.st.template get_v<I - 1>();
        w.st.template put_U<I>(U);
        SVT<T> pP = crf(vp, mul_rigid(pk, vp));
        if (fext) pP = pP - fext[I - 1];
        IP.A = IP.A + upper(A2); IP.H = IP.H + H2; IP.M = IP.M + M1;
        leg_inward<I - 1>(ph, lp, ls, tau, w, IP, pP + pp, out, fext);
    } else {
        w.st.template put_U<I>(U);
        out->I.A = upper(A2); out->I.H = H2; out->I.M = M1;
        out->p = pp;
    }
}

// The second half of a lane's work before the trunk: sole contact, foot inertia and bias, inward sweep (w holds the outward sweep's results and
// the leg-against-leg wrenches).  Returns this leg's (T = f2: both legs') contribution at the trunk.
template <class W, class T = typename W::Scalar>
BG_HD BaseContributionT<T> leg_phase1_inward(const Phys& ph, const TerrainDev& tr, const LegParamsT<T>& lp, const LegStateT<T>& ls, const T* tau, const BaseState& bs,
                                             W& w, SVT<T> vfoot, V3T<T> pfoot_rel, const SVT<T>* fext_in = nullptr) {
    const V3T<T> pfoot = splat_v3<T>(bs.pos) + pfoot_rel;
    V3T<T> unused;
    SVT<T> fext[LEG_LINKS];  // applied wrenches per link: the caller's, plus the leg-against-leg contacts on the shank and the foot
    for (int i = 0; i < LEG_LINKS; i++) fext[i] = fext_in ? fext_in[i] : svt_zero<T>();
    fext[SELF_SHANK] = fext[SELF_SHANK] + w.self_fx[0];
    fext[SELF_FOOT] = fext[SELF_FOOT] + w.self_fx[1];
    foot_contact(ph, tr, lp, w, vfoot, pfoot, &unused);
    bg_pin(w.Bc); bg_pin(w.f0c);
    BG_PHASE("foot_inertia_bias");
    const LinkConstT<T> fk = w.st.template link<LEG_LINKS - 1>(lp);
    SIT<T> IA = rigid_inertia(fk);
    SVT<T> pA = crf(vfoot, mul_rigid(fk, vfoot));
    pA = pA - fext[LEG_LINKS - 1];
    if (w.contact) {
        // f_ext = f0 - B a_true = (f0 - B ag) - B a'   with a' = a_true - ag  (gravity field in foot coords)
        SVT<T> ag; ag.a = v3_zero<T>(); ag.l = mulT(w.Rfoot, splat_v3<T>(ph.g));
        SVT<T> Bag = mul(w.Bc, ag);
        IA.A = IA.A + w.Bc.A; IA.H = IA.H + w.Bc.H; IA.M = IA.M + w.Bc.M;
        pA = pA - (w.f0c - Bag);
    }
    BaseContributionT<T> out;
    leg_inward<LEG_LINKS - 1>(ph, lp, ls, tau, w, IA, pA, &out, fext);
    return out;
}

// Everything a lane does before the pair exchange: kinematics, contact, inward sweep.
// gb = gravity in base coordinates.  Returns this leg's contribution at the trunk.
// fext (optional): applied wrench on each link about its own origin, link coordinates (a = torque, l = force).
template <int SELF, class W, class X, class T = typename W::Scalar>
BG_HD BaseContributionT<T> leg_phase1(const Phys& ph, const TerrainDev& tr, const ModelDev& M, int leg, const LegParamsT<T>& lp, const LegStateT<T>& ls, const T* tau,
                                      const BaseState& bs, M3 R0, SV v0, W& w, X& x, V3T<T>* foot_force_w0, const SVT<T>* fext_in = nullptr) {
    w.zmask = decltype(w.st)::ZSPEC ? ph.zmask : 0;
#ifdef BG_CENSUS_ZMASK   // tools/isa_census.py --t1: count what a T1 launch executes (the wave-uniform branches on zmask resolved at compile time)
    w.zmask = BG_CENSUS_ZMASK;
#endif
    SVT<T> vfoot;
    V3T<T> pfoot_rel;  // link origins are carried RELATIVE to the trunk origin: the leg-against-leg distances must not lose digits to the world position
    leg_outward<0>(lp, ls, w, splat_sv<T>(v0), splat_m3<T>(R0), v3_zero<T>(), &vfoot, &pfoot_rel);
    BG_PHASE("self_clearance");
    self_contacts<SELF>(ph, M, leg, w, R0, pfoot_rel, vfoot, x);
    *foot_force_w0 = v3_zero<T>();  // (the force that acts over the step comes out of substep_solve: f0 - B a)
    return leg_phase1_inward(ph, tr, lp, ls, tau, bs, w, vfoot, pfoot_rel, fext_in);
}

// Trunk: solve  IA0 a0' = -pA0  by block elimination on the 3x3 blocks.
BG_HD SV base_solve(const SI& I, SV p) {
    S3 Minv = inv_sym(I.M);
    // (A - H Minv H^T) alpha = -n + H Minv f
    M3 HMi = mul(I.H, full(Minv));
    M3 Sch = full(I.A) - mul(HMi, transpose(I.H));
    V3 rhs = mul(HMi, p.l) - p.a;
    SV a;
    a.a = mul(inv_sym(upper(Sch)), rhs);
    a.l = -1.0f * mul(Minv, p.l + mulT(I.H, a.a));
    return a;
}

// ---------------------------------------------------------------- outward acceleration sweep + joint integration
template <int I, class W, class T = typename W::Scalar>
BG_HD void leg_accel(const Phys& ph, const LegParamsT<T>& lp, LegStateT<T>& ls, const W& w, SVT<T> apar, T* qdd, SVT<T>* afoot) {
    constexpr int AX = LEG_AXIS[I], A = AX - 1;
    bg_pin(apar);
    BG_PHASE("accel_link");
    T c = w.c[I], s = w.s[I];
    SVT<T> a;
    a.a = rotT<AX>(c, s, apar.a);
    const V3T<T> lpos = w.st.template link_pos<I>(lp);
    V3T<T> alin;
    if ((w.zmask >> I) & 1) alin = apar.l + cross(apar.a, v3t<T>(splat<T>(0.f), splat<T>(0.f), lpos.e[2]));  // r = (0, 0, z): wave-uniform branch (Phys::zmask)
    else alin = apar.l + cross(apar.a, lpos);
    a.l = rotT<AX>(c, s, alin);
    a = a + w.st.template get_cb<I>();
    T qa = (w.u[I] - dot(w.st.template get_U<I>(), a)) * w.dinv[I];
    a.a.e[A] += qa;
    qdd[I] = qa;
    if constexpr (I + 1 < LEG_LINKS) leg_accel<I + 1>(ph, lp, ls, w, a, qdd, afoot);
    else *afoot = a;
}

BG_HD void integrate_leg(const Phys& ph, const LegParams& lp, LegState& ls, const float* qdd) {
    for (int i = 0; i < LEG_LINKS; i++) {
        float qd = ls.qd[i] + ph.dt * qdd[i];
        if (ph.clamp_qd) qd = fminf(fmaxf(qd, -lp.qd_max[i]), lp.qd_max[i]);
        ls.qd[i] = qd;
        ls.q[i] += ph.dt * qd;
    }
}

// true base acceleration -> d/dt of the Isaac-style world velocities
BG_HD void base_world_rates(M3 R0, SV v0, SV a0p, V3 g, V3* lin_w, V3* ang_w) {
    V3 gb = mulT(R0, g);
    V3 lin_b = a0p.l + gb + cross(v0.a, v0.l);  // classical acceleration of the origin
    *lin_w = mul(R0, lin_b);
    *ang_w = mul(R0, a0p.a);
}

BG_HD void integrate_base(const Phys& ph, BaseState& bs, V3 lin_w, V3 ang_w) {
    bs.vlin = bs.vlin + ph.dt * lin_w;
    bs.vang = bs.vang + ph.dt * ang_w;
    bs.pos = bs.pos + ph.dt * bs.vlin;
    // q+ = exp(dt w_world) * q
    V3 wv = bs.vang;
    float w2 = dot(wv, wv);
    float dqx = 0.f, dqy = 0.f, dqz = 0.f, dqw = 1.f;
    if (w2 > 1e-24f) {
        float wn = bg_sqrt(w2);
        float sh, ch;
        bg_sincos(0.5f * wn * ph.dt, &sh, &ch);
        float k = sh * bg_rcp(wn);
        dqx = wv.e[0] * k; dqy = wv.e[1] * k; dqz = wv.e[2] * k; dqw = ch;
    }
    float x = bs.quat[0], y = bs.quat[1], z = bs.quat[2], w = bs.quat[3];
    float nw = dqw * w - dqx * x - dqy * y - dqz * z;
    float nx = dqw * x + dqx * w + dqy * z - dqz * y;
    float ny = dqw * y - dqx * z + dqy * w + dqz * x;
    float nz = dqw * z + dqx * y - dqy * x + dqz * w;
    float inv = bg_rsqrt(nx * nx + ny * ny + nz * nz + nw * nw);
    bs.quat[0] = nx * inv; bs.quat[1] = ny * inv; bs.quat[2] = nz * inv; bs.quat[3] = nw * inv;
}

// PD actuator with joint friction and torque clipping (reference envs/t1.py:446-448)
BG_HD float pd_torque(float kp, float kd, float fric, float limit, float target, float q, float qd) {
    float t = kp * (target - q) - kd * qd;
    float fr = fminf(fric, fabsf(t));
    t -= t > 0.f ? fr : (t < 0.f ? -fr : 0.f);
    return fminf(fmaxf(t, -limit), limit);
}

// Body-coordinate trunk velocity from the world-frame root state
BG_HD SV base_body_velocity(M3 R0, const BaseState& bs) {
    SV v;
    v.a = mulT(R0, bs.vang);
    v.l = mulT(R0, bs.vlin);
    return v;
}

// Trunk's own articulated terms (rigid inertia, velocity bias, applied wrench in base coords)
BG_HD BaseContribution base_own(const LinkConst& bk, SV v0, SV wrench /* a = torque, l = force */) {
    BaseContribution b;
    b.I = rigid_inertia(bk);
    b.p = crf(v0, mul_rigid(bk, v0)) - wrench;
    return b;
}

// A force / torque given in a body's own frame with the force acting at the centre of mass
// (gym.apply_rigid_body_force_tensors(..., LOCAL_SPACE), reference t1.py:522-527) as a wrench about the body origin.
BG_HD SV local_wrench_at_com(const LinkConst& k, V3 f, V3 t) {
    SV w;
    w.l = f;
    w.a = t + bg_rcp(k.m) * cross(k.mc, f);
    return w;
}

// Rigid-body state rows of one leg (Isaac Gym rigid_body_state tensor, t1.py:220): origin position, quaternion xyzw (w >= 0),
// linear velocity of the origin and angular velocity in the world frame; 13 floats per link, link I at out + 13 * I.
template <int I>
BG_HD void leg_body_states(const LegParams& lp, const LegState& ls, SV vpar, M3 Rpar, V3 ppar, float* out) {
    constexpr int AX = LEG_AXIS[I], A = AX - 1;
    float s, c;
    bg_sincos(ls.q[I], &s, &c);
    SV v;
    v.a = rotT<AX>(c, s, vpar.a);
    v.l = rotT<AX>(c, s, vpar.l + cross(vpar.a, lp.lk[I].pos));
    v.a.e[A] += ls.qd[I];
    V3 p = ppar + mul(Rpar, lp.lk[I].pos);
    M3 R = Rpar;
    {
        constexpr int J = Plane<AX>::J, K = Plane<AX>::K;
        for (int r = 0; r < 3; r++) {
            R.e[r][J] = c * Rpar.e[r][J] + s * Rpar.e[r][K];
            R.e[r][K] = -s * Rpar.e[r][J] + c * Rpar.e[r][K];
        }
    }
    float qt[4];
    mat_to_quat(R, qt);
    V3 vl = mul(R, v.l), va = mul(R, v.a);
    float* o = out + 13 * I;
    for (int a = 0; a < 3; a++) { o[a] = p.e[a]; o[7 + a] = vl.e[a]; o[10 + a] = va.e[a]; }
    for (int a = 0; a < 4; a++) o[3 + a] = qt[a];
    if constexpr (I + 1 < LEG_LINKS) leg_body_states<I + 1>(lp, ls, v, R, p, out);
}

// ---------------------------------------------------------------- model constants and per-env parameters

// nominal body + per-env randomisation (mass scale, com offset; inertia scales with the mass as
// Isaac Gym's recomputeInertia=True does, reference t1.py:129-131) -> constants about the body origin
template <class T>
BG_HD LinkConstT<T> make_link_t(V3T<T> pos, T mass, V3T<T> com, const T* inertia /*[6] about the centre of mass*/, T mass_scale, V3T<T> com_off) {
    LinkConstT<T> k;
    k.pos = pos;
    k.m = mass * mass_scale;
    V3T<T> c = com + com_off;
    k.mc = k.m * c;
    T cc = dot(c, c);
    k.Io.e[0] = inertia[0] * mass_scale + k.m * (cc - c.e[0] * c.e[0]);
    k.Io.e[1] = inertia[1] * mass_scale + k.m * (cc - c.e[1] * c.e[1]);
    k.Io.e[2] = inertia[2] * mass_scale + k.m * (cc - c.e[2] * c.e[2]);
    k.Io.e[3] = inertia[3] * mass_scale - k.m * c.e[0] * c.e[1];
    k.Io.e[4] = inertia[4] * mass_scale - k.m * c.e[0] * c.e[2];
    k.Io.e[5] = inertia[5] * mass_scale - k.m * c.e[1] * c.e[2];
    return k;
}
BG_HD LinkConst make_link(const ModelDev& m, int b, float mass_scale, V3 com_off) {
    return make_link_t<float>(v3(m.pos[b][0], m.pos[b][1], m.pos[b][2]), m.mass[b], v3(m.com[b][0], m.com[b][1], m.com[b][2]), m.inertia[b], mass_scale, com_off);
}

struct ContactCfg { float k, d, terrain_mu, terrain_restitution; };

// per-env parameter arrays are SoA: value(field f, env e) = p[f * n + e]
BG_HD void load_leg_params(const ModelDev& m, const ContactCfg& cc, int leg, int e, int n, const float* mass_scale /*[13][n]*/,
                           const float* com_off /*[39][n]*/, const float* foot_mat /*[6][n]*/, LegParams& lp) {
    for (int i = 0; i < LEG_LINKS; i++) {
        int b = 1 + leg * LEG_LINKS + i, j = leg * LEG_LINKS + i;
        lp.lk[i] = make_link(m, b, mass_scale[(size_t)b * n + e],
                             v3(com_off[(size_t)(3 * b) * n + e], com_off[(size_t)(3 * b + 1) * n + e], com_off[(size_t)(3 * b + 2) * n + e]));
        lp.q_lo[i] = m.q_lo[j]; lp.q_hi[i] = m.q_hi[j]; lp.qd_max[i] = m.qd_max[j];
    }
    for (int k = 0; k < 4; k++) lp.corner[k] = v3(m.corner[k][0], m.corner[k][1], m.corner[k][2]);
    float mu_f = foot_mat[(size_t)(3 * leg) * n + e], compl_f = foot_mat[(size_t)(3 * leg + 1) * n + e], rest_f = foot_mat[(size_t)(3 * leg + 2) * n + e];
    lp.mu = 0.5f * (mu_f + cc.terrain_mu);  // PhysX default material combine: average
    lp.kn = cc.k * bg_rcp(compl_f);
    lp.dn = cc.d * (1.0f - 0.5f * (rest_f + cc.terrain_restitution));
}
BG_HD LinkConst load_base_link(const ModelDev& m, int e, int n, const float* mass_scale, const float* com_off) {
    return make_link(m, 0, mass_scale[e], v3(com_off[e], com_off[(size_t)n + e], com_off[(size_t)2 * n + e]));
}

// ---------------------------------------------------------------- one substep, split around the lane-pair exchange
template <class W>
struct SubstepCtxT { M3 R0; SV v0; W w; };
using SubstepCtx = SubstepCtxT<LegWork>;
using SubstepCtxLdsLink = SubstepCtxT<LegWorkT<LdsLinkStore>>;

// World-frame contact forces of the non-foot bodies of one lane's half of the env (net_contact_force rows; reward `collision`, t1.py:627-629)
struct BodyContactOut {
    bool active;          // the trunk was low enough for the spheres to be evaluated
    V3 trunk;             // this lane's share of the trunk's force (sum the two lanes); zero when not active
    V3 link[LEG_LINKS];   // this leg's links: terrain contacts of their spheres (when active) + the shank's share of the leg-against-leg contacts;
                          // the foot's entry stays zero (its force comes out of substep_solve)
};

// Are the non-foot body contacts evaluated for this env?  Only while the trunk origin is less than ph.body_gate above the terrain: from a
// standing or walking posture those shapes cannot reach the ground.  The callers test this ONCE per launch and per env (the fused env step:
// at the start of the env step) and run one of two complete instantiations of their lane code, so that the common path carries no trace of
// the body-contact code (a runtime branch around it inside the substep loop cost 30 % of the env step in register pressure: measured).
BG_HD bool body_contacts_active(const Phys& ph, const TerrainDev& tr, const ModelDev& M, const V3& base_pos) {
    return M.sph_n > 0 && base_pos.e[2] - terrain_height(tr, base_pos.e[0], base_pos.e[1]) < ph.body_gate;
}

// fext (optional): applied wrench on each link about its own origin, link coordinates.  BODY: also evaluate the contact spheres of the
// non-foot bodies (M / leg: model constants and which leg this lane owns); their wrenches join `fext`, the trunk's share is subtracted from
// this lane's contribution at the trunk.
template <bool BODY, int SELF = SELF_INLINE, class Ctx, class X>
BG_HD BaseContribution substep_pre(const Phys& ph, const TerrainDev& tr, const ModelDev& M, int leg, const LegParams& lp, const LegState& ls,
                                   const float* tau, const BaseState& bs, Ctx& cx, X& x, const SV* fext = nullptr, BodyContactOut* bo = nullptr) {
    BG_PHASE("base_kinematics");
    cx.R0 = quat_to_mat(bs.quat);
    cx.v0 = base_body_velocity(cx.R0, bs);
    V3 unused;
    if (bo) bo->active = BODY;
    if constexpr (!BODY) {
        BaseContribution out = leg_phase1<SELF>(ph, tr, M, leg, lp, ls, tau, bs, cx.R0, cx.v0, cx.w, x, &unused, fext);
        if (bo) {
            bo->trunk = v3(0.f, 0.f, 0.f);
            for (int i = 0; i < LEG_LINKS; i++) bo->link[i] = v3(0.f, 0.f, 0.f);
            bo->link[SELF_SHANK] = cx.w.self_shank;
        }
        return out;
    } else {
        SV fx[LEG_LINKS];
        V3 fw[LEG_LINKS];
        for (int i = 0; i < LEG_LINKS; i++) { fx[i] = fext ? fext[i] : sv_zero(); fw[i] = v3(0.f, 0.f, 0.f); }
        leg_contact_prepass<0>(ph, tr, M, 1 + leg * LEG_LINKS, cx.w.st, lp, ls, cx.v0, cx.R0, bs.pos, fx, fw);
        V3 ft = v3(0.f, 0.f, 0.f);
        const SV wt = body_contact_wrench(ph, tr, M, 0, leg, cx.R0, bs.pos, cx.v0, &ft);
        BaseContribution out = leg_phase1<SELF>(ph, tr, M, leg, lp, ls, tau, bs, cx.R0, cx.v0, cx.w, x, &unused, fx);
        if (bo) {
            bo->trunk = ft;
            for (int i = 0; i < LEG_LINKS; i++) bo->link[i] = fw[i];
            bo->link[SELF_SHANK] = bo->link[SELF_SHANK] + cx.w.self_shank;
        }
        out.p = out.p - wt;
        return out;
    }
}
// net contact force on the foot in the world frame: terrain (the force that acts over the step: f0 - B a_true) + the other leg
template <class W, class T = typename W::Scalar>
BG_HD V3T<T> foot_force_over_step(const Phys& ph, const W& w, SVT<T> afoot) {
    V3T<T> fw = v3_zero<T>();
    if (w.contact) {
        SVT<T> at = afoot; at.l = at.l + mulT(w.Rfoot, splat_v3<T>(ph.g));
        V3T<T> fb = w.f0c.l - (mulT(w.Bc.H, at.a) + mul(w.Bc.M, at.l));
        fw = mul(w.Rfoot, fb);
    }
    return fw + w.self_foot;
}
// `both` = this leg's contribution + the partner leg's.  Returns accelerations; does not integrate.
template <class Ctx>
BG_HD void substep_solve(const Phys& ph, const LinkConst& bk, const LegParams& lp, LegState& ls, const Ctx& cx, const BaseContribution& both,
                         SV wrench, float* qdd, V3* lin_w, V3* ang_w, V3* foot_force_w) {
    BG_PHASE("base_solve");
    BaseContribution own = base_own(bk, cx.v0, wrench);
    SI I; I.A = own.I.A + both.I.A; I.H = own.I.H + both.I.H; I.M = own.I.M + both.I.M;
    SV a0p = base_solve(I, own.p + both.p);
    SV afoot;
    leg_accel<0>(ph, lp, ls, cx.w, a0p, qdd, &afoot);
    bg_pin(afoot); bg_pin(a0p);
    BG_PHASE("rates_and_foot_force");
    base_world_rates(cx.R0, cx.v0, a0p, ph.g, lin_w, ang_w);
    *foot_force_w = foot_force_over_step(ph, cx.w, afoot);
}
BG_HD void substep_integrate(const Phys& ph, const LegParams& lp, LegState& ls, BaseState& bs, const float* qdd, V3 lin_w, V3 ang_w) {
    integrate_leg(ph, lp, ls, qdd);
    integrate_base(ph, bs, lin_w, ang_w);
}

}  // namespace bg
