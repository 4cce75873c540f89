/*
 * oracle/dyn_ref.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Plain-C, double-precision restatement of the physics substep that the HIP
 * kernels in booster_gym_amd/csrc implement.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this file's shared object.
 *
 * What it restates.  The reference delegates physics to third-party binaries
 * that are absent from /root/reference: Isaac Gym Preview 4 / PhysX
 * (envs/t1.py:450-455 `gym.simulate`) for training and MuJoCo `mj_step`
 * (play_mujoco.py:756) for cross-simulation.  Neither can be built or
 * imported here, and the reference holds no test that pins any of their
 * outputs => PARITY UNPINNED for the dynamics.  This
 * file therefore restates the *published* algorithms on the reference's own
 * model constants (resources/T1/T1_locomotion.xml:36-139, the collapsed URDF):
 *   - floating-base articulated-body algorithm, R. Featherstone, "Rigid Body
 *     Dynamics Algorithms" (2008), Table 9.4, dense 6x6 spatial algebra
 *   - an independent recursive Newton-Euler inverse dynamics (Table 9.6) used
 *     by the tests to verify the ABA result  (ABA == inverse of RNEA)
 *   - semi-implicit Euler as in MuJoCo's default integrator
 *     (v+ = v + dt*qacc ; q+ = q (+) dt*v+), step structure play_mujoco.py:751-756
 *   - PD actuator with latency / joint friction / clipping: envs/t1.py:443-456
 *   - bilinear terrain height: utils/terrain.py:101-121
 * The contact / joint-limit model (linearly-implicit penalty forces folded into
 * the articulated inertia) is this build's own choice; DESIGN.md section 4 states it.
 *
 * Everything here is written for clarity: dense matrices, generic tree,
 * no shared code with the product kernels.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
/* Second build of the SAME source in single precision (libdynref32.so: `-DREF_F32 -fsingle-precision-constant`, Makefile): every
 * `double` below, the ABI structs and arrays included, becomes float and the libm calls their float entry points.  It serves
 * (1) the parity tests, as the measure of how far fp32 rounding ALONE moves this very algorithm on a given state (tests/test_gpu_env.py:
 * an env may deviate from the float64 result only as far as this build does, times a fixed factor), and (2) bench.py's cpu_baseline
 * leg (the CPU baseline computes in fp32 like the reference's PhysX-CPU path).  Parity proper stays on the double build. */
#ifdef REF_F32
#define double float
#define sqrt sqrtf
#define cos cosf
#define sin sinf
#define fabs fabsf
#define floor floorf
#define fmin fminf
#define REF_TINY 1e-30f
#else
#define REF_TINY 1e-300
#endif

#define NB 13
#define ND 12

typedef struct {
    int32_t nb;
    int32_t parent[NB];
    int32_t axis[NB]; /* 0 base, 1 x, 2 y, 3 z */
    double pos[NB][3];
    double mass[NB];
    double com[NB][3];
    double inertia[NB][6]; /* xx yy zz xy xz yz about com */
    double q_lower[ND], q_upper[ND], qd_limit[ND];
    int32_t foot_body[2];
    double foot_corner[4][3];
    /* contact spheres of the non-foot collision shapes (URDF <collision>: trunk box corners with radius 0, cylinder end caps) */
    int32_t n_sph;
    int32_t sph_body[16];
    double sph_pos[16][3];
    double sph_r[16];
    /* self-collision capsules (asset.self_collisions, envs/T1.yaml:69; create_actor(..., self_collisions) envs/t1.py:128): sphere-swept segments
     * standing in for the shank cylinders and the foot boxes of the URDF <collision> elements, in link coordinates */
    int32_t n_cap;
    int32_t cap_body[4];
    double cap_a[4][3], cap_b[4][3], cap_r[4];
} ref_model_t;

typedef struct {
    double dt;
    double g[3];
    double contact_k;     /* N/m per corner */
    double contact_d;     /* N s/m per corner */
    double contact_ramp;  /* m: damping ramps in over this penetration */
    double friction_visc; /* N s/m: stick regularisation */
    double limit_k, limit_d;
    double terrain_mu, terrain_restitution;
    int32_t clamp_qd;
    int32_t pad;
    double body_gate_height; /* m: the spheres above are evaluated only while the trunk origin is lower than this above the terrain */
    /* leg-against-leg contacts: explicit penalty between the capsules of the left and of the right leg */
    double self_k, self_d, self_mu, self_visc; /* N/m, N s/m, Coulomb coefficient, N s/m cap of the regularised friction */
    int32_t self_collisions;                   /* 1 = modelled (asset.self_collisions: 0 in the yaml, Isaac Gym's "no filter") */
    int32_t pad2;
} ref_phys_t;

typedef struct {
    int32_t type; /* 0 plane, 1 heightfield */
    int32_t rows, cols, border_px;
    double hscale, vscale;
    const int16_t *hf; /* [rows][cols] */
} ref_terrain_t;

/* ------------------------------------------------------------------ small linear algebra */
static void m3_mul(const double a[3][3], const double b[3][3], double c[3][3]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a[i][k] * b[k][j];
            c[i][j] = s;
        }
}
static void m3_vec(const double a[3][3], const double v[3], double o[3]) {
    for (int i = 0; i < 3; i++) o[i] = a[i][0] * v[0] + a[i][1] * v[1] + a[i][2] * v[2];
}
static void m3t_vec(const double a[3][3], const double v[3], double o[3]) {
    for (int i = 0; i < 3; i++) o[i] = a[0][i] * v[0] + a[1][i] * v[1] + a[2][i] * v[2];
}
static void skew(const double v[3], double s[3][3]) {
    s[0][0] = 0; s[0][1] = -v[2]; s[0][2] = v[1];
    s[1][0] = v[2]; s[1][1] = 0; s[1][2] = -v[0];
    s[2][0] = -v[1]; s[2][1] = v[0]; s[2][2] = 0;
}
static void cross(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static void m6_vec(const double a[6][6], const double v[6], double o[6]) {
    for (int i = 0; i < 6; i++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += a[i][k] * v[k];
        o[i] = s;
    }
}
static void m6t_vec(const double a[6][6], const double v[6], double o[6]) {
    for (int i = 0; i < 6; i++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s += a[k][i] * v[k];
        o[i] = s;
    }
}
/* xyzw quaternion -> rotation matrix (body -> world) */
static void quat_to_mat(const double q[4], double r[3][3]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    r[0][0] = 1 - 2 * (y * y + z * z); r[0][1] = 2 * (x * y - z * w); r[0][2] = 2 * (x * z + y * w);
    r[1][0] = 2 * (x * y + z * w); r[1][1] = 1 - 2 * (x * x + z * z); r[1][2] = 2 * (y * z - x * w);
    r[2][0] = 2 * (x * z - y * w); r[2][1] = 2 * (y * z + x * w); r[2][2] = 1 - 2 * (x * x + y * y);
}
/* rotation of the child frame relative to the parent: R(axis, q) */
static void axis_rot(int axis, double q, double r[3][3]) {
    double c = cos(q), s = sin(q);
    memset(r, 0, 9 * sizeof(double));
    if (axis == 1) { r[0][0] = 1; r[1][1] = c; r[1][2] = -s; r[2][1] = s; r[2][2] = c; }
    else if (axis == 2) { r[1][1] = 1; r[0][0] = c; r[0][2] = s; r[2][0] = -s; r[2][2] = c; }
    else { r[2][2] = 1; r[0][0] = c; r[0][1] = -s; r[1][0] = s; r[1][1] = c; }
}
/* Pluecker motion transform parent->child for child frame rotated by Rpc (child axes in
 * parent coords) and displaced by r:  X = [E 0; -E rx  E],  E = Rpc^T   (RBDA eq. 2.24-2.26) */
static void make_X(const double Rpc[3][3], const double r[3], double X[6][6]) {
    double E[3][3], rx[3][3], Erx[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) E[i][j] = Rpc[j][i];
    skew(r, rx);
    m3_mul(E, rx, Erx);
    memset(X, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            X[i][j] = E[i][j];
            X[i + 3][j + 3] = E[i][j];
            X[i + 3][j] = -Erx[i][j];
        }
}
/* spatial cross products, RBDA eq. 2.31-2.32 */
static void crm(const double v[6], const double m[6], double o[6]) {
    double t[3], u[3];
    cross(v, m, o);          /* w x m_ang */
    cross(v, m + 3, t);      /* w x m_lin */
    cross(v + 3, m, u);      /* v x m_ang */
    for (int i = 0; i < 3; i++) o[i + 3] = t[i] + u[i];
}
static void crf(const double v[6], const double f[6], double o[6]) {
    double t[3], u[3];
    cross(v, f, t);          /* w x n */
    cross(v + 3, f + 3, u);  /* v x f */
    for (int i = 0; i < 3; i++) o[i] = t[i] + u[i];
    cross(v, f + 3, o + 3);  /* w x f */
}
/* spatial inertia about the body origin from (m, c, Ic): RBDA eq. 2.63 */
static void make_inertia(double m, const double c[3], const double ic6[6], double I[6][6]) {
    double cx[3][3], cxT[3][3], cc[3][3];
    skew(c, cx);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cxT[i][j] = cx[j][i];
    m3_mul(cx, cxT, cc);
    double Ic[3][3] = {{ic6[0], ic6[3], ic6[4]}, {ic6[3], ic6[1], ic6[5]}, {ic6[4], ic6[5], ic6[2]}};
    memset(I, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            I[i][j] = Ic[i][j] + m * cc[i][j];
            I[i][j + 3] = m * cx[i][j];
            I[i + 3][j] = m * cxT[i][j];
        }
    for (int i = 0; i < 3; i++) I[i + 3][i + 3] = m;
}

/* ------------------------------------------------------------------ terrain (utils/terrain.py:101-121) */
static void terrain_query(const ref_terrain_t *t, double x, double y, double *h, double n[3]) {
    if (t->type == 0) { *h = 0; n[0] = 0; n[1] = 0; n[2] = 1; return; }
    double px = t->border_px + x / t->hscale, py = t->border_px + y / t->hscale;
    int x1 = (int)floor(px), y1 = (int)floor(py);
    /* the reference has no bounds check (negative indices alias in numpy); clamp instead */
    if (x1 < 0) x1 = 0; if (x1 > t->rows - 2) x1 = t->rows - 2;
    if (y1 < 0) y1 = 0; if (y1 > t->cols - 2) y1 = t->cols - 2;
    double fx = px - x1, fy = py - y1;
    double h00 = t->hf[x1 * t->cols + y1], h10 = t->hf[(x1 + 1) * t->cols + y1];
    double h01 = t->hf[x1 * t->cols + y1 + 1], h11 = t->hf[(x1 + 1) * t->cols + y1 + 1];
    *h = ((1 - fx) * (1 - fy) * h00 + fx * (1 - fy) * h10 + (1 - fx) * fy * h01 + fx * fy * h11) * t->vscale;
    double hx = ((1 - fy) * (h10 - h00) + fy * (h11 - h01)) * t->vscale / t->hscale;
    double hy = ((1 - fx) * (h01 - h00) + fx * (h11 - h10)) * t->vscale / t->hscale;
    double inv = 1.0 / sqrt(hx * hx + hy * hy + 1.0);
    n[0] = -hx * inv; n[1] = -hy * inv; n[2] = inv;
}
double ref_terrain_height(const ref_terrain_t *t, double x, double y) {
    double h, n[3];
    terrain_query(t, x, y, &h, n);
    return h;
}

/* ------------------------------------------------------------------ kinematics shared by ABA and RNEA */
typedef struct {
    double X[NB][6][6];    /* parent -> body motion transform */
    double Rw[NB][3][3];   /* body -> world rotation */
    double pw[NB][3];      /* body origin in world */
    double v[NB][6];       /* spatial velocity, body coords */
    double c[NB][6];       /* velocity-product acceleration */
    double I[NB][6][6];    /* rigid-body inertia with per-env randomisation */
    double ag[NB][6];      /* gravitational acceleration field in body coords */
} kin_t;

static void kinematics(const ref_model_t *m, const ref_phys_t *p, const double *mass_scale, const double *com_off,
                       const double *root, const double *q, const double *qd, kin_t *k) {
    double R0[3][3];
    quat_to_mat(root + 3, R0);
    memcpy(k->Rw[0], R0, sizeof(R0));
    memcpy(k->pw[0], root, 3 * sizeof(double));
    /* base spatial velocity in base coords from Isaac-style world velocities (t1.py:221-222) */
    m3t_vec(R0, root + 10, k->v[0]);
    m3t_vec(R0, root + 7, k->v[0] + 3);
    memset(k->c[0], 0, 6 * sizeof(double));
    for (int i = 0; i < m->nb; i++) {
        double ms = mass_scale ? mass_scale[i] : 1.0;
        double c[3], ic[6];
        for (int a = 0; a < 3; a++) c[a] = m->com[i][a] + (com_off ? com_off[3 * i + a] : 0.0);
        for (int a = 0; a < 6; a++) ic[a] = m->inertia[i][a] * ms; /* recomputeInertia=True: t1.py:131 */
        make_inertia(m->mass[i] * ms, c, ic, k->I[i]);
    }
    for (int i = 1; i < m->nb; i++) {
        int par = m->parent[i];
        double Rpc[3][3];
        axis_rot(m->axis[i], q[i - 1], Rpc);
        make_X(Rpc, m->pos[i], k->X[i]);
        m3_mul(k->Rw[par], Rpc, k->Rw[i]);
        double t[3];
        m3_vec(k->Rw[par], m->pos[i], t);
        for (int a = 0; a < 3; a++) k->pw[i][a] = k->pw[par][a] + t[a];
        double vj[6] = {0, 0, 0, 0, 0, 0};
        vj[m->axis[i] - 1] = qd[i - 1];
        m6_vec(k->X[i], k->v[par], k->v[i]);
        for (int a = 0; a < 6; a++) k->v[i][a] += vj[a];
        crm(k->v[i], vj, k->c[i]);
    }
    for (int i = 0; i < m->nb; i++) {
        memset(k->ag[i], 0, 3 * sizeof(double));
        m3t_vec(k->Rw[i], p->g, k->ag[i] + 3);
    }
}

/* ------------------------------------------------------------------ contact: linearly-implicit penalty.
 * For every sole corner with penetration: f = f0 - B * a_body, where a_body is the (true) spatial
 * acceleration of the foot.  f0 is the force at the current state, B = dt * J^T C J collects the
 * stiffness/damping that act on the end-of-step velocity.  Both are expressed in foot coordinates. */
typedef struct {
    double B[2][6][6];
    double f0[2][6];
    double fw[2][3]; /* world-frame net force at the current state (for contact_forces tensor) */
} contact_t;

static void contacts(const ref_model_t *m, const ref_phys_t *p, const ref_terrain_t *t, const double *foot_mat, const kin_t *k,
                     contact_t *ct) {
    memset(ct, 0, sizeof(*ct));
    for (int f = 0; f < 2; f++) {
        int b = m->foot_body[f];
        double mu_foot = foot_mat ? foot_mat[3 * f + 0] : 1.0;
        double compliance = foot_mat ? foot_mat[3 * f + 1] : 1.0;
        double rest_foot = foot_mat ? foot_mat[3 * f + 2] : 0.0;
        double mu = 0.5 * (mu_foot + p->terrain_mu);                    /* PhysX default combine: average */
        double e = 0.5 * (rest_foot + p->terrain_restitution);
        double kn = p->contact_k / compliance;
        double dn = p->contact_d * (1.0 - e);
        for (int cidx = 0; cidx < 4; cidx++) {
            const double *r = m->foot_corner[cidx];
            double rw[3], xw[3], vb[3], t1[3], vw[3];
            m3_vec(k->Rw[b], r, rw);
            for (int a = 0; a < 3; a++) xw[a] = k->pw[b][a] + rw[a];
            cross(k->v[b], r, t1);
            for (int a = 0; a < 3; a++) vb[a] = k->v[b][3 + a] + t1[a];
            m3_vec(k->Rw[b], vb, vw);
            double h, n[3];
            terrain_query(t, xw[0], xw[1], &h, n);
            double pen = (h - xw[2]) * n[2];
            if (pen <= 0) continue;
            double vn = vw[0] * n[0] + vw[1] * n[1] + vw[2] * n[2];
            double ramp = pen < p->contact_ramp ? pen / p->contact_ramp : 1.0;
            double d_eff = dn * ramp;
            double fn0 = kn * pen - d_eff * vn;
            if (fn0 <= 0) continue;
            double vt[3];
            for (int a = 0; a < 3; a++) vt[a] = vw[a] - vn * n[a];
            double vtn = sqrt(vt[0] * vt[0] + vt[1] * vt[1] + vt[2] * vt[2]);
            double c_t = p->friction_visc;
            double cap = mu * fn0 / (vtn + 1e-6);
            if (cap < c_t) c_t = cap;
            double fw[3];
            for (int a = 0; a < 3; a++) fw[a] = fn0 * n[a] - c_t * vt[a];
            for (int a = 0; a < 3; a++) ct->fw[f][a] += fw[a];
            /* to foot coords */
            double nb[3], fb[3];
            m3t_vec(k->Rw[b], n, nb);
            m3t_vec(k->Rw[b], fw, fb);
            double cn = p->dt * (d_eff + p->dt * kn), ctt = p->dt * c_t;
            double C[3][3];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) C[i][j] = (cn - ctt) * nb[i] * nb[j] + (i == j ? ctt : 0.0);
            /* J = [-rx 1]; wrench = J^T f ; B = J^T C J */
            double J[3][6];
            double rx[3][3];
            skew(r, rx);
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) { J[i][j] = -rx[i][j]; J[i][j + 3] = (i == j); }
            for (int i = 0; i < 6; i++) {
                for (int a = 0; a < 3; a++) ct->f0[f][i] += J[a][i] * fb[a];
                for (int j = 0; j < 6; j++) {
                    double s = 0;
                    for (int a = 0; a < 3; a++) for (int bb = 0; bb < 3; bb++) s += J[a][i] * C[a][bb] * J[bb][j];
                    ct->B[f][i][j] += s;
                }
            }
        }
    }
}

/* Non-foot bodies (trunk box, hip-yaw and shank cylinders): explicit penalty contact of their spheres, same normal / friction law as the
 * sole corners but evaluated at the current state only (no implicit term: these links are heavy enough for dt = 2 ms), default material
 * (friction 1, restitution 0) averaged with the terrain's.  Evaluated only while the trunk is low (body_gate_height): the spheres cannot
 * reach the ground from a standing or walking posture.  wrench[b] = [torque about the body origin; force] in body coords, fw[b] = world force. */
/* -1: the gate is evaluated on the state at hand (single-substep entry points); 0 / 1: decided by the caller for a whole env step
 * (ref_substeps: from the trunk height at the START of the env step, as the fused HIP env step does) */
static __thread int g_body_gate = -1;
static int body_gate_low(const ref_model_t *m, const ref_phys_t *p, const ref_terrain_t *t, const double *root) {
    return m->n_sph > 0 && root[2] - ref_terrain_height(t, root[0], root[1]) < p->body_gate_height;
}
static void body_contacts(const ref_model_t *m, const ref_phys_t *p, const ref_terrain_t *t, const double *root, const kin_t *k,
                          double wrench[NB][6], double fw_body[NB][3]) {
    memset(wrench, 0, sizeof(double) * NB * 6);
    memset(fw_body, 0, sizeof(double) * NB * 3);
    if (!(g_body_gate >= 0 ? g_body_gate : body_gate_low(m, p, t, root))) return;
    const double mu = 0.5 * (1.0 + p->terrain_mu), kn = p->contact_k, dn = p->contact_d * (1.0 - 0.5 * p->terrain_restitution);
    for (int s = 0; s < m->n_sph; s++) {
        const int b = m->sph_body[s];
        const double *c = m->sph_pos[s], r = m->sph_r[s];
        double cw[3], xw[3], h, n[3];
        m3_vec(k->Rw[b], c, cw);
        for (int a = 0; a < 3; a++) xw[a] = k->pw[b][a] + cw[a];
        terrain_query(t, xw[0], xw[1], &h, n);
        const double pen = (h - xw[2]) * n[2] + r;
        if (pen <= 0) continue;
        /* contact point = sphere centre - r n, in body coords rc = c - r R^T n */
        double nb[3], rc[3], t1[3], vb[3], vw[3];
        m3t_vec(k->Rw[b], n, nb);
        for (int a = 0; a < 3; a++) rc[a] = c[a] - r * nb[a];
        cross(k->v[b], rc, t1);
        for (int a = 0; a < 3; a++) vb[a] = k->v[b][3 + a] + t1[a];
        m3_vec(k->Rw[b], vb, vw);
        const double vn = vw[0] * n[0] + vw[1] * n[1] + vw[2] * n[2];
        const double ramp = pen < p->contact_ramp ? pen / p->contact_ramp : 1.0;
        const double fn0 = kn * pen - dn * ramp * vn;
        if (fn0 <= 0) continue;
        double vt[3];
        for (int a = 0; a < 3; a++) vt[a] = vw[a] - vn * n[a];
        const double vtn = sqrt(vt[0] * vt[0] + vt[1] * vt[1] + vt[2] * vt[2]);
        double c_t = p->friction_visc;
        const double cap = mu * fn0 / (vtn + 1e-6);
        if (cap < c_t) c_t = cap;
        double f_w[3], fb[3], tq[3];
        for (int a = 0; a < 3; a++) f_w[a] = fn0 * n[a] - c_t * vt[a];
        m3t_vec(k->Rw[b], f_w, fb);
        cross(rc, fb, tq);
        for (int a = 0; a < 3; a++) { wrench[b][a] += tq[a]; wrench[b][3 + a] += fb[a]; fw_body[b][a] += f_w[a]; }
    }
}

/* ------------------------------------------------------------------ self-collision: leg against leg.
 * Every capsule of a left-leg link against every capsule of a right-leg link (the reference enables PhysX self-collision, envs/T1.yaml:69,
 * envs/t1.py:128; links of the same chain never meet).  Closest points of the two segments, penetration = r1 + r2 - distance, explicit
 * penalty force along the line of centres with regularised Coulomb friction, applied with opposite signs to the two bodies at ONE point
 * (the middle of the overlap), so linear and angular momentum are conserved exactly.
 *
 * Closest points: minimise |P1(s) - P2(t)|^2 + SELF_REG (a (s - 1/2)^2 + e (t - 1/2)^2) over the unit square, a, e = squared segment lengths.
 * The small quadratic term makes the minimiser unique and continuous when the segments are parallel (feet side by side, shanks side by
 * side: the unregularised problem has a whole interval of minimisers there and the contact point would jump along the link under
 * rounding-size changes); it moves the closest distance only to second order.  Solved exactly as for the plain problem (clamp s, solve t,
 * clamp t, re-solve s: C. Ericson, Real-Time Collision Detection, 5.1.9) with a' = a (1 + REG), e' = e (1 + REG), c' = c - a REG / 2,
 * f' = f + e REG / 2. */
#define SELF_REG 1e-2
static double clamp01(double x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
static void segment_closest(const double A1[3], const double B1[3], const double A2[3], const double B2[3], double *s_out, double *t_out) {
    double d1[3], d2[3], r[3];
    for (int a = 0; a < 3; a++) { d1[a] = B1[a] - A1[a]; d2[a] = B2[a] - A2[a]; r[a] = A1[a] - A2[a]; }
    double a = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2], e = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
    double b = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
    double c = d1[0] * r[0] + d1[1] * r[1] + d1[2] * r[2], f = d2[0] * r[0] + d2[1] * r[1] + d2[2] * r[2];
    double ap = a * (1 + SELF_REG) + REF_TINY, ep = e * (1 + SELF_REG) + REF_TINY, cp = c - 0.5 * SELF_REG * a, fp = f + 0.5 * SELF_REG * e;
    double denom = ap * ep - b * b;
    double s = clamp01((b * fp - cp * ep) / denom);
    double t = (b * s + fp) / ep;
    if (t < 0) { t = 0; s = clamp01(-cp / ap); }
    else if (t > 1) { t = 1; s = clamp01((b - cp) / ap); }
    *s_out = s; *t_out = t;
}
/* test hook: the regularised closest points of two segments given as 12 numbers (A1, B1, A2, B2) */
void ref_segment_closest(const double *seg, double *st) { segment_closest(seg, seg + 3, seg + 6, seg + 9, st, st + 1); }
/* spatial velocity of body b (body coords) -> world-frame velocity of the body-fixed point at world position x */
static void point_velocity_w(const kin_t *k, int b, const double x[3], double vw[3]) {
    double rel[3], rb[3], wxr[3], vb[3];
    for (int a = 0; a < 3; a++) rel[a] = x[a] - k->pw[b][a];
    m3t_vec(k->Rw[b], rel, rb);
    cross(k->v[b], rb, wxr);
    for (int a = 0; a < 3; a++) vb[a] = k->v[b][3 + a] + wxr[a];
    m3_vec(k->Rw[b], vb, vw);
}
static void self_contacts(const ref_model_t *m, const ref_phys_t *p, const kin_t *k, double wrench[NB][6], double fw_body[NB][3]) {
    if (!p->self_collisions) return;
    for (int i = 0; i < m->n_cap; i++)
        for (int j = i + 1; j < m->n_cap; j++) {
            const int bi = m->cap_body[i], bj = m->cap_body[j];
            if ((bi <= 6) == (bj <= 6)) continue; /* same leg (bodies 1..6 = left chain, 7..12 = right chain) */
            double Ai[3], Bi[3], Aj[3], Bj[3], t3[3];
            m3_vec(k->Rw[bi], m->cap_a[i], t3); for (int a = 0; a < 3; a++) Ai[a] = k->pw[bi][a] + t3[a];
            m3_vec(k->Rw[bi], m->cap_b[i], t3); for (int a = 0; a < 3; a++) Bi[a] = k->pw[bi][a] + t3[a];
            m3_vec(k->Rw[bj], m->cap_a[j], t3); for (int a = 0; a < 3; a++) Aj[a] = k->pw[bj][a] + t3[a];
            m3_vec(k->Rw[bj], m->cap_b[j], t3); for (int a = 0; a < 3; a++) Bj[a] = k->pw[bj][a] + t3[a];
            double s, t, ci[3], cj[3], dv[3];
            segment_closest(Ai, Bi, Aj, Bj, &s, &t);
            for (int a = 0; a < 3; a++) { ci[a] = Ai[a] + s * (Bi[a] - Ai[a]); cj[a] = Aj[a] + t * (Bj[a] - Aj[a]); dv[a] = ci[a] - cj[a]; }
            const double d2 = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2], rs = m->cap_r[i] + m->cap_r[j];
            if (d2 >= rs * rs) continue;
            const double dist = sqrt(d2), pen = rs - dist;
            double n[3], x[3], vi[3], vj[3];
            const double inv = 1.0 / (dist + 1e-9); /* coincident axes: no direction, (almost) no force */
            for (int a = 0; a < 3; a++) { n[a] = dv[a] * inv; x[a] = cj[a] + n[a] * (m->cap_r[j] - 0.5 * pen); } /* n points from j to i */
            point_velocity_w(k, bi, x, vi);
            point_velocity_w(k, bj, x, vj);
            double vrel[3], vt[3];
            for (int a = 0; a < 3; a++) vrel[a] = vi[a] - vj[a];
            const double vn = vrel[0] * n[0] + vrel[1] * n[1] + vrel[2] * n[2];
            const double ramp = pen < p->contact_ramp ? pen / p->contact_ramp : 1.0;
            const double fn = p->self_k * pen - p->self_d * ramp * vn;
            if (fn <= 0) continue;
            for (int a = 0; a < 3; a++) vt[a] = vrel[a] - vn * n[a];
            const double vtn = sqrt(vt[0] * vt[0] + vt[1] * vt[1] + vt[2] * vt[2]);
            double c_t = p->self_visc;
            const double cap = p->self_mu * fn / (vtn + 1e-6);
            if (cap < c_t) c_t = cap;
            double F[3]; /* on body i; body j gets -F */
            for (int a = 0; a < 3; a++) F[a] = fn * n[a] - c_t * vt[a];
            for (int side = 0; side < 2; side++) {
                const int b = side ? bj : bi;
                const double sg = side ? -1.0 : 1.0;
                double rel[3], rb[3], fb[3], tq[3];
                for (int a = 0; a < 3; a++) rel[a] = x[a] - k->pw[b][a];
                m3t_vec(k->Rw[b], rel, rb);
                m3t_vec(k->Rw[b], F, fb);
                for (int a = 0; a < 3; a++) fb[a] *= sg;
                cross(rb, fb, tq);
                for (int a = 0; a < 3; a++) { wrench[b][a] += tq[a]; wrench[b][3 + a] += fb[a]; fw_body[b][a] += sg * F[a]; }
            }
        }
}

/* test hook: world-frame forces of the leg-against-leg contacts alone on every body, for one state: out [NB][3] */
void ref_self_contact_forces(const ref_model_t *m, const ref_phys_t *p, const double *root, const double *q, const double *qd, double *out) {
    static __thread kin_t k;
    static __thread double wr[NB][6];
    kinematics(m, p, 0, 0, root, q, qd, &k);
    memset(wr, 0, sizeof(wr));
    memset(out, 0, sizeof(double) * NB * 3);
    self_contacts(m, p, &k, wr, (double(*)[3])out);
}

/* joint-limit spring/damper, implicit in the joint velocity:  tau = t0 - bl * qdd */
static void joint_limits(const ref_model_t *m, const ref_phys_t *p, const double *q, const double *qd, double *t0, double *bl) {
    for (int j = 0; j < ND; j++) {
        double viol = 0;
        if (q[j] < m->q_lower[j]) viol = q[j] - m->q_lower[j];
        else if (q[j] > m->q_upper[j]) viol = q[j] - m->q_upper[j];
        if (viol != 0) {
            t0[j] = -p->limit_k * viol - p->limit_d * qd[j];
            bl[j] = p->dt * (p->limit_d + p->dt * p->limit_k);
        } else { t0[j] = 0; bl[j] = 0; }
    }
}

static int solve6(double A[6][6], double b[6]) {
    /* Gaussian elimination with partial pivoting (A is SPD here, pivoting is belt and braces) */
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (fabs(A[piv][c]) < REF_TINY) return -1;
        if (piv != c) {
            for (int k = 0; k < 6; k++) { double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
            double t = b[c]; b[c] = b[piv]; b[piv] = t;
        }
        for (int r = c + 1; r < 6; r++) {
            double f = A[r][c] / A[c][c];
            for (int k = c; k < 6; k++) A[r][k] -= f * A[c][k];
            b[r] -= f * b[c];
        }
    }
    for (int r = 5; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < 6; k++) s -= A[r][k] * b[k];
        b[r] = s / A[r][r];
    }
    return 0;
}

/* ------------------------------------------------------------------ forward dynamics (RBDA Table 9.4)
 * root: pos3, quat xyzw, lin vel world, ang vel world (Isaac Gym root-state layout, t1.py:221-222)
 * base_wrench_local: force(3) then torque(3) in base coords, applied at the base origin (t1.py:522-527)
 * qacc out: d/dt of (lin vel world, ang vel world, qd)   [18]
 * a_body out (optional): true spatial acceleration of every body, body coords [NB*6]
 */
static int forward_core(const ref_model_t *m, const ref_phys_t *p, const ref_terrain_t *t, const double *mass_scale, const double *com_off,
                        const double *foot_mat, const double *root, const double *q, const double *qd, const double *tau,
                        const double *base_wrench_local, const double *body_force, const double *body_torque, double *qacc,
                        double *contact_force_w, double *a_body) {
    kin_t k;
    contact_t ct;
    kinematics(m, p, mass_scale, com_off, root, q, qd, &k);
    contacts(m, p, t, foot_mat, &k, &ct);
    double tl0[ND], bl[ND];
    joint_limits(m, p, q, qd, tl0, bl);

    static __thread double IA[NB][6][6], pA[NB][6], U[NB][6], d[NB], u[NB];
    for (int i = 0; i < m->nb; i++) {
        memcpy(IA[i], k.I[i], sizeof(IA[i]));
        double Iv[6];
        m6_vec(k.I[i], k.v[i], Iv);
        crf(k.v[i], Iv, pA[i]);
    }
    /* external forces enter as  pA -= f_ext  with a' = a - ag as the unknown */
    if (base_wrench_local) {
        for (int a = 0; a < 3; a++) { pA[0][a] -= base_wrench_local[3 + a]; pA[0][3 + a] -= base_wrench_local[a]; }
    }
    /* per-body force / torque in the body's own frame, force acting at the body's centre of mass
     * (gym.apply_rigid_body_force_tensors(..., LOCAL_SPACE), envs/t1.py:522-527): wrench about the origin = (t + c x f, f) */
    if (body_force || body_torque) {
        for (int i = 0; i < m->nb; i++) {
            double f[3] = {0, 0, 0}, tq[3] = {0, 0, 0}, c[3], cxf[3];
            if (body_force) for (int a = 0; a < 3; a++) f[a] = body_force[3 * i + a];
            if (body_torque) for (int a = 0; a < 3; a++) tq[a] = body_torque[3 * i + a];
            for (int a = 0; a < 3; a++) c[a] = m->com[i][a] + (com_off ? com_off[3 * i + a] : 0.0);
            cross(c, f, cxf);
            for (int a = 0; a < 3; a++) { pA[i][a] -= tq[a] + cxf[a]; pA[i][3 + a] -= f[a]; }
        }
    }
    static __thread double bw[NB][6], bfw[NB][3];
    body_contacts(m, p, t, root, &k, bw, bfw);
    self_contacts(m, p, &k, bw, bfw);
    for (int i = 0; i < m->nb; i++) for (int a = 0; a < 6; a++) pA[i][a] -= bw[i][a];
    for (int f = 0; f < 2; f++) {
        int b = m->foot_body[f];
        double Bag[6];
        m6_vec(ct.B[f], k.ag[b], Bag);
        for (int i = 0; i < 6; i++) {
            pA[b][i] -= ct.f0[f][i] - Bag[i];
            for (int j = 0; j < 6; j++) IA[b][i][j] += ct.B[f][i][j];
        }
    }
    for (int i = m->nb - 1; i >= 1; i--) {
        int ax = m->axis[i] - 1, par = m->parent[i];
        for (int a = 0; a < 6; a++) U[i][a] = IA[i][a][ax];
        d[i] = U[i][ax] + bl[i - 1];
        u[i] = tau[i - 1] + tl0[i - 1] - pA[i][ax];
        double Ia[6][6], pa[6], Iac[6];
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) Ia[a][b] = IA[i][a][b] - U[i][a] * U[i][b] / d[i];
        m6_vec(Ia, k.c[i], Iac);
        for (int a = 0; a < 6; a++) pa[a] = pA[i][a] + Iac[a] + U[i][a] * u[i] / d[i];
        /* IA[par] += X^T Ia X ; pA[par] += X^T pa */
        double T[6][6];
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
            double s = 0;
            for (int c = 0; c < 6; c++) s += Ia[a][c] * k.X[i][c][b];
            T[a][b] = s;
        }
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
            double s = 0;
            for (int c = 0; c < 6; c++) s += k.X[i][c][a] * T[c][b];
            IA[par][a][b] += s;
        }
        double Xtp[6];
        m6t_vec(k.X[i], pa, Xtp);
        for (int a = 0; a < 6; a++) pA[par][a] += Xtp[a];
    }
    double A0[6][6], ap[NB][6];
    memcpy(A0, IA[0], sizeof(A0));
    for (int a = 0; a < 6; a++) ap[0][a] = -pA[0][a];
    if (solve6(A0, ap[0])) return -1;
    double qdd[ND];
    for (int i = 1; i < m->nb; i++) {
        int ax = m->axis[i] - 1, par = m->parent[i];
        m6_vec(k.X[i], ap[par], ap[i]);
        for (int a = 0; a < 6; a++) ap[i][a] += k.c[i][a];
        double s = 0;
        for (int a = 0; a < 6; a++) s += U[i][a] * ap[i][a];
        qdd[i - 1] = (u[i] - s) / d[i];
        ap[i][ax] += qdd[i - 1];
    }
    /* true base acceleration, then Isaac-style world-frame rates */
    double a0[6];
    for (int a = 0; a < 6; a++) a0[a] = ap[0][a] + k.ag[0][a];
    double wxv[3], lin_b[3];
    cross(k.v[0], k.v[0] + 3, wxv);
    for (int a = 0; a < 3; a++) lin_b[a] = a0[3 + a] + wxv[a]; /* classical acceleration of the origin */
    m3_vec(k.Rw[0], lin_b, qacc);
    m3_vec(k.Rw[0], a0, qacc + 3);
    memcpy(qacc + 6, qdd, sizeof(qdd));
    if (a_body)
        for (int i = 0; i < m->nb; i++) for (int a = 0; a < 6; a++) a_body[6 * i + a] = ap[i][a] + k.ag[i][a];
    if (contact_force_w) {
        memcpy(contact_force_w, bfw, NB * 3 * sizeof(double)); /* explicit contacts: body spheres against the terrain, leg against leg */
        for (int f = 0; f < 2; f++) {
            /* report the force that actually acted over the step: f0 - B a, rotated to world */
            int b = m->foot_body[f];
            double at[6], Ba[6], fb[3];
            for (int a = 0; a < 6; a++) at[a] = ap[b][a] + k.ag[b][a];
            m6_vec(ct.B[f], at, Ba);
            for (int a = 0; a < 3; a++) fb[a] = ct.f0[f][3 + a] - Ba[3 + a];
            double fww[3];
            m3_vec(k.Rw[b], fb, fww);
            for (int a = 0; a < 3; a++) contact_force_w[3 * b + a] += fww[a]; /* + the foot's share of the leg-against-leg contacts */
        }
    }
    return 0;
}

int ref_forward(const ref_model_t *m, const ref_phys_t *p, const ref_terrain_t *t, const double *mass_scale, const double *com_off,
                const double *foot_mat, const double *root, const double *q, const double *qd, const double *tau,
                const double *base_wrench_local, double *qacc, double *contact_force_w, double *a_body) {
    return forward_core(m, p, t, mass_scale, com_off, foot_mat, root, q, qd, tau, base_wrench_local, 0, 0, qacc, contact_force_w, a_body);
}
/* same with a force / torque on every body (local frame, force at the centre of mass): the Isaac Gym call of t1.py:522-527 */
int ref_forward_bw(const ref_model_t *m, const ref_phys_t *p, const ref_terrain_t *t, const double *mass_scale, const double *com_off,
                   const double *foot_mat, const double *root, const double *q, const double *qd, const double *tau,
                   const double *body_force, const double *body_torque, double *qacc, double *contact_force_w) {
    return forward_core(m, p, t, mass_scale, com_off, foot_mat, root, q, qd, tau, 0, body_force, body_torque, qacc, contact_force_w, 0);
}

/* ------------------------------------------------------------------ inverse dynamics (RBDA Table 9.6), independent check.
 * Given the true body accelerations implied by qacc, returns the generalized force residual
 *   res[0:6]  = net spatial force on the base (must equal the applied base wrench [torque; force])
 *   res[6:18] = joint torques required (must equal tau + limit torque)
 * Contact/limit forces are NOT included: call with contacts disabled (robot airborne, joints inside limits). */
int ref_inverse(const ref_model_t *m, const ref_phys_t *p, const double *mass_scale, const double *com_off, const double *root,
                const double *q, const double *qd, const double *qacc, double *res) {
    kin_t k;
    kinematics(m, p, mass_scale, com_off, root, q, qd, &k);
    double a[NB][6], f[NB][6];
    /* base spatial acceleration in body coords from world-frame rates */
    double lin_b[3], wxv[3];
    m3t_vec(k.Rw[0], qacc + 3, a[0]);
    m3t_vec(k.Rw[0], qacc, lin_b);
    cross(k.v[0], k.v[0] + 3, wxv);
    for (int i = 0; i < 3; i++) a[0][3 + i] = lin_b[i] - wxv[i];
    for (int i = 0; i < 6; i++) a[0][i] -= k.ag[0][i];
    for (int i = 1; i < m->nb; i++) {
        m6_vec(k.X[i], a[m->parent[i]], a[i]);
        for (int c = 0; c < 6; c++) a[i][c] += k.c[i][c];
        a[i][m->axis[i] - 1] += qacc[6 + i - 1];
    }
    for (int i = 0; i < m->nb; i++) {
        double Ia[6], Iv[6], vIv[6];
        m6_vec(k.I[i], a[i], Ia);
        m6_vec(k.I[i], k.v[i], Iv);
        crf(k.v[i], Iv, vIv);
        for (int c = 0; c < 6; c++) f[i][c] = Ia[c] + vIv[c];
    }
    for (int i = m->nb - 1; i >= 1; i--) {
        res[6 + i - 1] = f[i][m->axis[i] - 1];
        double Xtf[6];
        m6t_vec(k.X[i], f[i], Xtf);
        for (int c = 0; c < 6; c++) f[m->parent[i]][c] += Xtf[c];
    }
    for (int c = 0; c < 6; c++) res[c] = f[0][c];
    return 0;
}

/* world pose of every body: pos [NB][3], rot [NB][9] (row-major body->world); used by the task oracle for the feet */
void ref_body_poses(const ref_model_t *m, const double *root, const double *q, double *pos, double *rot) {
    ref_phys_t p;
    memset(&p, 0, sizeof(p));
    double qd[ND] = {0};
    kin_t k;
    kinematics(m, &p, 0, 0, root, q, qd, &k);
    for (int i = 0; i < m->nb; i++) {
        for (int a = 0; a < 3; a++) pos[3 * i + a] = k.pw[i][a];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) rot[9 * i + 3 * a + b] = k.Rw[i][a][b];
    }
}

/* ------------------------------------------------------------------ one substep: semi-implicit Euler */
static void quat_mul(const double a[4], const double b[4], double o[4]) { /* xyzw */
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
}

static void integrate(const ref_model_t *m, const ref_phys_t *p, double *root, double *q, double *qd, const double *qacc);

int ref_step(const ref_model_t *m, const ref_phys_t *p, const ref_terrain_t *t, const double *mass_scale, const double *com_off,
             const double *foot_mat, double *root, double *q, double *qd, const double *tau, const double *base_wrench_local,
             double *contact_force_w) {
    double qacc[18];
    if (ref_forward(m, p, t, mass_scale, com_off, foot_mat, root, q, qd, tau, base_wrench_local, qacc, contact_force_w, 0)) return -1;
    integrate(m, p, root, q, qd, qacc);
    return 0;
}
/* one gym.simulate (t1.py:451) with per-body applied forces */
int ref_step_bw(const ref_model_t *m, const ref_phys_t *p, const ref_terrain_t *t, const double *mass_scale, const double *com_off,
                const double *foot_mat, double *root, double *q, double *qd, const double *tau, const double *body_force,
                const double *body_torque, double *contact_force_w) {
    double qacc[18];
    if (ref_forward_bw(m, p, t, mass_scale, com_off, foot_mat, root, q, qd, tau, body_force, body_torque, qacc, contact_force_w)) return -1;
    integrate(m, p, root, q, qd, qacc);
    return 0;
}

static void integrate(const ref_model_t *m, const ref_phys_t *p, double *root, double *q, double *qd, const double *qacc) {
    for (int a = 0; a < 6; a++) root[7 + a] += p->dt * qacc[a];
    for (int j = 0; j < ND; j++) {
        qd[j] += p->dt * qacc[6 + j];
        if (p->clamp_qd) {
            if (qd[j] > m->qd_limit[j]) qd[j] = m->qd_limit[j];
            if (qd[j] < -m->qd_limit[j]) qd[j] = -m->qd_limit[j];
        }
        q[j] += p->dt * qd[j];
    }
    for (int a = 0; a < 3; a++) root[a] += p->dt * root[7 + a];
    /* orientation: q+ = exp(dt * w_world) * q  (world-frame angular velocity, Isaac convention) */
    double w[3] = {root[10], root[11], root[12]};
    double ang = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]) * p->dt;
    double dq[4] = {0, 0, 0, 1};
    if (ang > 1e-12) {
        double s = sin(0.5 * ang) / (ang / p->dt);
        dq[0] = w[0] * s; dq[1] = w[1] * s; dq[2] = w[2] * s; dq[3] = cos(0.5 * ang);
    }
    double nq[4];
    quat_mul(dq, root + 3, nq);
    double nn = 1.0 / sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
    for (int a = 0; a < 4; a++) root[3 + a] = nq[a] * nn;
}

/* rigid-body state tensor row per body (t1.py:220): origin position, orientation quaternion xyzw (w >= 0), linear velocity of the
 * origin and angular velocity, all in the world frame.  out [NB][13] */
void ref_body_states(const ref_model_t *m, const double *root, const double *q, const double *qd, double *out) {
    ref_phys_t p;
    memset(&p, 0, sizeof(p));
    kin_t k;
    kinematics(m, &p, 0, 0, root, q, qd, &k);
    for (int i = 0; i < m->nb; i++) {
        double *o = out + 13 * i;
        for (int a = 0; a < 3; a++) o[a] = k.pw[i][a];
        const double(*R)[3] = k.Rw[i];
        double tr = R[0][0] + R[1][1] + R[2][2], qx, qy, qz, qw;
        if (tr > 0) {
            double s = sqrt(tr + 1.0) * 2; qw = 0.25 * s; qx = (R[2][1] - R[1][2]) / s; qy = (R[0][2] - R[2][0]) / s; qz = (R[1][0] - R[0][1]) / s;
        } else if (R[0][0] > R[1][1] && R[0][0] > R[2][2]) {
            double s = sqrt(1.0 + R[0][0] - R[1][1] - R[2][2]) * 2; qw = (R[2][1] - R[1][2]) / s; qx = 0.25 * s; qy = (R[0][1] + R[1][0]) / s; qz = (R[0][2] + R[2][0]) / s;
        } else if (R[1][1] > R[2][2]) {
            double s = sqrt(1.0 + R[1][1] - R[0][0] - R[2][2]) * 2; qw = (R[0][2] - R[2][0]) / s; qx = (R[0][1] + R[1][0]) / s; qy = 0.25 * s; qz = (R[1][2] + R[2][1]) / s;
        } else {
            double s = sqrt(1.0 + R[2][2] - R[0][0] - R[1][1]) * 2; qw = (R[1][0] - R[0][1]) / s; qx = (R[0][2] + R[2][0]) / s; qy = (R[1][2] + R[2][1]) / s; qz = 0.25 * s;
        }
        if (qw < 0) { qx = -qx; qy = -qy; qz = -qz; qw = -qw; }
        o[3] = qx; o[4] = qy; o[5] = qz; o[6] = qw;
        m3_vec(R, k.v[i] + 3, o + 7);
        m3_vec(R, k.v[i], o + 10);
    }
}

/* the same for n envs (env-major arrays), OpenMP over envs: the task oracle's feet state and the CPU baseline */
void ref_body_states_batch(const ref_model_t *m, int n, const double *root, const double *q, const double *qd, double *out) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n; e++) ref_body_states(m, root + 13 * e, q + ND * e, qd + ND * e, out + (size_t)NB * 13 * e);
}

/* ------------------------------------------------------------------ decimation loop with the PD actuator
 * (envs/t1.py:443-456).  targets: this step's dof targets; last_targets in/out; delay: switch-over substep.
 * torques_mean out = mean clipped torque over the substeps (t1.py:449,456).  The base wrench acts on the
 * first substep only (Isaac Gym clears applied forces after each simulate). */
int ref_substeps(const ref_model_t *m, const ref_phys_t *p, const ref_terrain_t *t, int decimation, const double *mass_scale,
                 const double *com_off, const double *foot_mat, const double *kp, const double *kd, const double *fric,
                 const double *tau_limit, double *root, double *q, double *qd, const double *targets, double *last_targets,
                 int delay, const double *base_wrench_local, double *torques_mean, double *contact_force_w) {
    for (int j = 0; j < ND; j++) torques_mean[j] = 0;
    g_body_gate = body_gate_low(m, p, t, root);
    int rc = 0;
    for (int s = 0; s < decimation && !rc; s++) {
        double tau[ND];
        if (delay == s) for (int j = 0; j < ND; j++) last_targets[j] = targets[j];
        for (int j = 0; j < ND; j++) {
            double tq = kp[j] * (last_targets[j] - q[j]) - kd[j] * qd[j];
            double fr = fmin(fric[j], fabs(tq)) * (tq > 0 ? 1.0 : (tq < 0 ? -1.0 : 0.0));
            tq -= fr;
            if (tq > tau_limit[j]) tq = tau_limit[j];
            if (tq < -tau_limit[j]) tq = -tau_limit[j];
            tau[j] = tq;
            torques_mean[j] += tq;
        }
        if (ref_step(m, p, t, mass_scale, com_off, foot_mat, root, q, qd, tau, s == 0 ? base_wrench_local : 0, contact_force_w)) rc = -1;
    }
    g_body_gate = -1;
    for (int j = 0; j < ND; j++) torques_mean[j] /= decimation;
    return rc;
}

/* batch driver used by the CPU baseline (OpenMP over environments) */
int ref_substeps_batch(const ref_model_t *m, const ref_phys_t *p, const ref_terrain_t *t, int decimation, int n,
                       const double *mass_scale, const double *com_off, const double *foot_mat, const double *kp, const double *kd,
                       const double *fric, const double *tau_limit, double *root, double *q, double *qd, const double *targets,
                       double *last_targets, const int32_t *delay, const double *base_wrench_local, double *torques_mean,
                       double *contact_force_w) {
    int err = 0;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n; e++) {
        int r = ref_substeps(m, p, t, decimation, mass_scale + (size_t)e * NB, com_off + (size_t)e * NB * 3, foot_mat + (size_t)e * 6,
                             kp + (size_t)e * ND, kd + (size_t)e * ND, fric + (size_t)e * ND, tau_limit, root + (size_t)e * 13,
                             q + (size_t)e * ND, qd + (size_t)e * ND, targets + (size_t)e * ND, last_targets + (size_t)e * ND, delay[e],
                             base_wrench_local + (size_t)e * 6, torques_mean + (size_t)e * ND, contact_force_w + (size_t)e * NB * 3);
        if (r) err = r;
    }
    return err;
}
