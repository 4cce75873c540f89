// Host-only part of the C ABI: the thread-local error message behind bg_last_error, and the model object (bg_model_create / get / destroy and the
// topology checks shared with the URDF loader).  No HIP here: together with bg_urdf.cpp this file also builds with g++ and the address / undefined-
// behaviour sanitizers (tests/test_sanitizers.py; GPU sanitizers are not available on the pool).
#include <string>

#include "bg_model.h"

static thread_local std::string g_err;
int bg_set_error(int code, const char* msg) { g_err = msg ? msg : "error"; return code; }
static int fail(int code, const std::string& msg) { return bg_set_error(code, msg.c_str()); }
extern "C" const char* bg_last_error(void) { return g_err.c_str(); }
extern "C" const char* bg_version(void) { return "booster_gym_amd 0.1 (gfx950)"; }

static const int kLegAxis[6] = {2, 1, 3, 2, 2, 1};

int bg_model_validate(const bg_model_desc* d) {
    if (d->num_bodies != BG_NUM_BODIES || d->num_dofs != BG_NUM_DOFS)
        return fail(-1, "bg_model: this build supports the 13-body / 12-DoF collapsed T1 topology only (got " + std::to_string(d->num_bodies) + " bodies, " +
                            std::to_string(d->num_dofs) + " DoFs)");
    for (int leg = 0; leg < 2; leg++)
        for (int i = 0; i < 6; i++) {
            int b = 1 + leg * 6 + i;
            int want_parent = i == 0 ? 0 : b - 1;
            if (d->parent[b] != want_parent || d->joint_axis[b] != kLegAxis[i])
                return fail(-1, "bg_model: body " + std::to_string(b) + " does not match the T1 leg chain (parent/axis)");
        }
    if (d->parent[0] != -1 || d->joint_axis[0] != 0) return fail(-1, "bg_model: body 0 must be the floating base");
    for (int b = 0; b < BG_NUM_BODIES; b++)
        if (!(d->mass[b] > 0.f)) return fail(-1, "bg_model: non-positive mass on body " + std::to_string(b));
    if (d->num_body_spheres < 0 || d->num_body_spheres > BG_MAX_BODY_SPHERES) return fail(-1, "bg_model: num_body_spheres out of range");
    for (int k = 0; k < d->num_body_spheres; k++) {
        const int b = d->sphere_body[k];
        if (b < 0 || b >= BG_NUM_BODIES || b == 6 || b == 12) return fail(-1, "bg_model: contact spheres belong to the trunk or a non-foot leg link");
        if (k > 0 && b < d->sphere_body[k - 1]) return fail(-1, "bg_model: contact spheres must be sorted by body");
        if (!(d->sphere_radius[k] >= 0.f)) return fail(-1, "bg_model: negative sphere radius");
    }
    for (int leg = 0; leg < 2; leg++)
        for (int k = 0; k < 2; k++) {
            if (!(d->self_capsule_r[leg][k] >= 0.f)) return fail(-1, "bg_model: negative self-collision capsule radius");
            const int ax = k == 0 ? 2 : 0;  // shank capsule along z, foot capsule along x (the per-lane code is written for these axes)
            for (int a = 0; a < 3; a++)
                if (a != ax && d->self_capsule_a[leg][k][a] != d->self_capsule_b[leg][k][a])
                    return fail(-1, "bg_model: self-collision capsules must lie along z (shank) / x (foot) of their link");
        }
    return 0;
}
extern "C" int bg_model_create(const bg_model_desc* d, bg_model** out) {
    if (!d || !out) return fail(-1, "bg_model_create: null argument");
    if (int rc = bg_model_validate(d)) return rc;
    bg_model* m = new bg_model;
    m->desc = *d;
    *out = m;
    return 0;
}
extern "C" int bg_model_get(const bg_model* m, bg_model_desc* out) {
    if (!m || !out) return fail(-1, "bg_model_get: null argument");
    *out = m->desc;
    return 0;
}
extern "C" void bg_model_destroy(bg_model* m) { delete m; }
