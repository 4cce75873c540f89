// Counter-based RNG for per-env noise: Philox4x32-10 (Salmon et al., "Parallel random numbers:
// as easy as 1, 2, 3", SC'11).  Every draw is addressed by (seed, env, step, stream), so results
// do not depend on launch geometry or evaluation order.  The reference instead consumes torch's global
// generator in Python call order (utils/utils.py:11,15), which cannot be reproduced bit-for-bit;
// parity for stochastic quantities is therefore against the oracle's restatement of THIS generator
// (oracle/task_ref.py) and distributional against the reference.
#pragma once
#include "bg_math.h"

namespace bg {

struct Rand4 { float u[4]; float n[4]; };

BG_HD void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

BG_HD float u32_to_unit(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// 4 uniforms in (0,1) and 4 standard normals (Box-Muller on the same bits)
BG_HD Rand4 rand4(uint64_t seed, uint32_t env, uint32_t step, uint32_t stream) {
    uint32_t o[4];
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), env, step, stream, 0u, o);
    Rand4 r;
    for (int i = 0; i < 4; i++) r.u[i] = u32_to_unit(o[i]);
    for (int i = 0; i < 2; i++) {
        float rad = sqrtf(-2.0f * logf(r.u[2 * i]));
        float s, c;
        bg_sincos(6.283185307179586f * r.u[2 * i + 1], &s, &c);
        r.n[2 * i] = rad * c;
        r.n[2 * i + 1] = rad * s;
    }
    return r;
}

// RNG stream ids (third counter word).  Mirrored in oracle/task_ref.py.
enum {
    RS_OBS0 = 0,      // gravity xyz, ang_vel x
    RS_OBS1 = 1,      // ang_vel y z, lin_vel x y
    RS_OBS2 = 2,      // lin_vel z, height
    RS_DOFPOS = 4,    // + leg*2 + k : joints 4k..4k+3 of the leg
    RS_DOFVEL = 8,    // + leg*2 + k
    RS_KICK0 = 12,    // lin xyz, ang x
    RS_KICK1 = 13,    // ang y z
    RS_PUSH0 = 14,    // force xyz, torque x
    RS_PUSH1 = 15,    // torque y z
    RS_RESET0 = 16,   // base x, y, yaw, delay
    RS_RESET1 = 17,   // lin vel x y
    RS_RESETDOF = 20, // + leg*2 + k
    RS_CMD0 = 24,     // vx vy yaw gait_frequency
    RS_CMD1 = 25,     // still, resample time
    RS_CURR = 26,     // curriculum: grid cell, cmd x / y / yaw jitter
    RS_ACTOR = 32     // + k : action noise (bg_actor_sample)
};

}  // namespace bg
