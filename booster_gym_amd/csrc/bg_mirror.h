// The copies of a weight matrix that the optimiser launch keeps current while it updates the parameters (bg_param_mirror, include/booster_gym_amd.h):
// element (r, c) of the [rows][cols] matrix, new value pn.  Shared by bg_optimizer_step (bg_ppo.hip) and bg_update_tail (bg_tail.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/booster_gym_amd.h"

constexpr int OPT_MAX_MIRRORS = 16;
struct ParamMirrors { int n; bg_param_mirror m[OPT_MAX_MIRRORS]; };

__device__ __forceinline__ void bg_mirror_write(const bg_param_mirror& mm, int r, int c, float pn) {
    if (mm.transpose < 2) {
        mm.dst[mm.transpose ? (size_t)c * mm.ld + r : (size_t)r * mm.ld + c] = pn;
    } else {
        // the three bf16 planes of bg_mlp_split_weights (hi + mid + lo == pn exactly), [n][ld / 32][3][32] with the chained kernels' k order inside
        // a 32-chunk; 2: planes of W (n = r, k = c), 3: planes of W^T (n = c, k = r)
        const int n = mm.transpose == 2 ? r : c, k = mm.transpose == 2 ? c : r;
        const unsigned u = __float_as_uint(pn);
        const float r1 = pn - __uint_as_float(u & 0xffff0000u);
        const unsigned v = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(v & 0xffff0000u);
        const int kin = k & 31, s = kin >> 3, hh = (kin >> 2) & 1, q = kin & 3;
        unsigned short* d = reinterpret_cast<unsigned short*>(mm.dst) + ((size_t)n * (mm.ld >> 5) + (k >> 5)) * 96 + (s >> 1) * 16 + hh * 8 + (s & 1) * 4 + q;
        d[0] = (unsigned short)(u >> 16);
        d[32] = (unsigned short)(v >> 16);
        d[64] = (unsigned short)(__float_as_uint(r2) >> 16);
        if (mm.pad > 0) {  // ... and the planes of -W, mm.pad uint16 behind (bg_mlp_split_weights_pm's layout)
            d += mm.pad;
            d[0] = (unsigned short)((u >> 16) ^ 0x8000u);
            d[32] = (unsigned short)((v >> 16) ^ 0x8000u);
            d[64] = (unsigned short)((__float_as_uint(r2) >> 16) ^ 0x8000u);
        }
    }
}
// host-side check of a descriptor against a flat buffer of n floats
inline bool bg_mirror_ok(const bg_param_mirror& q, int64_t n) {
    if (!q.dst || q.rows <= 0 || q.cols <= 0 || q.offset < 0 || (int64_t)q.offset + (int64_t)q.rows * q.cols > n || q.transpose < 0 || q.transpose > 3) return false;
    if (q.ld < ((q.transpose & 1) ? q.rows : q.cols)) return false;
    if (q.transpose >= 2 && ((q.ld & 31) != 0 || ((uintptr_t)q.dst & 15) != 0)) return false;
    return true;
}
