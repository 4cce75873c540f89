// Output ("head") layers of the two MLPs fused with the loss (gfx950 only).
//
// The 128 -> 12 (actor) and 128 -> 1 (critic) output layers of reference utils/model.py:9-26 are skinny: as library GEMMs + elementwise
// kernels they cost 7 launches per network and mini-epoch, each re-reading the [B][128] hidden activations (50 MB at B = 98,304), for work
// that is HBM-bound.  Here one launch per network reads the hidden activations once:
//   bg_actor_head           mu = h W^T + b -> PPO actor loss forward + analytic backward (runner.py:145-174, bg_ppo_math.h) ->
//                           dL/dz of the last hidden layer (g W * elu'(h)), weight / bias gradients of the output layer, bias gradient of the
//                           hidden layer, log-std gradient and the loss statistics
//   bg_critic_head_forward  values = h w + b for every row (the GAE scan between forward and backward needs all of them first)
//   bg_critic_head_backward value loss (runner.py:148) backward through the output layer into the last hidden layer
// Arithmetic is plain fp32 FMA on the vector ALU: 4.6 kflop per row against 1 KB of traffic is under the HBM ridge, so nothing here is
// reshaped into an MFMA GEMM.  Reductions over rows are deterministic: every workgroup writes its partial sums, a second kernel adds them
// in a fixed order (the loss statistics and the log-std gradient keep the float64 atomics of bg_ppo_loss).
#include <hip/hip_runtime.h>

#include "../../include/booster_gym_amd.h"
#include "bg_ppo_math.h"

extern int bg_set_error(int code, const char* msg);
#define HIP_OK(expr)                                                          \
    do {                                                                      \
        hipError_t _e = (expr);                                               \
        if (_e != hipSuccess) return bg_set_error(-2, hipGetErrorString(_e)); \
    } while (0)

namespace {

constexpr int HK = 128;    // hidden width (both networks end in a 128-wide ELU layer)
constexpr int HT = 64;     // rows per tile
constexpr int HLD = 132;   // LDS row stride of the activation tile: 16-byte aligned, rows 4 banks apart
constexpr int HA = BG_NUM_DOFS;
constexpr int HEAD_MAX_GRID = 768;

__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float elu_grad_from_output(float a) { return a > 0.f ? 1.0f : a + 1.0f; }

// coalesced load of a [HT][HK] activation tile into LDS (rows past B read as zero), in two halves so that a persistent workgroup can have the
// NEXT tile's 8 float4 per thread in flight while it computes on the current one (tile_fetch before the compute, tile_put after it)
constexpr int HTV = HT * HK / 4 / 256;
__device__ __forceinline__ void tile_fetch(const float* __restrict__ h, int row0, int B, float4 (&v)[HTV]) {
#pragma unroll
    for (int i = 0; i < HTV; i++) {
        const int idx = threadIdx.x + 256 * i, r = idx >> 5, c4 = idx & 31;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < B) v[i] = *reinterpret_cast<const float4*>(h + (size_t)(row0 + r) * HK + c4 * 4);
    }
}
__device__ __forceinline__ void tile_put(const float4 (&v)[HTV], float* s_h) {
#pragma unroll
    for (int i = 0; i < HTV; i++) {
        const int idx = threadIdx.x + 256 * i, r = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<float4*>(s_h + r * HLD + c4 * 4) = v[i];
    }
}
__device__ __forceinline__ void load_tile(const float* __restrict__ h, int row0, int B, float* s_h) {
    float4 v[HTV];
    tile_fetch(h, row0, B, v);
    tile_put(v, s_h);
}

// Per-workgroup partial sums, added in a fixed order by head_finish_kernel.  scratch = [HEAD_MAX_GRID records of head_record<NO>() floats:
// [NO][HK] output-layer weight gradient, [HK] hidden bias gradient, [NO] output bias gradient] followed by the float64 loss statistics,
// statistic-major [HEAD_NSTAT][groups].  (Float64 atomics from every workgroup on the 17 shared addresses cost 45-60 us per launch.)
constexpr int HEAD_NSTAT = HA + 5;
template <int NO> constexpr int head_record() { return NO * HK + HK + 16; }
template <int NO> constexpr size_t head_stat_base() { return (size_t)HEAD_MAX_GRID * head_record<NO>(); }  // even: float64 aligned

__device__ __forceinline__ float quad_sum(float v) {  // sum over the 4 lanes of a quad (DPP quad_perm)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));  // [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));  // [2,3,0,1]
    return v;
}
// sum over the 16 lanes of a wave that share (lane & 3); valid in lanes 0..3
__device__ __forceinline__ double quadcol_sum(double v) {
    for (int o = 32; o >= 4; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float quadcol_sum(float v) {
    for (int o = 32; o >= 4; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// MODE 0: forward only (mu_out), MODE 1: forward + loss + backward.
// Thread mappings: forward + loss: thread = (row t >> 2 of the tile, actions 3 (t & 3) .. + 2), the four threads of a row exchange their
// partial log-prob / KL / bound sums with two DPP moves; backward: thread = (hidden column t & 127, row half t >> 7) with its column of W and
// its accumulators in registers for the whole launch.
template <int MODE>
__global__ __launch_bounds__(256) void actor_head_kernel(int B, int tiles, const float* __restrict__ h, const float* __restrict__ W,
                                                         const float* __restrict__ bias, const float* __restrict__ logstd,
                                                         const float* __restrict__ actions, const float* __restrict__ old_mu,
                                                         const float* __restrict__ old_logstd, const float* __restrict__ old_logp,
                                                         const float* __restrict__ adv, const double* __restrict__ adv_stats, float e_clip,
                                                         float bound_coef, float* __restrict__ mu_out, float* __restrict__ g_hidden,
                                                         float* __restrict__ partial) {
    constexpr int A = HA;
    __shared__ __attribute__((aligned(16))) float s_h[HT * HLD];
    // rows of W 132 floats apart: the four (t & 3) groups of a wave read rows 3 apart at the same k, which with a 128-float stride were the same
    // banks (a 4-way conflict on three of the four LDS reads of the forward loop)
    __shared__ __attribute__((aligned(16))) float s_w[A * HLD];
    __shared__ __attribute__((aligned(16))) float s_g[HT * A];  // dL/dmu of the tile's rows
    const int t = threadIdx.x;
    for (int i = t; i < A * HK; i += 256) s_w[(i >> 7) * HLD + (i & (HK - 1))] = W[i];
    const int fr = t >> 2, fq = t & 3;
    float fb[3];
    for (int i = 0; i < 3; i++) fb[i] = bias[3 * fq + i];
    const int kc = t & (HK - 1), half = t >> 7;
    float wcol[A], dW[A], cs = 0.f;
    for (int j = 0; j < A; j++) { wcol[j] = MODE == 1 ? W[j * HK + kc] : 0.f; dW[j] = 0.f; }
    // loss constants of this thread's three actions (bg_ppo_math.h states the same formulas for a whole row)
    float ls[3], ols[3], isig2[3], osig2[3], dbias[3] = {0.f, 0.f, 0.f};
    float ent = 0.f, mean = 0.f, inv_std = 0.f, invB = 0.f, bscale = 0.f;
    double acc_ls[3] = {0.0, 0.0, 0.0}, acc_st[4] = {0.0, 0.0, 0.0, 0.0};
    if (MODE == 1) {
        bg::ActorLossConsts<A> c;
        bg::actor_loss_consts<A>(c, B, logstd, old_logstd, adv_stats, e_clip, bound_coef);
        for (int i = 0; i < 3; i++) { ls[i] = logstd[3 * fq + i]; ols[i] = old_logstd[3 * fq + i]; }
        for (int i = 0; i < 3; i++) { const float sg = expf(ls[i]), os = expf(ols[i]); isig2[i] = 1.0f / (sg * sg); osig2[i] = os * os; }
        ent = c.ent; mean = c.mean; inv_std = c.inv_std; invB = c.invB; bscale = c.bscale;
    }
    __syncthreads();
    // (no register prefetch of the next tile here, unlike critic_head_backward_kernel: the 32 extra registers cost this kernel a resident
    // workgroup per CU, 55.9 -> 62.9 us measured)
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int row0 = tile * HT;
        load_tile(h, row0, B, s_h);
        __syncthreads();
        {   // mu = h W^T + b
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            const float4* hr = reinterpret_cast<const float4*>(s_h + fr * HLD);
            const float4* w0 = reinterpret_cast<const float4*>(s_w + (3 * fq) * HLD);
            const float4* w1 = reinterpret_cast<const float4*>(s_w + (3 * fq + 1) * HLD);
            const float4* w2 = reinterpret_cast<const float4*>(s_w + (3 * fq + 2) * HLD);
#pragma unroll 2
            for (int k4 = 0; k4 < HK / 4; k4++) {
                const float4 x = hr[k4], u = w0[k4], v = w1[k4], w = w2[k4];
                a0 = fmaf(x.x, u.x, a0); a0 = fmaf(x.y, u.y, a0); a0 = fmaf(x.z, u.z, a0); a0 = fmaf(x.w, u.w, a0);
                a1 = fmaf(x.x, v.x, a1); a1 = fmaf(x.y, v.y, a1); a1 = fmaf(x.z, v.z, a1); a1 = fmaf(x.w, v.w, a1);
                a2 = fmaf(x.x, w.x, a2); a2 = fmaf(x.y, w.y, a2); a2 = fmaf(x.z, w.z, a2); a2 = fmaf(x.w, w.w, a2);
            }
            const float m[3] = {a0 + fb[0], a1 + fb[1], a2 + fb[2]};
            const int b = row0 + fr;
            const bool live = b < B;
            const size_t o = (size_t)(live ? b : B - 1) * A + 3 * fq;
            if (mu_out && live) for (int i = 0; i < 3; i++) mu_out[o + i] = m[i];
            if (MODE == 1) {
                float d[3], hl[3], lp = 0.f, kl = 0.f, bound = 0.f;
                for (int i = 0; i < 3; i++) {
                    d[i] = actions[o + i] - m[i];
                    lp += -0.5f * d[i] * d[i] * isig2[i] - ls[i] - bg::kHalfLog2Pi;
                    const float dm = m[i] - old_mu[o + i];
                    kl += ls[i] - ols[i] + 0.5f * (osig2[i] + dm * dm) * isig2[i] - 0.5f;
                    const float hi = fmaxf(m[i] - 1.0f, 0.f), lo = fminf(m[i] + 1.0f, 0.f);
                    bound += hi * hi + lo * lo;
                    hl[i] = hi + lo;
                }
                lp = quad_sum(lp); kl = quad_sum(kl); bound = quad_sum(bound);
                const float An = (adv[live ? b : B - 1] - mean) * inv_std;
                const float ratio = expf(lp - old_logp[live ? b : B - 1]);
                const float rc = fminf(fmaxf(ratio, 1.0f - e_clip), 1.0f + e_clip);
                const float s1 = -An * ratio, s2 = -An * rc;
                // d max(s1,s2)/d logp: through s1 when it wins or ties, through the clamp only inside the clip range
                const bool inside = ratio >= 1.0f - e_clip && ratio <= 1.0f + e_clip;
                const float dlogp = (live && (inside || s1 > s2)) ? -An * ratio * invB : 0.f;
                for (int i = 0; i < 3; i++) {
                    const float gm = live ? dlogp * d[i] * isig2[i] + bscale * hl[i] : 0.f;
                    s_g[fr * A + 3 * fq + i] = gm;
                    dbias[i] += gm;
                    acc_ls[i] += (double)(dlogp * (d[i] * d[i] * isig2[i] - 1.0f));
                }
                if (live && fq == 0) {
                    acc_st[0] += (double)fmaxf(s1, s2); acc_st[1] += (double)bound; acc_st[2] += (double)ent; acc_st[3] += (double)kl;
                }
            }
        }
        __syncthreads();
        if (MODE == 1) {
            // g_hidden = (dL/dmu W) * elu'(h), dW += dL/dmu^T h, hidden bias gradient = column sums of g_hidden
            for (int rr = 0; rr < HT / 2; rr++) {
                const int r = half * (HT / 2) + rr;
                const float hv = s_h[r * HLD + kc];
                const float4* gp = reinterpret_cast<const float4*>(s_g + r * A);
                const float4 g0 = gp[0], g1 = gp[1], g2 = gp[2];
                const float g[A] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w};
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int j = 0; j < A; j += 2) {
                    sa = fmaf(g[j], wcol[j], sa); sb = fmaf(g[j + 1], wcol[j + 1], sb);
                    dW[j] = fmaf(g[j], hv, dW[j]); dW[j + 1] = fmaf(g[j + 1], hv, dW[j + 1]);
                }
                const float gz = (sa + sb) * elu_grad_from_output(hv);
                cs += gz;
                if (row0 + r < B) g_hidden[(size_t)(row0 + r) * HK + kc] = gz;
            }
            __syncthreads();
        }
    }
    if (MODE == 0) return;
    // ---- this workgroup's partial sums: the two row halves of the column-mapped sums meet in LDS
    float* red = s_h;  // [2][A + 1][HK]
    for (int j = 0; j < A; j++) red[(half * (A + 1) + j) * HK + kc] = dW[j];
    red[(half * (A + 1) + A) * HK + kc] = cs;
    __syncthreads();
    float* rec = partial + (size_t)blockIdx.x * head_record<A>();
    for (int i = t; i < (A + 1) * HK; i += 256) rec[i] = red[i] + red[(A + 1) * HK + i];
    // per-action sums: lanes sharing (lane & 3) within a wave, then the four waves through LDS.  slot = 3 fq + i; slots 12..15 = loss terms.
    const int wave = t >> 6, lane = t & 63;
    __syncthreads();
    float* redf = s_h;          // [4 waves][16]
    double* redd = reinterpret_cast<double*>(s_h + 64);  // [4 waves][16]
    for (int i = 0; i < 3; i++) {
        const float v = quadcol_sum(dbias[i]);
        const double w = quadcol_sum(acc_ls[i]);
        if (lane < 4) { redf[wave * 16 + 3 * lane + i] = v; redd[wave * 16 + 3 * lane + i] = w; }
    }
    for (int i = 0; i < 4; i++) {
        const double w = quadcol_sum(acc_st[i]);  // only fq == 0 lanes carry values
        if (lane == 0) redd[wave * 16 + 12 + i] = w;
    }
    __syncthreads();
    if (t < 16) {
        double* srec = reinterpret_cast<double*>(partial + head_stat_base<A>());
        const double w = redd[t] + redd[16 + t] + redd[32 + t] + redd[48 + t];
        // statistic index: [0, A) dL/dlogstd, A unused (the value error is the critic head's), A+1.. = surrogate, bound, entropy, kl
        const int k = t < A ? t : t + 1;
        srec[(size_t)k * gridDim.x + blockIdx.x] = w;
        if (t < A) rec[(A + 1) * HK + t] = redf[t] + redf[16 + t] + redf[32 + t] + redf[48 + t];
    }
}

__global__ __launch_bounds__(256) void critic_head_backward_kernel(int B, int tiles, const float* __restrict__ h, const float* __restrict__ w,
                                                                   const float* __restrict__ values, const float* __restrict__ returns,
                                                                   float* __restrict__ g_hidden, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float s_h[HT * HLD];
    __shared__ float s_g[HT];
    const int t = threadIdx.x, kc = t & (HK - 1), half = t >> 7;
    const float wk = w[kc], invB = 1.0f / (float)B;
    float dW = 0.f, cs = 0.f, db = 0.f;
    double verr2 = 0.0;
    float4 nxt[HTV];
    tile_fetch(h, blockIdx.x * HT, B, nxt);
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int row0 = tile * HT;
        tile_put(nxt, s_h);
        if (tile + (int)gridDim.x < tiles) tile_fetch(h, (tile + gridDim.x) * HT, B, nxt);  // in flight under this tile's compute
        if (t < HT) {
            const int b = row0 + t;
            float g = 0.f;
            if (b < B) {
                const float verr = values[b] - returns[b];
                g = 2.0f * verr * invB;  // d mean((v - ret)^2) / dv, runner.py:148
                verr2 += (double)(verr * verr);
                db += g;
            }
            s_g[t] = g;
        }
        __syncthreads();
        for (int rr = 0; rr < HT / 2; rr++) {
            const int r = half * (HT / 2) + rr;
            const float hv = s_h[r * HLD + kc], g = s_g[r];
            dW = fmaf(g, hv, dW);
            const float gz = g * wk * elu_grad_from_output(hv);
            cs += gz;
            if (row0 + r < B) g_hidden[(size_t)(row0 + r) * HK + kc] = gz;
        }
        __syncthreads();
    }
    float* red = s_h;  // [2][2][HK]
    red[(half * 2) * HK + kc] = dW;
    red[(half * 2 + 1) * HK + kc] = cs;
    __syncthreads();
    float* rec = partial + (size_t)blockIdx.x * head_record<1>();
    for (int i = t; i < 2 * HK; i += 256) rec[i] = red[i] + red[2 * HK + i];
    if (t < 64) {
        float v = db;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        const double s = wave_sum_d(verr2);
        if (t == 0) { rec[2 * HK] = v; reinterpret_cast<double*>(partial + head_stat_base<1>())[blockIdx.x] = s; }
    }
}

// One float64 statistic (row k of the stat-major [n_stat][groups] block) added up by one workgroup in a fixed order, then ONE atomic: k < n_ls goes to
// grad_logstd[k] (+ entropy_coef: d(entropy.mean())/dlogstd = 1), the rest to stats[k - n_ls]; a statistic whose stat_skip bit is set is skipped.
// (One workgroup PER statistic: a single workgroup walking the 17 statistics of the actor head one after the other -- a load, a butterfly and two
// barriers each -- was a 66 us launch on the actor's chain of every mini-epoch.)
__device__ __forceinline__ void finish_statistic(const double* __restrict__ sp, int groups, int k, int n_ls, unsigned stat_skip, double entropy_coef,
                                                 double* __restrict__ grad_logstd, double* __restrict__ stats) {
    if ((stat_skip >> k) & 1u) return;
    __shared__ double sd[4];
    double s = 0.0;
#pragma unroll 4
    for (int g = threadIdx.x; g < groups; g += 256) s += sp[(size_t)k * groups + g];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double v = sd[0] + sd[1] + sd[2] + sd[3];
        if (k < n_ls) atomicAdd(&grad_logstd[k], v + entropy_coef);
        else atomicAdd(&stats[k - n_ls], v);
    }
}

// fixed-order sum of the workgroups' records: out[i] = sum_g partial[g][i]; 16 outputs x 16 record slices per workgroup.  The n_stat workgroups
// behind those add up the float64 statistics (stat-major [n_stat][groups] at stat_base), one each (finish_statistic).
__global__ __launch_bounds__(256) void head_finish_kernel(int groups, int record, int n_out, const float* __restrict__ partial, float* __restrict__ grad_w,
                                                          int n_w, float* __restrict__ grad_b_hidden, float* __restrict__ grad_b, size_t stat_base,
                                                          int n_stat, int n_ls, unsigned stat_skip, double entropy_coef,
                                                          double* __restrict__ grad_logstd, double* __restrict__ stats) {
    const int nsum = (n_out + 15) / 16;
    if ((int)blockIdx.x >= nsum) {
        finish_statistic(reinterpret_cast<const double*>(partial + stat_base), groups, blockIdx.x - nsum, n_ls, stat_skip, entropy_coef, grad_logstd, stats);
        return;
    }
    __shared__ float sm[16][17];
    const int o = threadIdx.x & 15, gs = threadIdx.x >> 4, i = blockIdx.x * 16 + o;
    float s = 0.f;
    if (i < n_out) {
#pragma unroll 8  // 8 loads in flight: the 48 dependent adds of a thread were a chain of 48 L2 round trips
        for (int g = gs; g < groups; g += 16) s += partial[(size_t)g * record + i];
    }
    sm[gs][o] = s;
    __syncthreads();
    if (threadIdx.x < 16 && i < n_out) {
        float v = 0.f;
        for (int k = 0; k < 16; k++) v += sm[k][o];
        if (i < n_w) grad_w[i] = v;
        else if (i < n_w + HK) grad_b_hidden[i - n_w] = v;
        else grad_b[i - n_w - HK] = v;
    }
}

// ---- deferred reductions (include/booster_gym_amd.h: bg_reduce_problem / bg_reduce_group): the work of head_finish_kernel (and of the backward
// layer's column-sum finish) for up to 8 descriptors in one launch.  Workgroups [begin_k, begin_k + nblk_k) serve descriptor k: 16 outputs x 16
// slices of the groups each, followed by one workgroup per float64 statistic of the descriptor (finish_statistic).
constexpr int RG_MAX = 8;
struct ReduceGroup { int np; int begin[RG_MAX]; bg_reduce_problem p[RG_MAX]; };
__global__ __launch_bounds__(256) void reduce_group_kernel(ReduceGroup grp) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < RG_MAX; j++)
        if (j < grp.np && (int)blockIdx.x >= grp.begin[j]) k = j;
    const bg_reduce_problem& pr = grp.p[k];
    const int b = blockIdx.x - grp.begin[k], nsum = (pr.n_out + 15) / 16;
    if (b >= nsum) {  // one of the statistics blocks of this descriptor
        finish_statistic(reinterpret_cast<const double*>(pr.partial + pr.stat_base), pr.groups, b - nsum, pr.n_ls, pr.stat_skip, pr.entropy_coef, pr.grad_logstd,
                         pr.stats);
        return;
    }
    __shared__ float sm[16][17];
    const int o = threadIdx.x & 15, gs = threadIdx.x >> 4, i = b * 16 + o;
    float s = 0.f;
    if (i < pr.n_out) {
#pragma unroll 8
        for (int g = gs; g < pr.groups; g += 16) s += pr.partial[(size_t)g * pr.record + i];
    }
    sm[gs][o] = s;
    __syncthreads();
    if (threadIdx.x < 16 && i < pr.n_out) {
        float v = 0.f;
        for (int j = 0; j < 16; j++) v += sm[j][o];
        if (i < pr.n[0]) pr.out[0][i] = v;
        else if (i < pr.n[0] + pr.n[1]) pr.out[1][i - pr.n[0]] = v;
        else pr.out[2][i - pr.n[0] - pr.n[1]] = v;
    }
}
extern "C" int bg_reduce_group(const bg_reduce_problem* problems, int32_t count, void* stream) {
    if (!problems || count <= 0 || count > RG_MAX) return bg_set_error(-1, "bg_reduce_group: 1 to 8 descriptors");
    ReduceGroup grp;
    grp.np = count;
    int blocks = 0;
    for (int k = 0; k < count; k++) {
        const bg_reduce_problem& q = problems[k];
        if (!q.partial || q.groups <= 0 || q.record <= 0 || q.n_out <= 0 || q.n_out > q.record || !q.out[0] || q.n[0] <= 0 ||
            q.n[0] + q.n[1] + q.n[2] != q.n_out || (q.n[1] > 0 && !q.out[1]) || (q.n[2] > 0 && !q.out[2]))
            return bg_set_error(-1, "bg_reduce_group: bad descriptor");
        if (q.n_stat < 0 || q.n_stat > 32 || (q.n_stat > 0 && (!q.stats || (q.n_ls > 0 && !q.grad_logstd) || (q.stat_base & 1))))
            return bg_set_error(-1, "bg_reduce_group: bad statistics descriptor");
        grp.begin[k] = blocks;
        grp.p[k] = q;
        blocks += (q.n_out + 15) / 16 + q.n_stat;
    }
    hipLaunchKernelGGL(reduce_group_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    HIP_OK(hipGetLastError());
    return 0;
}

// values = h w + b: half a wave per row (32 lanes x 16 bytes = one 512-byte row), 4 rows in flight per half-wave
__global__ __launch_bounds__(256) void critic_head_forward_kernel(int rows, const float* __restrict__ h, const float* __restrict__ w,
                                                                  const float* __restrict__ b, float* __restrict__ values) {
    const int lane = threadIdx.x & 31;
    const int hw = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, nhw = (gridDim.x * blockDim.x) >> 5;
    const float4 wv = *reinterpret_cast<const float4*>(w + lane * 4);
    const float bias = b[0];
    for (int r0 = hw * 4; r0 < rows; r0 += nhw * 4) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = r0 + u < rows ? r0 + u : rows - 1;
            x[u] = *reinterpret_cast<const float4*>(h + (size_t)r * HK + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float s = x[u].x * wv.x;
            s = fmaf(x[u].y, wv.y, s); s = fmaf(x[u].z, wv.z, s); s = fmaf(x[u].w, wv.w, s);
            for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0 && r0 + u < rows) values[r0 + u] = s + bias;
        }
    }
}

// values = h w + b for all T + 1 time rows of 16 envs, then the GAE scan of those envs, in ONE launch (what bg_critic_head_forward, a fill and bg_gae do
// as three dependent launches in front of the actor's loss: ~45 us of small kernels and gaps per mini-epoch).  One workgroup = 16 envs: the value
// phase is critic_head_forward_kernel's (half a wave per row, same sums: bit-identical values), the values go to LDS and to values_all, threads
// 0..15 then run gae_kernel's scan on registers (its loads were issued at the top of the kernel).  Moments of the advantages: one float64 triple per
// workgroup in `partial`; the last workgroup to finish (ticket) adds them up in a fixed order and WRITES sums (no zero fill, no float atomics).
// Envs per workgroup: 16 where the launch also evaluates the output layer (25 x 16 rows of activations per workgroup); 64 for the scan alone, which is what
// the training loop runs (values from the chained forward kernel's value head): a quarter of the workgroups, tickets and fences beside the actor's
// forward chain -- update 21.12 against 21.26 ms on one box, three of three alternating pairs (tools/ab_env.sh)
#ifndef BG_GAE_SCAN_ENVS
#define BG_GAE_SCAN_ENVS 64
#endif
constexpr int VG_ENVS_VALUES = 16, VG_ENVS_SCAN = BG_GAE_SCAN_ENVS;
template <int TMAX, int VG_ENVS>
__global__ __launch_bounds__(256) void critic_values_gae_kernel(int T, int N, const float* __restrict__ h, const float* __restrict__ w,
                                                                const float* __restrict__ b, float* __restrict__ rewards,
                                                                const uint8_t* __restrict__ dones, const uint8_t* __restrict__ touts, float gamma,
                                                                float lam, float* __restrict__ values_all, float* __restrict__ adv,
                                                                float* __restrict__ ret, double* __restrict__ partial, unsigned* __restrict__ ticket,
                                                                double* __restrict__ sums) {
    __shared__ float sv[(TMAX + 1) * VG_ENVS];
    __shared__ double sd[3 * 4];
    __shared__ unsigned s_last;
    const int e0 = blockIdx.x * VG_ENVS, ne = N - e0 < VG_ENVS ? N - e0 : VG_ENVS;
    // the scan's operands (threads 0..15: one env each), in flight during the value phase
    float r[TMAX];
    uint8_t dn[TMAX], to[TMAX];
    const bool scan = (int)threadIdx.x < ne;
    if (scan) {
#pragma unroll
        for (int t = 0; t < TMAX; t++)
            if (t < T) {
                const size_t k = (size_t)t * N + e0 + threadIdx.x;
                r[t] = rewards[k]; dn[t] = dones[k]; to[t] = touts[k];
            }
    }
    const int lane = threadIdx.x & 31, hw = threadIdx.x >> 5, R = (T + 1) * VG_ENVS;
    if (!h) {  // the values are an input (the chained forward kernel's value head wrote them)
        for (int q = threadIdx.x; q < R; q += 256) {
            const int t = q / VG_ENVS, e = q % VG_ENVS;
            sv[q] = values_all[(size_t)t * N + e0 + (e < ne ? e : ne - 1)];
        }
    }
    const float4 wv = h ? *reinterpret_cast<const float4*>(w + lane * 4) : float4{0.f, 0.f, 0.f, 0.f};
    const float bias = h ? b[0] : 0.f;
    for (int r0 = hw * 4; h && r0 < R; r0 += 32) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = r0 + u < R ? r0 + u : R - 1, t = q / VG_ENVS, e = q % VG_ENVS;
            x[u] = *reinterpret_cast<const float4*>(h + ((size_t)t * N + e0 + (e < ne ? e : ne - 1)) * HK + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float s = x[u].x * wv.x;
            s = fmaf(x[u].y, wv.y, s); s = fmaf(x[u].z, wv.z, s); s = fmaf(x[u].w, wv.w, s);
            for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const int q = r0 + u, t = q / VG_ENVS, e = q % VG_ENVS;
            if (lane == 0 && q < R) {
                sv[q] = s + bias;
                if (e < ne) values_all[(size_t)t * N + e0 + e] = s + bias;
            }
        }
    }
    __syncthreads();
    double acc[3] = {0.0, 0.0, 0.0};
    if (scan) {
        float next_v = sv[T * VG_ENVS + threadIdx.x], last_adv = 0.f;
#pragma unroll
        for (int t = TMAX - 1; t >= 0; t--)
            if (t < T) {
                const size_t k = (size_t)t * N + e0 + threadIdx.x;
                const float v = sv[t * VG_ENVS + threadIdx.x];
                float rr = r[t];
                if (to[t]) { rr = v; rewards[k] = v; }  // runner.py:135 (in place, repeated every mini-epoch with the current critic)
                const float nn = (dn[t] != 0 || to[t] != 0) ? 0.f : 1.f;
                const float delta = rr + gamma * nn * next_v - v;
                last_adv = delta + gamma * lam * nn * last_adv;
                adv[k] = last_adv;
                ret[k] = v + last_adv;
                acc[0] += (double)last_adv; acc[1] += (double)last_adv * (double)last_adv; acc[2] += 1.0;
                next_v = v;
            }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {  // (the scanning threads are the first VG_ENVS of the workgroup; the others carry zeros)
        const double s = wave_sum_d(acc[k]);
        if ((threadIdx.x & 63) == 0) sd[k * 4 + (threadIdx.x >> 6)] = s;
    }
    __syncthreads();
    // The triple is PUBLISHED, not fenced: an agent-scope store goes through the XCD's L2 to memory, and once the wave's vmcnt is back at zero it
    // has arrived; the ticket is taken after that.  (A release fence -- __threadfence -- writes back the whole L2 of the XCD instead, every dirty
    // line of the forward chain that runs beside this launch included, once per workgroup: this launch sits between the critic's forward pass and
    // the actor's loss.)  The reader takes agent-scope loads, which do not hit a stale line of its own L2.
    // That argument rests on two properties of gfx942 / gfx950 that the HIP memory model does not promise (the return of an sc1 store into vmcnt =
    // visible at agent scope; one vmcnt counter for loads and stores), so it is compiled for those targets only: anything else gets the portable
    // release (fence + relaxed ticket), slower but correct by the model.  DESIGN.md section 5 records the assumption.
#if defined(__gfx942__) || defined(__gfx950__)
    if (threadIdx.x < 3) {
        __hip_atomic_store(&partial[(size_t)blockIdx.x * 3 + threadIdx.x], sd[threadIdx.x * 4] + sd[threadIdx.x * 4 + 1] + sd[threadIdx.x * 4 + 2] + sd[threadIdx.x * 4 + 3],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // threads 0..2 and the ticket's thread 0 are one wave
    }
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
#else
    if (threadIdx.x < 3)
        __hip_atomic_store(&partial[(size_t)blockIdx.x * 3 + threadIdx.x], sd[threadIdx.x * 4] + sd[threadIdx.x * 4 + 1] + sd[threadIdx.x * 4 + 2] + sd[threadIdx.x * 4 + 3],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
#endif
    __syncthreads();
    if (s_last) {  // every workgroup's triple has arrived: fixed-order total
        for (int k = 0; k < 3; k++) {
            double s = 0.0;
            for (int g = threadIdx.x; g < (int)gridDim.x; g += 256) s += __hip_atomic_load(&partial[(size_t)g * 3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s = wave_sum_d(s);
            if ((threadIdx.x & 63) == 0) sd[k * 4 + (threadIdx.x >> 6)] = s;
        }
        __syncthreads();
        if (threadIdx.x < 3) sums[threadIdx.x] = sd[threadIdx.x * 4] + sd[threadIdx.x * 4 + 1] + sd[threadIdx.x * 4 + 2] + sd[threadIdx.x * 4 + 3];
        if (threadIdx.x == 0) *ticket = 0u;
    }
}

int head_grid(int tiles) { return tiles < HEAD_MAX_GRID ? tiles : HEAD_MAX_GRID; }
bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int bg_critic_head_forward(int32_t rows, const float* h, const float* w, const float* b, float* values, void* stream) {
    if (rows <= 0 || !h || !w || !b || !values) return bg_set_error(-1, "bg_critic_head_forward: bad argument");
    if (!aligned16(h) || !aligned16(w)) return bg_set_error(-1, "bg_critic_head_forward: h and w must be 16-byte aligned");
    int blocks = (rows + 31) / 32;  // 8 half-waves x 4 rows per 256-thread workgroup and pass
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(critic_head_forward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, h, w, b, values);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_critic_values_gae(int32_t T, int32_t N, const float* h, const float* w, const float* b, float* rewards, const uint8_t* dones,
                                    const uint8_t* time_outs, float gamma, float lam, float* values_all, float* advantages, float* returns, double* sums,
                                    double* scratch, void* stream) {
    if (T <= 0 || N <= 0 || (h && (!w || !b)) || !rewards || !dones || !time_outs || !values_all || !advantages || !returns || !sums || !scratch)
        return bg_set_error(-1, "bg_critic_values_gae: bad argument");
    if (h && (!aligned16(h) || !aligned16(w))) return bg_set_error(-1, "bg_critic_values_gae: h and w must be 16-byte aligned");
    if (T > 32) return bg_set_error(-4, "bg_critic_values_gae: horizon above 32 (use bg_critic_head_forward + bg_gae)");
    unsigned* ticket = reinterpret_cast<unsigned*>(scratch + (size_t)((N + 15) / 16) * 3);  // the scratch's last element, whichever form runs
    if (h) hipLaunchKernelGGL((critic_values_gae_kernel<32, VG_ENVS_VALUES>), dim3((N + VG_ENVS_VALUES - 1) / VG_ENVS_VALUES), dim3(256), 0, (hipStream_t)stream, T, N, h, w,
                              b, rewards, dones, time_outs, gamma, lam, values_all, advantages, returns, scratch, ticket, sums);
    else hipLaunchKernelGGL((critic_values_gae_kernel<32, VG_ENVS_SCAN>), dim3((N + VG_ENVS_SCAN - 1) / VG_ENVS_SCAN), dim3(256), 0, (hipStream_t)stream, T, N, h, w, b,
                            rewards, dones, time_outs, gamma, lam, values_all, advantages, returns, scratch, ticket, sums);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_actor_head(int32_t B, int32_t mode, const float* h, const float* W, const float* bias, const float* logstd, const float* actions,
                             const float* old_mu, const float* old_logstd, const float* old_logp, const float* adv, const double* adv_stats,
                             float e_clip, float bound_coef, float entropy_coef, float* mu_out, float* g_hidden, float* grad_W, float* grad_b,
                             float* grad_b_hidden, double* grad_logstd, double* stats, float* scratch, void* stream) {
    if (B <= 0 || !h || !W || !bias) return bg_set_error(-1, "bg_actor_head: bad argument");
    if (!aligned16(h)) return bg_set_error(-1, "bg_actor_head: h must be 16-byte aligned");
    const int tiles = (B + HT - 1) / HT, grid = head_grid(tiles);
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) {
        if (!mu_out) return bg_set_error(-1, "bg_actor_head: mode 0 needs mu_out");
        hipLaunchKernelGGL(actor_head_kernel<0>, dim3(grid), dim3(256), 0, st, B, tiles, h, W, bias, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, 0.f, 0.f, mu_out, nullptr, nullptr);
        HIP_OK(hipGetLastError());
        return 0;
    }
    if (mode != 1) return bg_set_error(-1, "bg_actor_head: mode must be 0 (forward) or 1 (forward + loss + backward)");
    if (!logstd || !actions || !old_mu || !old_logstd || !old_logp || !adv || !adv_stats || !g_hidden || !grad_W || !grad_b || !grad_b_hidden ||
        !grad_logstd || !stats || !scratch)
        return bg_set_error(-1, "bg_actor_head: bad argument");
    hipLaunchKernelGGL(actor_head_kernel<1>, dim3(grid), dim3(256), 0, st, B, tiles, h, W, bias, logstd, actions, old_mu, old_logstd, old_logp, adv,
                       adv_stats, e_clip, bound_coef, mu_out, g_hidden, scratch);
    constexpr int n_out = HA * HK + HK + HA;
    hipLaunchKernelGGL(head_finish_kernel, dim3((n_out + 15) / 16 + HEAD_NSTAT), dim3(256), 0, st, grid, head_record<HA>(), n_out, scratch, grad_W, HA * HK,
                       grad_b_hidden, grad_b, head_stat_base<HA>(), HEAD_NSTAT, HA, 1u << HA, (double)entropy_coef, grad_logstd, stats);
    HIP_OK(hipGetLastError());
    return 0;
}

static void head_finish_desc(bg_reduce_problem* f, const float* scratch, int grid, int record, float* grad_w, int n_w, float* grad_b_hidden, float* grad_b, int n_b,
                             size_t stat_base, int n_stat, int n_ls, unsigned skip, double entropy_coef, double* grad_logstd, double* stats) {
    f->partial = scratch; f->groups = grid; f->record = record; f->n_out = n_w + HK + n_b;
    f->out[0] = grad_w; f->n[0] = n_w; f->out[1] = grad_b_hidden; f->n[1] = HK; f->out[2] = grad_b; f->n[2] = n_b;
    f->stat_base = stat_base; f->n_stat = n_stat; f->n_ls = n_ls; f->stat_skip = skip; f->entropy_coef = entropy_coef; f->grad_logstd = grad_logstd; f->stats = stats;
}
extern "C" int bg_actor_head_partial(int32_t B, const float* h, const float* W, const float* bias, const float* logstd, const float* actions,
                                     const float* old_mu, const float* old_logstd, const float* old_logp, const float* adv, const double* adv_stats,
                                     float e_clip, float bound_coef, float entropy_coef, float* mu_out, float* g_hidden, float* grad_W, float* grad_b,
                                     float* grad_b_hidden, double* grad_logstd, double* stats, float* scratch, bg_reduce_problem* finish, void* stream) {
    if (B <= 0 || !h || !W || !bias || !logstd || !actions || !old_mu || !old_logstd || !old_logp || !adv || !adv_stats || !g_hidden || !grad_W || !grad_b ||
        !grad_b_hidden || !grad_logstd || !stats || !scratch || !finish)
        return bg_set_error(-1, "bg_actor_head_partial: bad argument");
    if (!aligned16(h)) return bg_set_error(-1, "bg_actor_head_partial: h must be 16-byte aligned");
    const int tiles = (B + HT - 1) / HT, grid = head_grid(tiles);
    hipLaunchKernelGGL(actor_head_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, tiles, h, W, bias, logstd, actions, old_mu, old_logstd, old_logp,
                       adv, adv_stats, e_clip, bound_coef, mu_out, g_hidden, scratch);
    HIP_OK(hipGetLastError());
    head_finish_desc(finish, scratch, grid, head_record<HA>(), grad_W, HA * HK, grad_b_hidden, grad_b, HA, head_stat_base<HA>(), HEAD_NSTAT, HA, 1u << HA,
                     (double)entropy_coef, grad_logstd, stats);
    return 0;
}
extern "C" int bg_critic_head_backward_partial(int32_t B, const float* h, const float* w, const float* values, const float* returns, float* g_hidden,
                                               float* grad_w, float* grad_b, float* grad_b_hidden, double* stats, float* scratch, bg_reduce_problem* finish,
                                               void* stream) {
    if (B <= 0 || !h || !w || !values || !returns || !g_hidden || !grad_w || !grad_b || !grad_b_hidden || !stats || !scratch || !finish)
        return bg_set_error(-1, "bg_critic_head_backward_partial: bad argument");
    if (!aligned16(h)) return bg_set_error(-1, "bg_critic_head_backward_partial: h must be 16-byte aligned");
    const int tiles = (B + HT - 1) / HT, grid = head_grid(tiles);
    hipLaunchKernelGGL(critic_head_backward_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, B, tiles, h, w, values, returns, g_hidden, scratch);
    HIP_OK(hipGetLastError());
    head_finish_desc(finish, scratch, grid, head_record<1>(), grad_w, HK, grad_b_hidden, grad_b, 1, head_stat_base<1>(), 1, 0, 0u, 0.0, nullptr, stats);
    return 0;
}

extern "C" int bg_critic_head_backward(int32_t B, const float* h, const float* w, const float* values, const float* returns, float* g_hidden,
                                       float* grad_w, float* grad_b, float* grad_b_hidden, double* stats, float* scratch, void* stream) {
    if (B <= 0 || !h || !w || !values || !returns || !g_hidden || !grad_w || !grad_b || !grad_b_hidden || !stats || !scratch)
        return bg_set_error(-1, "bg_critic_head_backward: bad argument");
    if (!aligned16(h)) return bg_set_error(-1, "bg_critic_head_backward: h must be 16-byte aligned");
    const int tiles = (B + HT - 1) / HT, grid = head_grid(tiles);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(critic_head_backward_kernel, dim3(grid), dim3(256), 0, st, B, tiles, h, w, values, returns, g_hidden, scratch);
    constexpr int n_out = HK + HK + 1;
    hipLaunchKernelGGL(head_finish_kernel, dim3((n_out + 15) / 16 + 1), dim3(256), 0, st, grid, head_record<1>(), n_out, scratch, grad_w, HK, grad_b_hidden,
                       grad_b, head_stat_base<1>(), 1, 0, 0u, 0.0, (double*)nullptr, stats);
    HIP_OK(hipGetLastError());
    return 0;
}
