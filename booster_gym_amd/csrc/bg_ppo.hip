// HIP kernels + C ABI for the PPO half of the hot path (gfx950 only):
//   bg_gae            reference utils/utils.py:33-44 (discount_values) + utils/runner.py:135,144 fused into one backward scan
//   bg_ppo_loss       reference utils/runner.py:145-174 + utils/utils.py:47-52: forward AND analytic backward in one pass
//   bg_gaussian_logp  reference utils/runner.py:123-125
//   bg_actor_sample   reference utils/model.py:29-32 + dist.sample() (runner.py:109-111) as one launch
//   bg_adam_step      reference utils/runner.py:162-165 (clip_grad_norm_ + torch.optim.Adam.step) on a flat buffer
//   bg_adapt_lr       reference utils/runner.py:174-180 without the host sync
// All of these are HBM/latency-bound elementwise or scan work; none is reshaped into a GEMM.
#include <hip/hip_runtime.h>
#include <string>

#include "../../include/booster_gym_amd.h"
#include "bg_mirror.h"
#include "bg_ppo_math.h"
#include "bg_rng.h"

extern int bg_set_error(int code, const char* msg);
#define HIP_OK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return bg_set_error(-2, hipGetErrorString(_e));               \
    } while (0)

// ------------------------------------------------------------------ block reduction helper (wave64)
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int NV>
__device__ __forceinline__ void block_atomic_add(double (&v)[NV], double* dst, double* smem /*[NV * waves]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
    for (int k = 0; k < NV; k++) {
        double s = wave_sum(v[k]);
        if (lane == 0) smem[k * waves + wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0;
        for (int w = 0; w < waves; w++) s += smem[threadIdx.x * waves + w];
        atomicAdd(&dst[threadIdx.x], s);
    }
}

// ------------------------------------------------------------------ GAE: one lane per env, backward scan over T
// The scan is a chain of T dependent steps per env, and it sits on the critical path of every mini-epoch (the actor's loss waits for the
// advantages).  All 4 T loads of a lane are independent of the chain, so they are issued up front into registers (TMAX-way unrolled) and the
// chain then runs on registers: one memory latency per launch instead of T of them (25 -> ~5 us at T = 24, N = 4096).
template <int TMAX>
__global__ __launch_bounds__(64) void gae_kernel(int T, int N, float* __restrict__ rewards, const uint8_t* __restrict__ dones,
                                                 const uint8_t* __restrict__ touts, const float* __restrict__ values,
                                                 const float* __restrict__ last_values, float gamma, float lam, float* __restrict__ adv,
                                                 float* __restrict__ ret, double* __restrict__ sums) {
    __shared__ double sm[3 * 4];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[3] = {0.0, 0.0, 0.0};
    if (e < N) {
        float v[TMAX], r[TMAX];
        uint8_t dn[TMAX], to[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; t++)
            if (t < T) {
                const size_t k = (size_t)t * N + e;
                v[t] = values[k]; r[t] = rewards[k]; dn[t] = dones[k]; to[t] = touts[k];
            }
        float next_v = last_values[e], last_adv = 0.f;
#pragma unroll
        for (int t = TMAX - 1; t >= 0; t--)
            if (t < T) {
                const size_t k = (size_t)t * N + e;
                float rr = r[t];
                if (to[t]) { rr = v[t]; rewards[k] = v[t]; }  // runner.py:135 (in place, repeated every mini-epoch with the current critic)
                const float nn = (dn[t] != 0 || to[t] != 0) ? 0.f : 1.f;
                const float delta = rr + gamma * nn * next_v - v[t];
                last_adv = delta + gamma * lam * nn * last_adv;
                adv[k] = last_adv;
                ret[k] = v[t] + last_adv;
                acc[0] += (double)last_adv; acc[1] += (double)last_adv * (double)last_adv; acc[2] += 1.0;
                next_v = v[t];
            }
    }
    block_atomic_add<3>(acc, sums, sm);
}
// any horizon: the plain loop (loads inside the chain)
__global__ __launch_bounds__(256) void gae_kernel_loop(int T, int N, float* __restrict__ rewards, const uint8_t* __restrict__ dones,
                                                       const uint8_t* __restrict__ touts, const float* __restrict__ values,
                                                       const float* __restrict__ last_values, float gamma, float lam, float* __restrict__ adv,
                                                       float* __restrict__ ret, double* __restrict__ sums) {
    __shared__ double sm[3 * 4];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[3] = {0.0, 0.0, 0.0};
    if (e < N) {
        float next_v = last_values[e], last_adv = 0.f;
        for (int t = T - 1; t >= 0; t--) {
            const size_t k = (size_t)t * N + e;
            const float v = values[k];
            const bool to = touts[k] != 0;
            float r = rewards[k];
            if (to) { r = v; rewards[k] = v; }
            const float nn = (dones[k] != 0 || to) ? 0.f : 1.f;
            const float delta = r + gamma * nn * next_v - v;
            last_adv = delta + gamma * lam * nn * last_adv;
            adv[k] = last_adv;
            ret[k] = v + last_adv;
            acc[0] += (double)last_adv; acc[1] += (double)last_adv * (double)last_adv; acc[2] += 1.0;
            next_v = v;
        }
    }
    block_atomic_add<3>(acc, sums, sm);
}

// ------------------------------------------------------------------ PPO loss forward + backward, one lane per sample
using bg::kHalfLog2Pi;

template <int A>
__global__ __launch_bounds__(256) void ppo_loss_kernel(int B, const float* __restrict__ mu, const float* __restrict__ logstd,
                                                       const float* __restrict__ actions, const float* __restrict__ old_mu,
                                                       const float* __restrict__ old_logstd, const float* __restrict__ old_logp,
                                                       const float* __restrict__ adv, const double* __restrict__ adv_stats,
                                                       const float* __restrict__ values, const float* __restrict__ returns, float e_clip,
                                                       float bound_coef, float entropy_coef, float* __restrict__ grad_mu,
                                                       float* __restrict__ grad_values, double* __restrict__ grad_logstd,
                                                       double* __restrict__ stats) {
    __shared__ double sm[(A + 5) * 4];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    bg::ActorLossConsts<A> c;
    bg::actor_loss_consts<A>(c, B, logstd, old_logstd, adv_stats, e_clip, bound_coef);
    double acc[A + 5];
    for (int k = 0; k < A + 5; k++) acc[k] = 0.0;
    if (b < B) {
        float m[A], act[A], om[A], gmu[A];
        for (int a = 0; a < A; a++) { m[a] = mu[(size_t)b * A + a]; act[a] = actions[(size_t)b * A + a]; om[a] = old_mu[(size_t)b * A + a]; }
        bg::actor_loss_row<A>(c, m, act, om, old_logp[b], adv[b], gmu, acc);
        for (int a = 0; a < A; a++) grad_mu[(size_t)b * A + a] = gmu[a];
        const float verr = values[b] - returns[b];
        grad_values[b] = 2.0f * verr * c.invB;
        acc[A + 0] = (double)(verr * verr);
    }
    // reduce: grad_logstd[A] and stats[5]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
    for (int k = 0; k < A + 5; k++) {
        double s = wave_sum(acc[k]);
        if (lane == 0) sm[k * waves + wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < A + 5) {
        double s = 0;
        for (int w = 0; w < waves; w++) s += sm[threadIdx.x * waves + w];
        if (threadIdx.x < A) atomicAdd(&grad_logstd[threadIdx.x], s);
        else atomicAdd(&stats[threadIdx.x - A], s);
    }
    if (blockIdx.x == 0 && threadIdx.x < A) atomicAdd(&grad_logstd[threadIdx.x], (double)entropy_coef);  // d(entropy.mean())/dlogstd = 1
}

template <int A>
__global__ void gaussian_logp_kernel(int B, const float* __restrict__ mu, const float* __restrict__ logstd, const float* __restrict__ actions,
                                     float* __restrict__ logp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int a = 0; a < A; a++) {
        const float d = actions[(size_t)b * A + a] - mu[(size_t)b * A + a];
        const float sg = expf(logstd[a]);
        s += -0.5f * d * d / (sg * sg) - logstd[a] - kHalfLog2Pi;
    }
    logp[b] = s;
}

// ------------------------------------------------------------------ fused actor MLP + Gaussian sample (rollout inference), fp32 MFMA
// One workgroup (4 waves) = 16 observation rows; activations ping-pong between two LDS tiles; every wave owns output-neuron
// tiles of 16 and accumulates them with v_mfma_f32_16x16x4_f32 (exact fp32, D = A*B + C):
//     A[i = lane & 15][k = lane >> 4] = X[row i][k]          (from LDS, one ds_read_b128 feeds 4 MFMAs)
//     B[k = lane >> 4][j = lane & 15] = W[neuron j][k]       (straight from L2: torch layout [out][in], one 16-byte load feeds 4 MFMAs)
//     C/D: neuron j = lane & 15, row i = (lane >> 4) * 4 + reg
// The k index inside a group of 16 is permuted identically for A and B (k = 16 t + 4 (lane >> 4) + step), which a sum over k allows.
// 63,244 weights = 253 kB stay L2-resident; 992 MFMAs per workgroup.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int AROWS = 16;
constexpr int ALD = 260;  // LDS row stride (floats)
__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : expm1f(x); }

// A layer = its weight fetch (straight from L2 into registers, nothing of it depends on the activations) and its MFMAs.  The two are separate calls so that
// the kernel can issue layer L + 1's fetch BEFORE it computes layer L: with fetch and MFMAs back to back per tile (the first form of this kernel) a
// workgroup spent ~1 us of exposed L2 latency per tile, 9 tiles per wave, in a launch of 15 us that the rollout waits for 24 times per iteration.
// K % 16 == 0, weight rows 16-byte aligned; tiles of 16 neurons, tile = wave + 4 j.
template <int K, int OUT>
struct LayerW {
    static constexpr int TILES = (OUT + 15) / 16, NT = (TILES + 3) / 4, G = K / 16;
    float4 b[NT][G];
    float bias[NT];
    __device__ __forceinline__ void fetch(const float* __restrict__ W, const float* __restrict__ bv, int wave, int lane) {
        const int r = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int n = (wave + 4 * j) * 16 + r;
            const bool nv = wave + 4 * j < TILES && n < OUT;
            const float* wrow = W + (size_t)(nv ? n : 0) * K + 4 * kg;
#pragma unroll
            for (int t = 0; t < G; t++) {
                b[j][t] = *reinterpret_cast<const float4*>(wrow + 16 * t);
                if (!nv) b[j][t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            bias[j] = nv ? bv[n] : 0.f;
        }
    }
    template <bool ACT>
    __device__ __forceinline__ void run(const float* in /*LDS [16][ALD]*/, float* out /*LDS [16][ALD]*/, int wave, int lane) const {
        const int r = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int tile = wave + 4 * j;
            if (tile >= TILES) break;
            f32x4 acc = {bias[j], bias[j], bias[j], bias[j]};
#pragma unroll
            for (int t = 0; t < G; t++) {
                const float4 a = *reinterpret_cast<const float4*>(in + r * ALD + 16 * t + 4 * kg);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[j][t].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[j][t].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[j][t].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[j][t].w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float v = ACT ? elu(acc[q]) : acc[q];
                out[(kg * 4 + q) * ALD + tile * 16 + r] = v;
            }
        }
    }
};

// first layer: K = 47 (rows not 16-byte aligned, K not a multiple of 4): scalar operand loads, k padded to 48 with zeros
template <int K, int OUT>
struct FirstLayerW {
    static constexpr int TILES = OUT / 16, NT = TILES / 4, STEPS = (K + 3) / 4;
    float b[NT][STEPS];
    float bias[NT];
    __device__ __forceinline__ void fetch(const float* __restrict__ W, const float* __restrict__ bv, int wave, int lane) {
        const int r = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int n = (wave + 4 * j) * 16 + r;
#pragma unroll
            for (int s2 = 0; s2 < STEPS; s2++) { const int k = 4 * s2 + kg; b[j][s2] = k < K ? W[(size_t)n * K + k] : 0.f; }
            bias[j] = bv[n];
        }
    }
    __device__ __forceinline__ void run(const float* in, float* out, int wave, int lane) const {
        const int r = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const int tile = wave + 4 * j;
            f32x4 acc = {bias[j], bias[j], bias[j], bias[j]};
#pragma unroll
            for (int s2 = 0; s2 < STEPS; s2++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(in[r * ALD + 4 * s2 + kg], b[j][s2], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; q++) out[(kg * 4 + q) * ALD + tile * 16 + r] = elu(acc[q]);
        }
    }
};

__global__ __launch_bounds__(256) void actor_sample_kernel(int N, const float* __restrict__ obs, const float* __restrict__ w0,
                                                           const float* __restrict__ b0, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, const float* __restrict__ w3,
                                                           const float* __restrict__ b3, const float* __restrict__ logstd, uint64_t seed,
                                                           uint32_t counter, float* __restrict__ mu_out, float* __restrict__ act_out) {
    __shared__ __attribute__((aligned(16))) float bufA[AROWS * ALD];
    __shared__ __attribute__((aligned(16))) float bufB[AROWS * ALD];
    const int r0 = blockIdx.x * AROWS, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    FirstLayerW<BG_NUM_OBS, 256> l0;
    l0.fetch(w0, b0, wave, lane);
    LayerW<256, 128> l1;
    l1.fetch(w1, b1, wave, lane);
    for (int k = threadIdx.x; k < AROWS * 48; k += blockDim.x) {  // obs tile, k padded 47 -> 48 with zeros
        const int r = k / 48, c = k % 48;
        bufA[r * ALD + c] = (r0 + r < N && c < BG_NUM_OBS) ? obs[(size_t)(r0 + r) * BG_NUM_OBS + c] : 0.f;
    }
    __syncthreads();
    // same arithmetic and order as one fetch + MFMA pass per layer; only WHEN the weights are fetched differs: a layer ahead
    l0.run(bufA, bufB, wave, lane);
    LayerW<128, 128> l2;
    l2.fetch(w2, b2, wave, lane);
    __syncthreads();
    l1.template run<true>(bufB, bufA, wave, lane);
    LayerW<128, BG_NUM_DOFS> l3;
    l3.fetch(w3, b3, wave, lane);
    __syncthreads();
    l2.template run<true>(bufA, bufB, wave, lane);
    __syncthreads();
    l3.template run<false>(bufB, bufA, wave, lane);
    __syncthreads();
    // sample: one thread per (row, group of 4 actions)
    if (threadIdx.x < AROWS * 3) {
        const int r = threadIdx.x / 3, g = threadIdx.x % 3, row = r0 + r;
        if (row < N) {
            bg::Rand4 rn = bg::rand4(seed, (uint32_t)row, counter, bg::RS_ACTOR + g);
            for (int k = 0; k < 4; k++) {
                const int a = g * 4 + k;
                const float m = bufA[r * ALD + a];
                if (mu_out) mu_out[(size_t)row * BG_NUM_DOFS + a] = m;
                act_out[(size_t)row * BG_NUM_DOFS + a] = m + expf(logstd[a]) * rn.n[k];
            }
        }
    }
}

// ------------------------------------------------------------------ clip_grad_norm_ + Adam on a flat buffer
__global__ __launch_bounds__(256) void sqnorm_kernel(int n, const float* __restrict__ g, double* __restrict__ out) {
    __shared__ double sm[4];
    double acc[1] = {0.0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc[0] += (double)g[i] * (double)g[i];
    block_atomic_add<1>(acc, out, sm);
}
__global__ __launch_bounds__(256) void adam_kernel(int n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ lr_dev, float bc1, float bc2_sqrt, float beta1,
                                                   float beta2, float eps, float max_norm, const double* __restrict__ sqnorm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float total = (float)sqrt(*sqnorm);
    const float coef = max_norm > 0.f ? fminf(max_norm / (total + 1e-6f), 1.0f) : 1.0f;  // torch.nn.utils.clip_grad_norm_
    const float gi = g[i] * coef;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float step_size = *lr_dev / bc1;
    p[i] -= step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
}
__global__ void adapt_lr_kernel(const double* __restrict__ kl_sum, float count, float desired, float lr_min, float lr_max, float* __restrict__ lr) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float kl = (float)(kl_sum[0] / (double)count);
    float l = *lr;
    if (kl > desired * 2.0f) l = fmaxf(lr_min, l / 1.5f);
    else if (kl < desired / 2.0f) l = fminf(lr_max, l * 1.5f);
    *lr = l;
}

// The whole tail of a mini-epoch in ONE launch (reference utils/runner.py:162-180: clip_grad_norm_, Adam step, KL-adaptive learning rate, plus
// this build's bookkeeping of the loss statistics).  As separate launches (memset, sqnorm, adam, adapt_lr, a stats add, two fills) these seven tiny
// dependent kernels were a serial stretch of ~60 us per mini-epoch in which the GPU was otherwise idle: 1.2 ms of a 27 ms iteration.
//   phase 1  every workgroup computes the FULL squared gradient norm itself (the 712 kB bucket is L2-resident; OPT_GRID x 712 kB of L2 reads),
//            in the same order, so all workgroups hold the bit-identical norm and the result is deterministic (no atomics, no grid barrier);
//   phase 2  Adam on the workgroup's slice with the clip coefficient;
//   tail     the last workgroup to finish (ticket counter; every workgroup has read the learning rate before it takes its ticket) applies the KL rule
//            to the learning rate, publishes / accumulates the loss statistics and zeroes the accumulators for the next mini-epoch.
// grad_logstd (optional): the log-std gradient as the head kernels leave it (float64 [ls_n]); it stands for grads[ls_off .. ls_off + ls_n).
//   mirrors  every updated parameter inside one of the caller's [rows][cols] weight matrices is also written to that matrix's mirror: a transposed
//            copy (the operand layout of the backward layer kernel) or a copy with a wider row stride (the zero-padded first layer).  Six strided torch
//            copies per mini-epoch (5-20 us each, inside the two chains) otherwise.
constexpr int OPT_GRID = 64, OPT_THREADS = 1024;
__global__ __launch_bounds__(OPT_THREADS) void optimizer_step_kernel(int n, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                                     float* __restrict__ v, float* __restrict__ lr_dev, float bc1, float bc2_sqrt,
                                                                     float beta1, float beta2, float eps, float max_norm,
                                                                     double* __restrict__ grad_logstd, int ls_off, int ls_n,
                                                                     double* __restrict__ stats, double* __restrict__ stats_acc,
                                                                     double* __restrict__ stats_last, int n_stats, int kl_index, float kl_count,
                                                                     float desired_kl, float lr_min, float lr_max, unsigned* __restrict__ ticket,
                                                                     ParamMirrors mir) {
    __shared__ double s_part[OPT_THREADS / 64];
    __shared__ double s_total;
    const int t = threadIdx.x;
    const float lr = *lr_dev;  // read by every thread before this workgroup takes its ticket
    // the log-std gradient arrives in float64 from the heads: put it into the flat buffer (every workgroup writes the same values)
    if (grad_logstd && t < ls_n) g[ls_off + t] = (float)grad_logstd[t];
    __syncthreads();
    double acc = 0.0;
    const int n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
#pragma unroll 8  // 8 loads in flight (the adds stay in order): 44 dependent L2 round trips per thread otherwise
    for (int i = t; i < n4; i += OPT_THREADS) {
        const float4 x = g4[i];
        acc += (double)x.x * (double)x.x + (double)x.y * (double)x.y + (double)x.z * (double)x.z + (double)x.w * (double)x.w;
    }
    for (int i = (n4 << 2) + t; i < n; i += OPT_THREADS) acc += (double)g[i] * (double)g[i];
    acc = wave_sum(acc);
    if ((t & 63) == 0) s_part[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        double tot = 0.0;
        for (int w = 0; w < OPT_THREADS / 64; w++) tot += s_part[w];
        s_total = tot;
    }
    __syncthreads();
    const float total = (float)sqrt(s_total);
    const float coef = max_norm > 0.f ? fminf(max_norm / (total + 1e-6f), 1.0f) : 1.0f;  // torch.nn.utils.clip_grad_norm_
    const float step_size = lr / bc1;
    const int per = (n + OPT_GRID - 1) / OPT_GRID, i0 = blockIdx.x * per, i1 = min(n, i0 + per);
    for (int i = i0 + t; i < i1; i += OPT_THREADS) {
        const float gi = g[i] * coef;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float pn = p[i] - step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = pn;
        for (int k = 0; k < mir.n; k++) {
            const bg_param_mirror& mm = mir.m[k];
            const int j = i - mm.offset;
            if (j >= 0 && j < mm.rows * mm.cols) {
                const int r = j / mm.cols, c = j - r * mm.cols;
                bg_mirror_write(mm, r, c, pn);
            }
        }
    }
    __syncthreads();
    if (t == 0) {
        // (no fence: the last workgroup needs the others' READS of lr and the log-std gradient to be over, which they are behind the barrier above --
        // nothing they wrote; see tail_adam_kernel)
        const unsigned k = atomicAdd(ticket, 1u);
        if (k == gridDim.x - 1) {  // every workgroup has read lr and finished its slice
            if (stats) {
                const float kl = (float)(stats[kl_index] / (double)kl_count);
                float l = lr;
                if (kl > desired_kl * 2.0f) l = fmaxf(lr_min, l / 1.5f);
                else if (kl < desired_kl / 2.0f) l = fminf(lr_max, l * 1.5f);
                *lr_dev = l;
                for (int j = 0; j < n_stats; j++) {
                    const double sj = stats[j];
                    stats_last[j] = sj;
                    stats_acc[j] += sj;
                    stats[j] = 0.0;
                }
            }
            if (grad_logstd) for (int j = 0; j < ls_n; j++) grad_logstd[j] = 0.0;
            *ticket = 0u;
        }
    }
}


// ------------------------------------------------------------------ MLP backward helper: g <- g * elu'(a) in place and db = column sums of g
// (replaces torch's elu_backward + the separate bias-gradient reduction: the gradient tile is read once).  elu'(z) expressed through the
// OUTPUT a = elu(z): 1 if a > 0 else a + 1.  Deterministic two-stage column sum (no float atomics).
constexpr int CS_ROWS = 128;
// generic (any C): one thread per column, used for the narrow output layers (C = 12, 1)
__global__ __launch_bounds__(256) void elu_bwd_colsum_kernel(int B, int C, float* __restrict__ g, const float* __restrict__ act,
                                                             float* __restrict__ partial /*[gridDim.x][C]*/) {
    const int r0 = blockIdx.x * CS_ROWS, r1 = min(B, r0 + CS_ROWS);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int r = r0; r < r1; r++) {
            const size_t k = (size_t)r * C + c;
            float v = g[k];
            if (act) { const float a = act[k]; v *= a > 0.f ? 1.0f : a + 1.0f; g[k] = v; }
            acc += v;
        }
        partial[(size_t)blockIdx.x * C + c] = acc;
    }
}
// C % 4 == 0 and C <= 256: 16-byte accesses, 1 KiB per wave instruction, 4 independent rows in flight per thread.
// thread = (row lane, column group of 4); a 256-thread block covers CS_ROWS rows.
__global__ __launch_bounds__(256) void elu_bwd_colsum_vec4_kernel(int B, int C, float* __restrict__ g, const float* __restrict__ act,
                                                                  float* __restrict__ partial) {
    __shared__ float4 sm[256];
    const int groups = C >> 2;                 // column groups per row (<= 64)
    const int lanes = 256 / groups;            // rows processed concurrently by the block
    const int cg = threadIdx.x % groups, rl = threadIdx.x / groups;
    const int r0 = blockIdx.x * CS_ROWS, r1 = min(B, r0 + CS_ROWS);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < lanes) {
        int r = r0 + rl;
        for (; r + 3 * lanes < r1; r += 4 * lanes) {
            float4 gv[4], av[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const size_t k = ((size_t)(r + u * lanes) * C) / 4 + cg;
                gv[u] = reinterpret_cast<const float4*>(g)[k];
                av[u] = reinterpret_cast<const float4*>(act)[k];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const size_t k = ((size_t)(r + u * lanes) * C) / 4 + cg;
                gv[u].x *= av[u].x > 0.f ? 1.0f : av[u].x + 1.0f; gv[u].y *= av[u].y > 0.f ? 1.0f : av[u].y + 1.0f;
                gv[u].z *= av[u].z > 0.f ? 1.0f : av[u].z + 1.0f; gv[u].w *= av[u].w > 0.f ? 1.0f : av[u].w + 1.0f;
                reinterpret_cast<float4*>(g)[k] = gv[u];
                acc.x += gv[u].x; acc.y += gv[u].y; acc.z += gv[u].z; acc.w += gv[u].w;
            }
        }
        for (; r < r1; r += lanes) {
            const size_t k = ((size_t)r * C) / 4 + cg;
            float4 gv = reinterpret_cast<const float4*>(g)[k];
            const float4 av = reinterpret_cast<const float4*>(act)[k];
            gv.x *= av.x > 0.f ? 1.0f : av.x + 1.0f; gv.y *= av.y > 0.f ? 1.0f : av.y + 1.0f;
            gv.z *= av.z > 0.f ? 1.0f : av.z + 1.0f; gv.w *= av.w > 0.f ? 1.0f : av.w + 1.0f;
            reinterpret_cast<float4*>(g)[k] = gv;
            acc.x += gv.x; acc.y += gv.y; acc.z += gv.z; acc.w += gv.w;
        }
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < lanes; l++) {
            const float4 o = sm[l * groups + cg];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * C)[cg] = acc;
    }
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(int nb, int C, const float* __restrict__ partial, float* __restrict__ out) {
    __shared__ float sm[256];
    const int c = blockIdx.x;
    float acc = 0.f;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) acc += partial[(size_t)b * C + c];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[c] = sm[0];
}

// ------------------------------------------------------------------ ABI
extern "C" int bg_gae(int32_t T, int32_t N, float* rewards, const uint8_t* dones, const uint8_t* time_outs, const float* values,
                      const float* last_values, float gamma, float lam, float* advantages, float* returns, double* sums, void* stream) {
    if (T <= 0 || N <= 0 || !rewards || !dones || !time_outs || !values || !last_values || !advantages || !returns || !sums)
        return bg_set_error(-1, "bg_gae: bad argument");
    if (T <= 32)
        hipLaunchKernelGGL(gae_kernel<32>, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, T, N, rewards, dones, time_outs, values, last_values,
                           gamma, lam, advantages, returns, sums);
    else
        hipLaunchKernelGGL(gae_kernel_loop, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, N, rewards, dones, time_outs, values, last_values,
                           gamma, lam, advantages, returns, sums);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_ppo_loss(int32_t B, int32_t A, const float* mu, const float* logstd, const float* actions, const float* old_mu,
                           const float* old_logstd, const float* old_logp, const float* adv, const double* adv_stats, const float* values,
                           const float* returns, float e_clip, float bound_coef, float entropy_coef, float* grad_mu, float* grad_values,
                           double* grad_logstd, double* stats, void* stream) {
    if (B <= 0 || !mu || !logstd || !actions || !old_mu || !old_logstd || !old_logp || !adv || !adv_stats || !values || !returns || !grad_mu ||
        !grad_values || !grad_logstd || !stats)
        return bg_set_error(-1, "bg_ppo_loss: bad argument");
    if (A != BG_NUM_DOFS) return bg_set_error(-1, "bg_ppo_loss: this build is compiled for 12 actions");
    hipLaunchKernelGGL(ppo_loss_kernel<BG_NUM_DOFS>, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, mu, logstd, actions, old_mu,
                       old_logstd, old_logp, adv, adv_stats, values, returns, e_clip, bound_coef, entropy_coef, grad_mu, grad_values, grad_logstd,
                       stats);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_gaussian_logp(int32_t B, int32_t A, const float* mu, const float* logstd, const float* actions, float* logp, void* stream) {
    if (B <= 0 || !mu || !logstd || !actions || !logp) return bg_set_error(-1, "bg_gaussian_logp: bad argument");
    if (A != BG_NUM_DOFS) return bg_set_error(-1, "bg_gaussian_logp: this build is compiled for 12 actions");
    hipLaunchKernelGGL(gaussian_logp_kernel<BG_NUM_DOFS>, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, mu, logstd, actions, logp);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_actor_sample(int32_t N, const float* obs, const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                               const float* b2, const float* w3, const float* b3, const float* logstd, uint64_t seed, uint64_t counter, float* mu,
                               float* actions, void* stream) {
    if (N <= 0 || !obs || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !logstd || !actions)
        return bg_set_error(-1, "bg_actor_sample: bad argument");
    if (((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)w3) & 15) return bg_set_error(-1, "bg_actor_sample: weight matrices must be 16-byte aligned");
    hipLaunchKernelGGL(actor_sample_kernel, dim3((N + AROWS - 1) / AROWS), dim3(256), 0, (hipStream_t)stream, N, obs, w0, b0, w1, b1, w2, b2, w3, b3,
                       logstd, seed, (uint32_t)counter, mu, actions);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_adam_step(int32_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const float* lr_device, int32_t step,
                            float beta1, float beta2, float eps, float max_grad_norm, double* gnorm_scratch, void* stream) {
    if (n <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !lr_device || !gnorm_scratch || step < 1)
        return bg_set_error(-1, "bg_adam_step: bad argument");
    HIP_OK(hipMemsetAsync(gnorm_scratch, 0, sizeof(double), (hipStream_t)stream));
    int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks < 256 ? blocks : 256), dim3(256), 0, (hipStream_t)stream, n, grads, gnorm_scratch);
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, params, grads, exp_avg, exp_avg_sq, lr_device, bc1, bc2s,
                       beta1, beta2, eps, max_grad_norm, gnorm_scratch);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_adapt_lr(const double* kl_sum, float count, float desired_kl, float lr_min, float lr_max, float* lr_device, void* stream) {
    if (!kl_sum || !lr_device || !(count > 0.f)) return bg_set_error(-1, "bg_adapt_lr: bad argument");
    hipLaunchKernelGGL(adapt_lr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, kl_sum, count, desired_kl, lr_min, lr_max, lr_device);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_optimizer_step(int32_t n, float* params, float* grads, float* exp_avg, float* exp_avg_sq, float* lr_device, int32_t step, float beta1,
                                 float beta2, float eps, float max_grad_norm, double* grad_logstd, int32_t ls_off, int32_t ls_n, double* stats,
                                 double* stats_acc, double* stats_last, int32_t n_stats, int32_t kl_index, float kl_count, float desired_kl,
                                 float lr_min, float lr_max, uint32_t* ticket, const bg_param_mirror* mirrors, int32_t n_mirrors, void* stream) {
    if (n <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !lr_device || !ticket || step < 1) return bg_set_error(-1, "bg_optimizer_step: bad argument");
    if (n_mirrors < 0 || n_mirrors > OPT_MAX_MIRRORS || (n_mirrors > 0 && !mirrors)) return bg_set_error(-1, "bg_optimizer_step: 0 to 16 mirrors");
    ParamMirrors mir;
    mir.n = n_mirrors;
    for (int k = 0; k < n_mirrors; k++) {
        const bg_param_mirror& q = mirrors[k];
        if (!bg_mirror_ok(q, n)) return bg_set_error(-1, "bg_optimizer_step: bad mirror descriptor");
        mir.m[k] = q;
    }
    if ((((uintptr_t)grads) & 15) != 0) return bg_set_error(-1, "bg_optimizer_step: grads must be 16-byte aligned");
    if (grad_logstd && (ls_n <= 0 || ls_n > OPT_THREADS || ls_off < 0 || ls_off + ls_n > n)) return bg_set_error(-1, "bg_optimizer_step: log-std slice out of range");
    if (stats && (!stats_acc || !stats_last || n_stats <= 0 || kl_index < 0 || kl_index >= n_stats || !(kl_count > 0.f)))
        return bg_set_error(-1, "bg_optimizer_step: statistics arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(optimizer_step_kernel, dim3(OPT_GRID), dim3(OPT_THREADS), 0, (hipStream_t)stream, n, params, grads, exp_avg, exp_avg_sq, lr_device, bc1,
                       bc2s, beta1, beta2, eps, max_grad_norm, grad_logstd, ls_off, ls_n, stats, stats_acc, stats_last, n_stats, kl_index, kl_count,
                       desired_kl, lr_min, lr_max, ticket, mir);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_elu_backward_colsum(int32_t B, int32_t C, float* grad, const float* act, float* colsum, float* scratch, void* stream) {
    if (B <= 0 || C <= 0 || !grad || !colsum || !scratch) return bg_set_error(-1, "bg_elu_backward_colsum: bad argument");
    const int nb = (B + CS_ROWS - 1) / CS_ROWS;
    if (act && C % 4 == 0 && C <= 256 && 256 % (C / 4) == 0)
        hipLaunchKernelGGL(elu_bwd_colsum_vec4_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, B, C, grad, act, scratch);
    else
        hipLaunchKernelGGL(elu_bwd_colsum_kernel, dim3(nb), dim3(C >= 256 ? 256 : (C >= 128 ? 128 : 64)), 0, (hipStream_t)stream, B, C, grad, act, scratch);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, nb, C, scratch, colsum);
    HIP_OK(hipGetLastError());
    return 0;
}
