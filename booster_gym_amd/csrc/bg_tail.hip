// The tail of a mini-epoch (reference utils/runner.py:162-180: loss.backward()'s last sums, clip_grad_norm_, optimizer.step(), the KL rule) as TWO
// launches behind the main weight-gradient kernel, where rounds 2-3 had three around it: reduce_group_kernel (bg_head.hip) in front of the weight
// gradients, mlp_wgrad_group_finish_kernel (bg_wgrad.hip) behind them and optimizer_step_kernel (bg_ppo.hip), 5.3 + 5.6 + 20.4 us of kernels and three
// launch-to-launch gaps of 7-8 us on the critical path of every mini-epoch (tools/timeline.py on the round-3 trace).
//
//   tail_sums_kernel   one workgroup per block of either kernel's work -- the weight gradients' fixed-order sums over their slices and the deferred
//                      reductions (head / bias gradients, float64 loss statistics), the arithmetic and the order of the two kernels named above,
//                      statement for statement: the gradients are the same bits -- and, while the finished values are in registers, the block's sum
//                      of their squares (float64) into one slot of norm_partial.  Every element of the flat gradient except the log-std's is
//                      written by exactly one block, so the slots add up to the squared global norm;
//   tail_adam_kernel   every workgroup adds the slots in slot order (+ the log-std's squares): the same total everywhere, deterministic, no
//                      atomics -- optimizer_step_kernel spends half of its 20 us re-reading the whole gradient in each of its 64 workgroups for this --
//                      then clip + Adam + weight mirrors on its slice, the KL rule and the statistics' bookkeeping by the last workgroup to take a ticket.
// (First form of this file: ONE launch with two grid-wide barriers between the three phases.  Correct, and 35 us per mini-epoch SLOWER than the three
// launches it replaced (update 21.7 against 21.0 ms, tools/ab_env.sh): an agent-scope fence on this part writes back and invalidates the XCD's whole
// L2, 256 workgroups did that twice each while the others were reading.  Launch boundaries are the cheaper grid barrier.)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/booster_gym_amd.h"
#include "bg_mirror.h"
#include "bg_wgrad.h"

extern int bg_set_error(int code, const char* msg);
extern int bg_wgrad_group_fill(const bg_wgrad_problem* problems, int32_t count, WgradGroup& grp, int& wg, int& fin, const char* who);

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
// ADAM_GRID: one workgroup per CU.  Every workgroup adds the ~2,900 norm slots itself (23 KB out of L2, the same order everywhere: the same total), so
// the count changes no bit; with the weight copies the launch writes since round 6 (bf16 planes of W, -W, W^T, -W^T: up to 12 two-byte stores per
// parameter, the transposed ones a cache line each) 256 workgroups instead of 64 are worth 0.1 ms per iteration (21.82-21.90 against 21.95-22.00 ms).
constexpr int ADAM_GRID = 256, TAIL_THREADS = 256, RG_MAX = 8, TAIL_MAX_ITEMS = 8192;
struct ReduceGroup { int np; int begin[RG_MAX]; bg_reduce_problem p[RG_MAX]; };
struct OptArgs {
    int n; float *p, *g, *m, *v, *lr_dev; float bc1, bc2_sqrt, beta1, beta2, eps, max_norm;
    double* grad_logstd; int ls_off, ls_n;
    double *stats, *stats_acc, *stats_last; int n_stats, kl_index; float kl_count, desired_kl, lr_min, lr_max;
};

__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// one block of mlp_wgrad_group_finish_kernel (bg_wgrad.hip): dW element group e4 = sum over the slices, 16 slice groups x 16 float4 per workgroup
// returns (threads 0..15: the others 0) the squares of the values this thread wrote
__device__ __forceinline__ double wgrad_finish_block(const WgradGroup& grp, int b, f32x4 (*sm)[16]) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < WG_MAX_PROBLEMS; j++)
        if (j < grp.np && b >= grp.p[j].fin_begin) k = j;
    const WgradProblem& pr = grp.p[k];
    const int c = threadIdx.x & 15, sg = threadIdx.x >> 4, e4 = (b - pr.fin_begin) * 16 + c, n4 = pr.n4, S = pr.slices, Cin = pr.Cin, Cin_real = pr.Cin_real;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (e4 < n4)
        for (int s = sg; s < S; s += 16) acc += *reinterpret_cast<const f32x4*>(pr.P + ((size_t)s * n4 + e4) * 4);
    sm[sg][c] = acc;
    __syncthreads();
    if (sg == 0 && e4 < n4) {
        f32x4 v = sm[0][c];
#pragma unroll
        for (int j = 1; j < 16; j++) v += sm[j][c];
        const int row = (e4 * 4) / Cin, col = (e4 * 4) % Cin;
        double q = 0.0;
        if (Cin_real == Cin) {
            *reinterpret_cast<f32x4*>(pr.dW + (size_t)row * Cin + col) = v;
            q = (double)v[0] * (double)v[0] + (double)v[1] * (double)v[1] + (double)v[2] * (double)v[2] + (double)v[3] * (double)v[3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (col + j < Cin_real) { pr.dW[(size_t)row * Cin_real + col + j] = v[j]; q += (double)v[j] * (double)v[j]; }
        }
        return q;
    }
    return 0.0;
}
// one block of reduce_group_kernel (bg_head.hip): 16 outputs x 16 record slices, or one float64 statistic
__device__ __forceinline__ double reduce_block(const ReduceGroup& grp, int blk, float (*sm)[17], double* sd) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < RG_MAX; j++)
        if (j < grp.np && blk >= grp.begin[j]) k = j;
    const bg_reduce_problem& pr = grp.p[k];
    const int b = blk - grp.begin[k], nsum = (pr.n_out + 15) / 16;
    if (b >= nsum) {
        const int ks = b - nsum;
        if ((pr.stat_skip >> ks) & 1u) return 0.0;
        const double* sp = reinterpret_cast<const double*>(pr.partial + pr.stat_base);
        double s = 0.0;
#pragma unroll 4
        for (int g = threadIdx.x; g < pr.groups; g += 256) s += sp[(size_t)ks * pr.groups + g];
        s = wave_sum_d(s);
        if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double v = sd[0] + sd[1] + sd[2] + sd[3];
            if (ks < pr.n_ls) atomicAdd(&pr.grad_logstd[ks], v + pr.entropy_coef);
            else atomicAdd(&pr.stats[ks - pr.n_ls], v);
        }
        return 0.0;  // (statistics are not gradients; the log-std's gradient is squared by tail_adam_kernel, once every statistic has been added)
    }
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
        return (double)v * (double)v;
    }
    return 0.0;
}

__global__ __launch_bounds__(TAIL_THREADS) void tail_sums_kernel(WgradGroup wg, ReduceGroup rg, int rg_blocks, double* __restrict__ norm_partial) {
    __shared__ f32x4 sm4[16][16];
    __shared__ float smf[16][17];
    __shared__ double sd[4];
    const int item = blockIdx.x;
    double q = item < rg_blocks ? reduce_block(rg, item, smf, sd) : wgrad_finish_block(wg, item - rg_blocks, sm4);
    // the writers are threads 0..15 = the first lanes of wave 0: a butterfly over that wave (zeros elsewhere) in a fixed order
    if (threadIdx.x < 64) {
        q = wave_sum_d(q);
        if (threadIdx.x == 0) norm_partial[item] = q;
    }
}

__global__ __launch_bounds__(1024) void tail_adam_kernel(OptArgs o, ParamMirrors mir, const double* __restrict__ norm_partial, int n_partial,
                                                         unsigned* __restrict__ ticket) {
    __shared__ double s_part[16];
    __shared__ double s_total;
    const int t = threadIdx.x, G = gridDim.x;
    const float lr = *o.lr_dev;  // read by every thread before this workgroup takes its ticket
    // the log-std gradient arrives in float64 from the heads: put it into the flat buffer (every workgroup writes the same values) ...
    double acc = 0.0;
    if (o.grad_logstd && t < o.ls_n) {
        const float gl = (float)o.grad_logstd[t];
        o.g[o.ls_off + t] = gl;
        acc = (double)gl * (double)gl;  // ... and its share of the norm, the only elements tail_sums_kernel has not squared
    }
    for (int i = t; i < n_partial; i += 1024) acc += norm_partial[i];
    acc = wave_sum_d(acc);
    if ((t & 63) == 0) s_part[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        double tot = 0.0;
        for (int w = 0; w < 16; w++) tot += s_part[w];
        s_total = tot;
    }
    __syncthreads();
    const float total = (float)sqrt(s_total);
    const float coef = o.max_norm > 0.f ? fminf(o.max_norm / (total + 1e-6f), 1.0f) : 1.0f;  // torch.nn.utils.clip_grad_norm_
    const float step_size = lr / o.bc1;
    const int per = (o.n + G - 1) / G, i0 = blockIdx.x * per, i1 = min(o.n, i0 + per);
    for (int i = i0 + t; i < i1; i += 1024) {
        const float gi = o.g[i] * coef;
        const float mi = o.beta1 * o.m[i] + (1.0f - o.beta1) * gi;
        const float vi = o.beta2 * o.v[i] + (1.0f - o.beta2) * gi * gi;
        o.m[i] = mi; o.v[i] = vi;
        const float pn = o.p[i] - step_size * mi / (sqrtf(vi) / o.bc2_sqrt + o.eps);
        o.p[i] = pn;
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
        // No fence in front of the ticket: the last workgroup needs nothing the others WROTE, only that they have READ lr and the log-std gradient --
        // and those loads have returned (their values went into every thread's arithmetic in front of the barrier above).  An agent-scope
        // release here writes back the XCD's whole L2, once per workgroup.
        const unsigned k = atomicAdd(ticket, 1u);
        if (k == (unsigned)G - 1u) {  // every workgroup has read lr and the log-std gradient and finished its slice
            if (o.stats) {
                const float kl = (float)(o.stats[o.kl_index] / (double)o.kl_count);
                float l = lr;
                if (kl > o.desired_kl * 2.0f) l = fmaxf(o.lr_min, l / 1.5f);
                else if (kl < o.desired_kl / 2.0f) l = fminf(o.lr_max, l * 1.5f);
                *o.lr_dev = l;
                for (int j = 0; j < o.n_stats; j++) {
                    const double sj = o.stats[j];
                    o.stats_last[j] = sj;
                    o.stats_acc[j] += sj;
                    o.stats[j] = 0.0;
                }
            }
            if (o.grad_logstd) for (int j = 0; j < o.ls_n; j++) o.grad_logstd[j] = 0.0;
            *ticket = 0u;
        }
    }
}
}  // namespace

// the work list of tail_sums_kernel from the two descriptor lists (shared by bg_update_tail and bg_update_tail_sums)
static int tail_sums_fill(const bg_wgrad_problem* wgrad, int32_t n_wgrad, const bg_reduce_problem* reduce, int32_t n_reduce, WgradGroup& wg, ReduceGroup& rg,
                          int& blocks, int& fin, const char* who) {
    if (n_reduce < 0 || n_reduce > RG_MAX || (n_reduce > 0 && !reduce)) return bg_set_error(-1, "bg_update_tail: at most 8 reductions");
    wg.np = 0;
    int wgs = 0;
    fin = 0;
    if (n_wgrad > 0) {
        const int rc = bg_wgrad_group_fill(wgrad, n_wgrad, wg, wgs, fin, who);
        if (rc) return rc;
    }
    rg.np = n_reduce;
    blocks = 0;
    for (int k = 0; k < n_reduce; k++) {
        const bg_reduce_problem& q = reduce[k];
        if (!q.partial || q.groups <= 0 || q.record <= 0 || q.n_out <= 0 || q.n_out > q.record || !q.out[0] || q.n[0] <= 0 ||
            q.n[0] + q.n[1] + q.n[2] != q.n_out || (q.n[1] > 0 && !q.out[1]) || (q.n[2] > 0 && !q.out[2]))
            return bg_set_error(-1, "bg_update_tail: bad reduction descriptor");
        if (q.n_stat < 0 || q.n_stat > 32 || (q.n_stat > 0 && (!q.stats || (q.n_ls > 0 && !q.grad_logstd) || (q.stat_base & 1))))
            return bg_set_error(-1, "bg_update_tail: bad statistics descriptor");
        rg.begin[k] = blocks;
        rg.p[k] = q;
        blocks += (q.n_out + 15) / 16 + q.n_stat;
    }
    for (int k = n_reduce; k < RG_MAX; k++) rg.begin[k] = blocks;
    if (blocks + fin > TAIL_MAX_ITEMS) return bg_set_error(-4, "bg_update_tail: more than 8192 blocks of sums (norm_scratch holds one slot per block)");
    return 0;
}
// Launch (1) of bg_update_tail alone: the sums.  For the ranks of a multi-GPU job, whose gradient is averaged over the ranks between the sums and the
// optimiser: bg_update_tail_sums, the all-reduce, bg_optimizer_step (which takes the norm of the AVERAGED gradient itself).
extern "C" int bg_update_tail_sums(const bg_wgrad_problem* wgrad, int32_t n_wgrad, const bg_reduce_problem* reduce, int32_t n_reduce, double* norm_scratch,
                                   void* stream) {
    if (!norm_scratch) return bg_set_error(-1, "bg_update_tail_sums: bad argument");
    WgradGroup wg;
    ReduceGroup rg;
    int blocks = 0, fin = 0;
    const int rc = tail_sums_fill(wgrad, n_wgrad, reduce, n_reduce, wg, rg, blocks, fin, "bg_update_tail_sums");
    if (rc) return rc;
    if (blocks + fin > 0) hipLaunchKernelGGL(tail_sums_kernel, dim3(blocks + fin), dim3(TAIL_THREADS), 0, (hipStream_t)stream, wg, rg, blocks, norm_scratch);
    if (hipGetLastError() != hipSuccess) return bg_set_error(-2, "bg_update_tail_sums: launch failed");
    return 0;
}
extern "C" int bg_update_tail(const bg_wgrad_problem* wgrad, int32_t n_wgrad, const bg_reduce_problem* reduce, int32_t n_reduce, int32_t n, float* params,
                              float* grads, float* exp_avg, float* exp_avg_sq, float* lr_device, int32_t step, float beta1, float beta2, float eps,
                              float max_grad_norm, double* grad_logstd, int32_t ls_off, int32_t ls_n, double* stats, double* stats_acc, double* stats_last,
                              int32_t n_stats, int32_t kl_index, float kl_count, float desired_kl, float lr_min, float lr_max, uint32_t* sync,
                              double* norm_scratch, const bg_param_mirror* mirrors, int32_t n_mirrors, void* stream) {
    if (n <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !lr_device || step <= 0 || !sync || !norm_scratch)
        return bg_set_error(-1, "bg_update_tail: bad argument");
    if (grad_logstd && (ls_n <= 0 || ls_n > 1024 || ls_off < 0 || ls_off + ls_n > n)) return bg_set_error(-1, "bg_update_tail: log-std slice outside the buffer");
    if (stats && (!stats_acc || !stats_last || n_stats <= 0 || kl_index < 0 || kl_index >= n_stats || !(kl_count > 0.f)))
        return bg_set_error(-1, "bg_update_tail: statistics arguments");
    if (n_mirrors < 0 || n_mirrors > OPT_MAX_MIRRORS || (n_mirrors > 0 && !mirrors)) return bg_set_error(-1, "bg_update_tail: at most 16 weight mirrors");
    WgradGroup wg;
    ReduceGroup rg;
    int blocks = 0, fin = 0;
    {
        const int rc = tail_sums_fill(wgrad, n_wgrad, reduce, n_reduce, wg, rg, blocks, fin, "bg_update_tail");
        if (rc) return rc;
    }
    ParamMirrors mir;
    mir.n = n_mirrors;
    for (int k = 0; k < n_mirrors; k++) {
        const bg_param_mirror& q = mirrors[k];
        if (!bg_mirror_ok(q, n)) return bg_set_error(-1, "bg_update_tail: bad weight mirror");
        mir.m[k] = q;
    }
    OptArgs o;
    o.n = n; o.p = params; o.g = grads; o.m = exp_avg; o.v = exp_avg_sq; o.lr_dev = lr_device;
    o.bc1 = 1.0f - powf(beta1, (float)step); o.bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    o.beta1 = beta1; o.beta2 = beta2; o.eps = eps; o.max_norm = max_grad_norm;
    o.grad_logstd = grad_logstd; o.ls_off = ls_off; o.ls_n = ls_n;
    o.stats = stats; o.stats_acc = stats_acc; o.stats_last = stats_last; o.n_stats = n_stats; o.kl_index = kl_index; o.kl_count = kl_count;
    o.desired_kl = desired_kl; o.lr_min = lr_min; o.lr_max = lr_max;
    if (blocks + fin > 0) hipLaunchKernelGGL(tail_sums_kernel, dim3(blocks + fin), dim3(TAIL_THREADS), 0, (hipStream_t)stream, wg, rg, blocks, norm_scratch);
    hipLaunchKernelGGL(tail_adam_kernel, dim3(ADAM_GRID), dim3(1024), 0, (hipStream_t)stream, o, mir, (const double*)norm_scratch, blocks + fin, (unsigned*)sync);
    if (hipGetLastError() != hipSuccess) return bg_set_error(-2, "bg_update_tail: launch failed");
    return 0;
}
