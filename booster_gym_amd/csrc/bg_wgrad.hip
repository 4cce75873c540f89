// Weight-gradient kernel of the PPO update's MLPs (reference utils/runner.py:163 through utils/model.py:9-26), gfx950 only.
// Own translation unit: built with the default machine scheduler (Makefile) -- under -amdgpu-sched-strategy=max-ilp, which the lane-per-env
// simulator kernels need, the 256-accumulator MFMA loop below was spilled to scratch.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/booster_gym_amd.h"
#include "bg_wgrad.h"

extern int bg_set_error(int code, const char* msg);
#define HIP_OK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return bg_set_error(-2, hipGetErrorString(_e));               \
    } while (0)

#ifdef BG_PROBE_TIMELINE  // tools/archive/wgrad_timeline_probe.py: shader-clock stamps of every wave at the phase boundaries (never defined in the product build)
__device__ long long bg_wg_timeline[1024 * 4 * 8];
#define BG_STAMP(SLOT) do { if ((threadIdx.x & 63) == 0) bg_wg_timeline[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (SLOT)] = clock64(); } while (0)
extern "C" int bg_probe_read_wgrad_timeline(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(bg_wg_timeline), bytes); }
#else
#define BG_STAMP(SLOT) do { } while (0)
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in registers (HIP's float4 struct blocked SROA here)

// ---------------------------------------------------------------------------------------------------------------------------------------
// Weight gradient dW[C_out][C_in] = G^T A over the batch (reference utils/runner.py:163 `loss.backward()` through model.py:9-26's Linear
// layers): G [M][C_out] = dL/dz of the layer, A [M][C_in] = its input activations, M = 98,304 rows, C_* in {64, 128, 256}.
// The reduction dimension is the BATCH, and both operands are row-major over it, which is exactly the MFMA operand layout: for
// v_mfma_f32_32x32x2_f32 lane (i = lane & 31, h = lane >> 5) supplies A-operand element [i][k = h] and B-operand element [k = h][i]; with
// k = batch row 2*kp + h a lane loads 16 bytes of row (2*kp + h) of G and of A, and the four floats serve four different output tiles
// (output columns 4*i + t: a fixed permutation of the output, undone by the store).  So ONE 16-byte load of each operand feeds 16 MFMAs of
// a 128 x 128 output tile held in 256 accumulator registers: no LDS staging, no barriers in the main loop, 2 KB of loads per 65,536 flop.
// Split over the batch inside the launch: every wave owns a contiguous run of row pairs, the 4 waves of a workgroup add their tiles through
// LDS, one partial tile per workgroup goes to `P[slice]`, and a second kernel adds the slices in a fixed order (deterministic; no atomics).
// Workgroups that read the same rows (the tiles of one slice) are given block indices 8 apart, i.e. the same XCD and L2 (speed only).
constexpr int WG_RED_FLOATS = 4 * 64 * 16 * 2 * 4;  // LDS of the in-workgroup reduction: 4 waves x 2 co-tiles x 16 registers x 64 lanes x TCI (<= 4) floats = 128 KB
template <int TCI>  // 32-column C_in tiles per wave: 4 (128 input columns per workgroup) or 2 (64: the zero-padded first layers)
// A workgroup's 4 waves cover `tw` output tiles (tile0 .. tile0 + tw - 1) x ks = 4 / tw sub-ranges of the slice's rows: tw = 1 is the pure split
// over rows; with tw = 2 or 4 the waves that work on the SAME rows (different tiles) fetch each G / A row block once per CU instead of once per
// tile -- at 32 flop per loaded byte a 128 x 128 tile per wave sits on the machine's flop / byte ridge, so layers with several tiles trade
// partial-tile traffic (4 / ks times more of it) for input traffic (bg_mlp_weight_grad_group's planner picks tw per layer).
__device__ __forceinline__ void wgrad_tile(float* red_base, int M, int Cout, int Cin, const float* __restrict__ G, const float* __restrict__ A,
                                           float* __restrict__ P, int ntile_ci, int tile0, int tw, int slice, int slices) {
    constexpr int D = 8;  // row pairs per register set: 2 sets x D x (4 + TCI) registers of loads in flight under D x 4 x TCI MFMAs
    typedef float avec __attribute__((ext_vector_type(TCI)));
    float (*red)[64 * 16 * 2 * TCI] = reinterpret_cast<float (*)[64 * 16 * 2 * TCI]>(red_base);
    // wave-uniform values are forced into SGPRs: the loads then take the scalar-base + 32-bit lane offset form and the walk over the rows
    // is scalar arithmetic (per-lane 64-bit addresses for 2 x D x 2 loads in flight do not fit beside 256 accumulators)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int ks = 4 / tw, tl = wave % tw, ksub = wave / tw, tile = tile0 + tl;
    const int tco = tile / ntile_ci, tci = tile % ntile_ci;
    // runs are cut at multiples of 2 * D row pairs (one double set of the steady-state loop), the last one takes what is left: only that run has a
    // tail.  (Cut at arbitrary row pairs every wave ran the guarded tail -- two sets of mostly zero rows behind loads that are waited for one by
    // one: 22.8 k of a wave's 637 k cycles, tools/archive/wgrad_timeline_probe.py.)
    const long KP = M >> 1, W = (long)slices * ks, widx = (long)slice * ks + ksub, KG = KP / 16;
    const int kp0 = 16 * (int)(KG * widx / W), kp1 = widx == W - 1 ? (int)KP : 16 * (int)(KG * (widx + 1) / W);
    const unsigned gofb = 4u * (h * Cout + tco * 128 + 4 * i), aofb = 4u * (h * Cin + tci * (32 * TCI) + TCI * i);  // byte offsets of this lane
    const char* Gb = reinterpret_cast<const char*>(G);
    const char* Ab = reinterpret_cast<const char*>(A);
    f32x16 acc[4][TCI];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int u = 0; u < TCI; u++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][u][r] = 0.f;
    f32x4 g0[D], g1[D];
    avec a0[D], a1[D];
    // loads of D row pairs starting at kp; addresses past the end of the run are clamped (always valid memory).  GUARD: also zero what lies
    // past the end (the zeroing waits for the load, so the steady-state loop runs unguarded on full sets and only the tail is guarded)
    auto load = [&](f32x4 (&g)[D], avec (&a)[D], int kp, bool guard) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const bool ok = kp + d < kp1;  // scalar
            const size_t r = 2 * (size_t)(ok ? kp + d : (kp1 > 0 ? kp1 - 1 : 0));  // an empty run at the start of the rows (kp1 == 0) clamps to row 0, not to -1
            g[d] = *reinterpret_cast<const f32x4*>(Gb + r * Cout * 4 + gofb);
            a[d] = *reinterpret_cast<const avec*>(Ab + r * Cin * 4 + aofb);
            if (guard && !ok) g[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto fma_set = [&](const f32x4 (&g)[D], const avec (&a)[D]) {
#pragma unroll
        for (int d = 0; d < D; d++)
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int u = 0; u < TCI; u++) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[d][t], a[d][u], acc[t][u], 0, 0, 0);
    };
    // Steady state: full double sets, straight-line body (no conditional MFMAs: the accumulators must stay put in their registers).  The
    // scheduling barriers pin the order "issue the next set's 2 x D loads, then the D x 4 x TCI MFMAs of the current set"; left alone the
    // scheduler sinks every load to just before its first use and the HBM latency is exposed in front of each MFMA group.
    const int nfull = (kp1 - kp0) / (2 * D);
    int kp = kp0;
    BG_STAMP(0);
    load(g0, a0, kp, false);
    for (int it = 0; it < nfull; it++, kp += 2 * D) {
        __builtin_amdgcn_sched_barrier(0);
        load(g1, a1, kp + D, false);
        __builtin_amdgcn_sched_barrier(0);
        fma_set(g0, a0);
        __builtin_amdgcn_sched_barrier(0);
        load(g0, a0, kp + 2 * D, false);
        __builtin_amdgcn_sched_barrier(0);
        fma_set(g1, a1);
    }
    BG_STAMP(1);
    if (kp < kp1) {  // tail of fewer than 2 * D row pairs (batch sizes that do not divide evenly)
        load(g0, a0, kp, true);
        load(g1, a1, kp + D, true);
        fma_set(g0, a0);
        fma_set(g1, a1);
    }
    // Add the ks waves' copies of each tile through LDS in two rounds of two co-tiles (4 waves x 2 x 16 registers x 64 lanes x TCI floats = 128 KB
    // at TCI = 4): every wave writes its half, then the ks waves of a tile split the round's 32 register groups among them, sum the ks copies and
    // store.  The same code for every wave and every ks (no wave-conditional use of the accumulators; ks = 1 passes its own copy through LDS).
    // C layout of a 32 x 32 tile: column (B operand index) = lane & 31, row (A operand index) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
    // Tile (t, u) of a wave holds output rows co = 4 * row + t and columns ci = TCI * col + u: a lane stores TCI consecutive columns.
    float* pt = P + (size_t)slice * Cout * Cin + (size_t)(tco * 128) * Cin + tci * (32 * TCI) + TCI * i;
    const int ipw = 32 / ks;  // register groups per wave and round
    BG_STAMP(2);
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (half) __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                avec v;
#pragma unroll
                for (int u = 0; u < TCI; u++) v[u] = acc[2 * half + tt][u][r];
                *reinterpret_cast<avec*>(&red[wave][((tt * 16 + r) * 64 + lane) * TCI]) = v;
            }
        __syncthreads();
        for (int k = 0; k < ipw; k++) {
            const int it = ksub * ipw + k, tt = it >> 4, r = it & 15, idx = ((tt * 16 + r) * 64 + lane) * TCI;
            avec v = *reinterpret_cast<const avec*>(&red[tl][idx]);
            for (int c = 1; c < ks; c++) v += *reinterpret_cast<const avec*>(&red[tl + tw * c][idx]);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h, t = 2 * half + tt;
            *reinterpret_cast<avec*>(pt + (size_t)(4 * row + t) * Cin) = v;
        }
    }
    BG_STAMP(3);
}

// one layer per launch: workgroups that read the same rows (the tiles of one slice) get block ids 8 apart, i.e. the same XCD and L2 (speed only)
template <int TCI>
__global__ __launch_bounds__(256, 1) void mlp_wgrad_kernel(int M, int Cout, int Cin, const float* __restrict__ G, const float* __restrict__ A,
                                                           float* __restrict__ P, int ntile_ci, int ntiles, int slices) {
    __shared__ __attribute__((aligned(16))) float red[WG_RED_FLOATS / 4 * TCI];
    const int b = blockIdx.x, q = b >> 3, tile = q % ntiles, slice = (q / ntiles) * 8 + (b & 7);
    wgrad_tile<TCI>(red, M, Cout, Cin, G, A, P, ntile_ci, tile, 1, slice, slices);
}

// Several layers in ONE launch (bg_mlp_weight_grad_group): the weight gradients of all hidden layers of both networks, after both backward
// chains, with every workgroup given the same amount of MFMA work -- the caller sizes each layer's slice count in proportion to its cost
// (rows x tile width), so that one launch of ~256 workgroups keeps every CU busy for the same time instead of six launches with six
// prologue / reduction / finish tails.  Up to WG_MAX_PROBLEMS layers; descriptors travel as a kernel argument.
__global__ __launch_bounds__(256, 1) void mlp_wgrad_group_kernel(WgradGroup grp) {
    __shared__ __attribute__((aligned(16))) float red[WG_RED_FLOATS];
    const int b = blockIdx.x;
    int k = 0;
#pragma unroll
    for (int j = 1; j < WG_MAX_PROBLEMS; j++)
        if (j < grp.np && b >= grp.p[j].wg_begin) k = j;
    const WgradProblem& pr = grp.p[k];
    // Workgroups that read the same rows (the tile groups of one slice) get block indices 8 apart: workgroups are dealt round-robin over the 8
    // XCDs, so they share an L2 and the second reader of a row block hits it instead of crossing the fabric again.  Slices go in chunks of 8
    // (the last chunk of a layer may hold fewer): within a chunk of n slices, local index r -> (tile group r / n, slice r % n).
    const int groups = pr.ntiles / pr.tw, local = b - pr.wg_begin;
    const int chunk = local / (8 * groups), r = local - chunk * 8 * groups, left = pr.slices - chunk * 8, n = left < 8 ? left : 8;
    const int tile0 = (r / n) * pr.tw, slice = chunk * 8 + r % n;
    if (pr.tci == 4) wgrad_tile<4>(red, pr.M, pr.Cout, pr.Cin, pr.G, pr.A, pr.P, pr.ntile_ci, tile0, pr.tw, slice, pr.slices);
    else wgrad_tile<2>(red, pr.M, pr.Cout, pr.Cin, pr.G, pr.A, pr.P, pr.ntile_ci, tile0, pr.tw, slice, pr.slices);
}

// dW[co][ci < Cin_real] = sum over slices of P[s][co][ci], slices added in a fixed order; 16 float4 columns x 16 slice groups per workgroup
// (the slice loop is a chain of dependent-address loads: many short chains, not few long ones)
__global__ __launch_bounds__(256) void mlp_wgrad_finish_kernel(int S, int Cin, int Cin_real, int n4, const float* __restrict__ P, float* __restrict__ dW) {
    __shared__ f32x4 sm[16][16];
    const int c = threadIdx.x & 15, sg = threadIdx.x >> 4, e4 = blockIdx.x * 16 + c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (e4 < n4)
        for (int s = sg; s < S; s += 16) acc += *reinterpret_cast<const f32x4*>(P + ((size_t)s * n4 + e4) * 4);
    sm[sg][c] = acc;
    __syncthreads();
    if (sg == 0 && e4 < n4) {
        f32x4 v = sm[0][c];
#pragma unroll
        for (int k = 1; k < 16; k++) v += sm[k][c];
        const int row = (e4 * 4) / Cin, col = (e4 * 4) % Cin;
        if (Cin_real == Cin) {
            *reinterpret_cast<f32x4*>(dW + (size_t)row * Cin + col) = v;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (col + k < Cin_real) dW[(size_t)row * Cin_real + col + k] = v[k];
        }
    }
}

extern "C" int bg_mlp_weight_grad(int32_t M, int32_t C_out, int32_t C_in, int32_t C_in_real, const float* G, const float* A, float* dW,
                                  float* scratch, int32_t slices, void* stream) {
    if (M <= 0 || !G || !A || !dW || !scratch) return bg_set_error(-1, "bg_mlp_weight_grad: bad argument");
    if ((((uintptr_t)G | (uintptr_t)A | (uintptr_t)scratch) & 15) != 0) return bg_set_error(-1, "bg_mlp_weight_grad: G, A, scratch must be 16-byte aligned");
    if (C_in_real == C_in && (((uintptr_t)dW) & 15) != 0) return bg_set_error(-1, "bg_mlp_weight_grad: dW must be 16-byte aligned");
    if (C_out % 128 != 0 || C_out > 1024) return bg_set_error(-4, "bg_mlp_weight_grad: unsupported C_out (multiples of 128 up to 1024)");
    if (C_in != 64 && (C_in % 128 != 0 || C_in > 1024)) return bg_set_error(-4, "bg_mlp_weight_grad: unsupported C_in (64, or multiples of 128 up to 1024)");
    if (C_in_real <= 0 || C_in_real > C_in) return bg_set_error(-1, "bg_mlp_weight_grad: C_in_real must be in [1, C_in]");
    if (M % 2 != 0) return bg_set_error(-4, "bg_mlp_weight_grad: M must be even (rows are consumed in pairs)");
    if (slices <= 0 || slices % 8 != 0 || (long)slices * 8 > M) return bg_set_error(-4, "bg_mlp_weight_grad: slices must be a multiple of 8 with slices * 8 <= M");
    const int tci_w = C_in == 64 ? 64 : 128, ntile_ci = C_in / tci_w, ntiles = (C_out / 128) * ntile_ci;
    hipStream_t st = (hipStream_t)stream;
    if (C_in == 64) hipLaunchKernelGGL((mlp_wgrad_kernel<2>), dim3(ntiles * slices), dim3(256), 0, st, M, C_out, C_in, G, A, scratch, ntile_ci, ntiles, slices);
    else hipLaunchKernelGGL((mlp_wgrad_kernel<4>), dim3(ntiles * slices), dim3(256), 0, st, M, C_out, C_in, G, A, scratch, ntile_ci, ntiles, slices);
    const int n4 = C_out * C_in / 4;
    hipLaunchKernelGGL(mlp_wgrad_finish_kernel, dim3((n4 + 15) / 16), dim3(256), 0, st, slices, C_in, C_in_real, n4, scratch, dW);
    HIP_OK(hipGetLastError());
    return 0;
}


__global__ __launch_bounds__(256) void mlp_wgrad_group_finish_kernel(WgradGroup grp) {
    __shared__ f32x4 sm[16][16];
    const int b = blockIdx.x;
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
        if (Cin_real == Cin) {
            *reinterpret_cast<f32x4*>(pr.dW + (size_t)row * Cin + col) = v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (col + j < Cin_real) pr.dW[(size_t)row * Cin_real + col + j] = v[j];
        }
    }
}

// validation + descriptor build shared with bg_wgrad_split.hip; `who` prefixes the error messages
int bg_wgrad_group_fill(const bg_wgrad_problem* problems, int32_t count, WgradGroup& grp, int& wg, int& fin, const char* who) {
    static thread_local char msg[160];
    auto fail = [&](int code, const char* what) { snprintf(msg, sizeof msg, "%s: %s", who, what); return bg_set_error(code, msg); };
    if (!problems || count <= 0 || count > WG_MAX_PROBLEMS) return fail(-1, "1 to 8 problems");
    grp.np = count;
    wg = 0; fin = 0;
    for (int k = 0; k < count; k++) {
        const bg_wgrad_problem& q = problems[k];
        if (q.M <= 0 || !q.G || !q.A || !q.dW || !q.scratch) return fail(-1, "bad argument");
        if ((((uintptr_t)q.G | (uintptr_t)q.A | (uintptr_t)q.scratch) & 15) != 0) return fail(-1, "G, A, scratch must be 16-byte aligned");
        if (q.C_in_real == q.C_in && (((uintptr_t)q.dW) & 15) != 0) return fail(-1, "dW must be 16-byte aligned");
        if (q.C_out % 128 != 0 || q.C_out > 1024) return fail(-4, "unsupported C_out (multiples of 128 up to 1024)");
        if (q.C_in != 64 && (q.C_in % 128 != 0 || q.C_in > 1024)) return fail(-4, "unsupported C_in (64, or multiples of 128 up to 1024)");
        if (q.C_in_real <= 0 || q.C_in_real > q.C_in) return fail(-1, "C_in_real must be in [1, C_in]");
        if (q.M % 2 != 0) return fail(-4, "M must be even (rows are consumed in pairs)");
        if (q.slices <= 0 || (long)q.slices * 8 > q.M) return fail(-4, "slices must be in [1, M / 8]");
        WgradProblem& p = grp.p[k];
        p.G = q.G; p.A = q.A; p.P = q.scratch; p.dW = q.dW;
        p.M = q.M; p.Cout = q.C_out; p.Cin = q.C_in; p.Cin_real = q.C_in_real; p.tci = q.C_in == 64 ? 2 : 4;
        p.ntile_ci = q.C_in == 64 ? 1 : q.C_in / 128;
        p.ntiles = (q.C_out / 128) * p.ntile_ci;
        p.tw = q.tiles_per_workgroup <= 0 ? 1 : q.tiles_per_workgroup;
        if ((p.tw != 1 && p.tw != 2 && p.tw != 4) || p.ntiles % p.tw != 0)
            return fail(-4, "tiles_per_workgroup must be 1, 2 or 4 and divide the layer's tile count");
        if ((long)q.slices * (4 / p.tw) * 2 > q.M) return fail(-4, "too many slices for M");
        p.slices = q.slices;
        p.wg_begin = wg; wg += (p.ntiles / p.tw) * p.slices;
        p.n4 = q.C_out * q.C_in / 4;
        p.fin_begin = fin; fin += (p.n4 + 15) / 16;
    }
    return 0;
}
int bg_wgrad_group_finish_launch(const WgradGroup& grp, int fin, hipStream_t st) {
    hipLaunchKernelGGL(mlp_wgrad_group_finish_kernel, dim3(fin), dim3(256), 0, st, grp);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// the main kernel only: the fixed-order finish over the slices is left to bg_update_tail (bg_tail.hip), which runs it inside the mini-epoch's last launch
extern "C" int bg_mlp_weight_grad_group_partial(const bg_wgrad_problem* problems, int32_t count, void* stream) {
    WgradGroup grp;
    int wg = 0, fin = 0;
    const int rc = bg_wgrad_group_fill(problems, count, grp, wg, fin, "bg_mlp_weight_grad_group_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(mlp_wgrad_group_kernel, dim3(wg), dim3(256), 0, (hipStream_t)stream, grp);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_mlp_weight_grad_group(const bg_wgrad_problem* problems, int32_t count, void* stream) {
    WgradGroup grp;
    int wg = 0, fin = 0;
    const int rc = bg_wgrad_group_fill(problems, count, grp, wg, fin, "bg_mlp_weight_grad_group");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mlp_wgrad_group_kernel, dim3(wg), dim3(256), 0, st, grp);
    hipLaunchKernelGGL(mlp_wgrad_group_finish_kernel, dim3(fin), dim3(256), 0, st, grp);
    HIP_OK(hipGetLastError());
    return 0;
}
