// Hand-written fp32-MFMA layer kernels for the PPO update's MLPs (reference utils/model.py:9-26 Linear+ELU stacks, runner.py:132,147,163),
// gfx950 only.  Shapes are tall and skinny (M = 98,304 rows, K and N in {128, 256}), so one workgroup owns a 128-row slab and ALL N columns:
//   * the A operand (activations) is read once from HBM straight into registers (16-byte loads, k permuted identically for A and B so that
//     one load feeds four v_mfma_f32_32x32x2_f32) and reused across the N/32 column tiles;
//   * the B operand (weights, <= 256 kB, L2 resident) is staged per 32-wide k-chunk in LDS and shared by the 4 waves;
//   * bias + ELU are applied to the accumulators before the only store of the output (the library GEMM + elementwise pair writes and
//     re-reads the [M][N] tensor twice more).
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/booster_gym_amd.h"

extern int bg_set_error(int code, const char* msg);
#define HIP_OK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return bg_set_error(-2, hipGetErrorString(_e));               \
    } while (0)

#ifdef BG_PROBE_TIMELINE  // tools/archive/mlp_timeline_probe.py: shader-clock stamps of every wave at the phase boundaries (never defined in the product build)
__device__ long long bg_timeline_buf[2048 * 4 * 16];
#define BG_STAMP(SLOT) do { if ((threadIdx.x & 63) == 0) bg_timeline_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (SLOT)] = clock64(); } while (0)
#define BG_EPI_STAMP(SLOT) BG_STAMP(SLOT)
extern "C" int bg_probe_read_timeline(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(bg_timeline_buf), bytes); }
#else
#define BG_STAMP(SLOT) do { } while (0)
#endif
#include "bg_mlp_tile.h"

// one k-chunk (32 k-values) of MFMAs for this wave: 4 sub-steps x NT column tiles x 4 MFMAs
template <int NT>
__device__ __forceinline__ void mfma_chunk(f32x16 (&acc)[NT], const f32x4 (&a4)[4], const float* sw /* &sW[buf][i * FW_LDW + 4 * h] */) {
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sw + t * 32 * FW_LDW + s * 8);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].x, b4.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].y, b4.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].z, b4.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].w, b4.w, acc[t], 0, 0, 0);
        }
    }
}
template <int K, int LD4>
__device__ __forceinline__ void load_w_chunk(f32x4 (&wreg)[LD4], const float* __restrict__ W, int kc) {
#pragma unroll
    for (int u = 0; u < LD4; u++) {
        const int idx = threadIdx.x + u * 256, n = idx >> 3, c4 = idx & 7;
        wreg[u] = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + kc * FW_KC + 4 * c4);
    }
}
template <int LD4>
__device__ __forceinline__ void store_w_chunk(const f32x4 (&wreg)[LD4], float* sWbuf) {
#pragma unroll
    for (int u = 0; u < LD4; u++) {
        const int idx = threadIdx.x + u * 256, n = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<f32x4*>(&sWbuf[n * FW_LDW + 4 * c4]) = wreg[u];
    }
}
// Software pipeline, two k-chunks per loop trip with two explicit register sets (no copies, no scratch): the global loads of chunk c+1 are
// issued before the MFMAs of chunk c and first waited for after them, so HBM latency hides under 64 MFMAs (4096 cycles) of this wave alone.
// Each workgroup computes 128 rows x 128 columns (blockIdx.y = column block): with N = 256 the two column blocks of a row slab run
// concurrently and the second read of the X rows is served by L2 / Infinity Cache.
// EPI 0: Y = X W^T + bias          EPI 1: Y = elu(X W^T + bias)                                   (forward)
// EPI 2: Y = (X W^T) * elu'(aux), aux = the layer's OUTPUT activations (1 if aux > 0 else aux + 1), plus per-workgroup column sums of Y
//        written to colpart[blockIdx.x][ldy]  (backward: X = dL/dz of the layer above, W = its transposed weight, Y = dL/dz of this layer,
//        column sums = this layer's bias gradient)
// NB = number of 128-column blocks of the layer (1, 2; 0 = any): it only gives every layer shape its own kernel symbol, so that a profiler's
// per-kernel average is the average of ONE shape (bench.py's roofline line is checked against the rocprofv3 summary in profiles/).
template <int K, int EPI, int NB>
__global__ __launch_bounds__(256, 3) void mlp_fwd_kernel(int M, int ldy, const float* __restrict__ X, const float* __restrict__ Wfull,
                                                         const float* __restrict__ biasfull, float* __restrict__ Yfull,
                                                         const float* __restrict__ auxfull, float* __restrict__ colpart) {
    constexpr int N = 128;
    constexpr int NT = N / 32, CH = K / FW_KC, LD4 = N * (FW_KC / 4) / 256;
    // 1-D grid, XCD-aware: workgroups are dealt round-robin over the 8 XCDs, so ids that differ by 8 share an L2.  The column blocks of one row
    // slab get ids 8 apart (same XCD, dispatched together): the second read of the slab's X rows is an L2 hit instead of a second trip over
    // the fabric (PMC, profiles/r02_bench_pmc.json: FETCH_SIZE was 2 x |X| with by = column block).  Groups of 8 slabs x ncb column blocks.
    const int ncb = ldy / N, grp = blockIdx.x / (8 * ncb), rem = blockIdx.x % (8 * ncb);
    const int bx = grp * 8 + (rem & 7), by = rem >> 3;  // row slab, column block
    if (bx * FW_BM >= M) return;
    const float* __restrict__ W = Wfull + (size_t)by * N * K;
    const float* __restrict__ bias = EPI <= 1 ? biasfull + by * N : nullptr;
    float* __restrict__ Y = Yfull + by * N;
    static_assert(CH % 2 == 0, "K must be a multiple of 64");
    __shared__ __attribute__((aligned(16))) float sW[2][N * FW_LDW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int row = bx * FW_BM + wave * 32 + i;
    const float* xrow = X + (size_t)(row < M ? row : M - 1) * K + 4 * h;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    f32x4 wA[LD4], wB[LD4], aA[4], aB[4];
    f32x4 auxq[EPI == 2 ? NT : 1][4];  // elu' operand of the backward epilogue in the TRANSPOSED (row, 4 columns) layout of the stores
    BG_STAMP(0);
#ifdef BG_PROBE_TIMELINE  // slot 15: where the wave runs (HW_ID: cu [11:8], sh [12], se [15:13]; XCC_ID [3:0]), slot 11: launch-wide realtime clock
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        bg_timeline_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + 15] = (long long)hw | ((long long)xcc << 32);
        bg_timeline_buf[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + 11] = wall_clock64();
    }
#endif
    load_w_chunk<K, LD4>(wA, W, 0);
    load_a_chunk(aA, xrow, 0);
    store_w_chunk<LD4>(wA, sW[0]);
    __syncthreads();
    BG_STAMP(1);
    const float* sw0 = &sW[0][i * FW_LDW + 4 * h];
    const float* sw1 = &sW[1][i * FW_LDW + 4 * h];
    for (int kc = 0; kc < CH; kc += 2) {
        // chunk kc on (aA, sW[0]); prefetch kc+1 into (aB, wB)
        load_w_chunk<K, LD4>(wB, W, kc + 1);
        load_a_chunk(aB, xrow, kc + 1);
        mfma_chunk<NT>(acc, aA, sw0);
        store_w_chunk<LD4>(wB, sW[1]);
        __syncthreads();
        // chunk kc+1 on (aB, sW[1]); prefetch kc+2 into (aA, wA)
        if (kc + 2 < CH) {
            load_w_chunk<K, LD4>(wA, W, kc + 2);
            load_a_chunk(aA, xrow, kc + 2);
        } else if constexpr (EPI == 2) {
            // last chunk: fetch the epilogue's elu' operand now, so that its HBM latency hides under the final 64 MFMAs
            const float* __restrict__ auxp = auxfull + by * N;
            const int rb = bx * FW_BM + wave * 32;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int g = 0; g < 4; g++) {  // the epilogue's row layout: row (lane >> 3) + 8 g, columns 4 (lane & 7) ..
                    const int rr = rb + 8 * g + (lane >> 3);
                    auxq[t][g] = *reinterpret_cast<const f32x4*>(auxp + (size_t)(rr < M ? rr : M - 1) * ldy + t * 32 + 4 * (lane & 7));
                }
        }
        mfma_chunk<NT>(acc, aB, sw1);
        if (kc + 2 < CH) store_w_chunk<LD4>(wA, sW[0]);
        BG_STAMP(2 + kc / 2);  // own work on a pair of chunks, before the barrier
        __syncthreads();
    }
    BG_STAMP(2 + CH / 2);
    layer_epilogue<EPI, NT>(acc, auxq, M, ldy, bx, by, wave, lane, i, h, bias, Y, colpart, &sW[0][0]);  // csum reuses the weight staging buffer
    BG_STAMP(3 + CH / 2);
}

__global__ __launch_bounds__(256) void mlp_colsum_finish_kernel(int nb, int C, const float* __restrict__ partial, float* __restrict__ out) {
    __shared__ float sm[256];
    const int c = blockIdx.x;
    float acc = 0.f;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) acc += partial[(size_t)b * C + c];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[c] = sm[0];
}

extern "C" int bg_mlp_layer_forward(int32_t M, int32_t K, int32_t N, const float* X, const float* W, const float* bias, float* Y, int32_t elu,
                                    void* stream) {
    if (M <= 0 || !X || !W || !bias || !Y) return bg_set_error(-1, "bg_mlp_layer_forward: bad argument");
    if ((((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y) & 15) != 0) return bg_set_error(-1, "bg_mlp_layer_forward: pointers must be 16-byte aligned");
    if (N % 128 != 0 || N > 1024) return bg_set_error(-4, "bg_mlp_layer_forward: unsupported N (multiples of 128 up to 1024)");
    dim3 grid((((M + FW_BM - 1) / FW_BM + 7) / 8) * 8 * (N / 128)), block(256);  // slabs padded to groups of 8 (mlp_fwd_kernel's XCD-aware mapping)
    hipStream_t st = (hipStream_t)stream;
#define BG_FWD(KK)                                                                                                      \
    if (K == KK) {                                                                                                      \
        if (elu && N == 128) hipLaunchKernelGGL((mlp_fwd_kernel<KK, 1, 1>), grid, block, 0, st, M, N, X, W, bias, Y, nullptr, nullptr);      \
        else if (elu && N == 256) hipLaunchKernelGGL((mlp_fwd_kernel<KK, 1, 2>), grid, block, 0, st, M, N, X, W, bias, Y, nullptr, nullptr); \
        else if (elu) hipLaunchKernelGGL((mlp_fwd_kernel<KK, 1, 0>), grid, block, 0, st, M, N, X, W, bias, Y, nullptr, nullptr);             \
        else hipLaunchKernelGGL((mlp_fwd_kernel<KK, 0, 0>), grid, block, 0, st, M, N, X, W, bias, Y, nullptr, nullptr);                      \
        HIP_OK(hipGetLastError());                                                                                      \
        return 0;                                                                                                       \
    }
    BG_FWD(256)
    BG_FWD(128)
    BG_FWD(64)
#undef BG_FWD
    return bg_set_error(-4, "bg_mlp_layer_forward: unsupported K (64, 128, 256)");
}

extern "C" int bg_mlp_layer_backward(int32_t M, int32_t K, int32_t N, const float* G, const float* Wt, const float* act_below, float* Gout,
                                     float* bias_grad_below, float* scratch, void* stream) {
    if (M <= 0 || !G || !Wt || !act_below || !Gout || !bias_grad_below || !scratch) return bg_set_error(-1, "bg_mlp_layer_backward: bad argument");
    if ((((uintptr_t)G | (uintptr_t)Wt | (uintptr_t)Gout | (uintptr_t)act_below) & 15) != 0)
        return bg_set_error(-1, "bg_mlp_layer_backward: pointers must be 16-byte aligned");
    if (N % 128 != 0 || N > 1024) return bg_set_error(-4, "bg_mlp_layer_backward: unsupported N (multiples of 128 up to 1024)");
    const int nb = (M + FW_BM - 1) / FW_BM;
    dim3 grid(((nb + 7) / 8) * 8 * (N / 128)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define BG_BWD(KK)                                                                                                                \
    if (K == KK) {                                                                                                                \
        if (N == 128) hipLaunchKernelGGL((mlp_fwd_kernel<KK, 2, 1>), grid, block, 0, st, M, N, G, Wt, nullptr, Gout, act_below, scratch);      \
        else if (N == 256) hipLaunchKernelGGL((mlp_fwd_kernel<KK, 2, 2>), grid, block, 0, st, M, N, G, Wt, nullptr, Gout, act_below, scratch); \
        else hipLaunchKernelGGL((mlp_fwd_kernel<KK, 2, 0>), grid, block, 0, st, M, N, G, Wt, nullptr, Gout, act_below, scratch);               \
        hipLaunchKernelGGL(mlp_colsum_finish_kernel, dim3(N), dim3(256), 0, st, nb, N, scratch, bias_grad_below);                  \
        HIP_OK(hipGetLastError());                                                                                                \
        return 0;                                                                                                                 \
    }
    BG_BWD(256)
    BG_BWD(128)
#undef BG_BWD
    return bg_set_error(-4, "bg_mlp_layer_backward: unsupported K (128, 256)");
}

// bg_mlp_layer_backward without the column-sum finish: the descriptor of that reduction instead (bg_reduce_group runs it later)
extern "C" int bg_mlp_layer_backward_partial(int32_t M, int32_t K, int32_t N, const float* G, const float* Wt, const float* act_below, float* Gout,
                                             float* bias_grad_below, float* scratch, bg_reduce_problem* finish, void* stream) {
    if (M <= 0 || !G || !Wt || !act_below || !Gout || !bias_grad_below || !scratch || !finish) return bg_set_error(-1, "bg_mlp_layer_backward_partial: bad argument");
    if ((((uintptr_t)G | (uintptr_t)Wt | (uintptr_t)Gout | (uintptr_t)act_below) & 15) != 0)
        return bg_set_error(-1, "bg_mlp_layer_backward_partial: pointers must be 16-byte aligned");
    if (N % 128 != 0 || N > 1024) return bg_set_error(-4, "bg_mlp_layer_backward_partial: unsupported N (multiples of 128 up to 1024)");
    if (K != 256 && K != 128) return bg_set_error(-4, "bg_mlp_layer_backward_partial: unsupported K (128, 256)");
    const int nb = (M + FW_BM - 1) / FW_BM;
    dim3 grid(((nb + 7) / 8) * 8 * (N / 128)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define BG_BWDP(KK)                                                                                                               \
    if (K == KK) {                                                                                                                \
        if (N == 128) hipLaunchKernelGGL((mlp_fwd_kernel<KK, 2, 1>), grid, block, 0, st, M, N, G, Wt, nullptr, Gout, act_below, scratch);      \
        else if (N == 256) hipLaunchKernelGGL((mlp_fwd_kernel<KK, 2, 2>), grid, block, 0, st, M, N, G, Wt, nullptr, Gout, act_below, scratch); \
        else hipLaunchKernelGGL((mlp_fwd_kernel<KK, 2, 0>), grid, block, 0, st, M, N, G, Wt, nullptr, Gout, act_below, scratch);               \
    }
    BG_BWDP(256)
    BG_BWDP(128)
#undef BG_BWDP
    HIP_OK(hipGetLastError());
    memset(finish, 0, sizeof(*finish));
    finish->partial = scratch; finish->groups = nb; finish->record = N; finish->n_out = N;
    finish->out[0] = bias_grad_below; finish->n[0] = N;
    return 0;
}

// launch of the fixed-order column-sum finish for bg_mlp_split.hip's backward layer
int bg_colsum_finish_launch(int nb, int C, const float* partial, float* out, hipStream_t st) {
    hipLaunchKernelGGL(mlp_colsum_finish_kernel, dim3(C), dim3(256), 0, st, nb, C, partial, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
