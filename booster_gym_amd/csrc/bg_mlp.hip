// Hand-written fp32-MFMA layer kernels for the PPO update's MLPs (reference utils/model.py:9-26 Linear+ELU stacks, runner.py:132,147,163),
// gfx950 only.  Shapes are tall and skinny (M = 98,304 rows, K and N in {128, 256}), so one workgroup owns a 128-row slab and ALL N columns:
//   * the A operand (activations) is read once from HBM straight into registers (16-byte loads, k permuted identically for A and B so that
//     one load feeds four v_mfma_f32_32x32x2_f32) and reused across the N/32 column tiles;
//   * the B operand (weights, <= 256 kB, L2 resident) is staged per 32-wide k-chunk in LDS and shared by the 4 waves;
//   * bias + ELU are applied to the accumulators before the only store of the output (the library GEMM + elementwise pair writes and
//     re-reads the [M][N] tensor twice more).
#include <hip/hip_runtime.h>

#include "../../include/booster_gym_amd.h"

extern int bg_set_error(int code, const char* msg);
#define HIP_OK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return bg_set_error(-2, hipGetErrorString(_e));               \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// exp(x) - 1 through v_exp_f32: absolute error ~1e-7 on (-1, 0], far below fp32 activation noise; expm1f costs ~20 VALU per element
__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

constexpr int FW_BM = 128;   // rows per workgroup (4 waves x 32 rows)
constexpr int FW_KC = 32;    // k-chunk staged in LDS
constexpr int FW_LDW = 36;   // LDS row stride (floats): 16-byte aligned rows, spreads the 16-byte reads over the banks

template <int K, int N, bool ACT>
__global__ __launch_bounds__(256, (N <= 128 ? 2 : 1)) void mlp_fwd_kernel(int M, const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
                                                      float* __restrict__ Y) {
    constexpr int NT = N / 32;
    __shared__ __attribute__((aligned(16))) float sW[2][N * FW_LDW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * FW_BM + wave * 32 + i;
    const float* xrow = X + (size_t)(row < M ? row : M - 1) * K + 4 * h;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;

    // stage chunk 0 of W: N rows x 32 floats = N * 8 float4, 256 threads
    constexpr int LD4 = N * (FW_KC / 4) / 256;  // float4 per thread per chunk
    float4 wreg[LD4];
    auto load_w = [&](int kc) {
#pragma unroll
        for (int u = 0; u < LD4; u++) {
            const int idx = threadIdx.x + u * 256, n = idx >> 3, c4 = idx & 7;
            wreg[u] = *reinterpret_cast<const float4*>(W + (size_t)n * K + kc * FW_KC + 4 * c4);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int u = 0; u < LD4; u++) {
            const int idx = threadIdx.x + u * 256, n = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<float4*>(&sW[buf][n * FW_LDW + 4 * c4]) = wreg[u];
        }
    };
    load_w(0);
    store_w(0);
    float4 a4[4];
#pragma unroll
    for (int s = 0; s < 4; s++) a4[s] = *reinterpret_cast<const float4*>(xrow + s * 8);
    __syncthreads();
    constexpr int CH = K / FW_KC;
    for (int kc = 0; kc < CH; kc++) {
        const int buf = kc & 1;
        float4 a_next[4];
        if (kc + 1 < CH) {
            load_w(kc + 1);
#pragma unroll
            for (int s = 0; s < 4; s++) a_next[s] = *reinterpret_cast<const float4*>(xrow + (kc + 1) * FW_KC + s * 8);
        }
        const float* sw = &sW[buf][i * FW_LDW + 4 * h];
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const float4 b4 = *reinterpret_cast<const float4*>(sw + t * 32 * FW_LDW + s * 8);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].x, b4.x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].y, b4.y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].z, b4.z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].w, b4.w, acc[t], 0, 0, 0);
            }
        }
        if (kc + 1 < CH) {
            store_w(buf ^ 1);
#pragma unroll
            for (int s = 0; s < 4; s++) a4[s] = a_next[s];
        }
        __syncthreads();
    }
    // epilogue: C layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int rbase = blockIdx.x * FW_BM + wave * 32;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float bv = bias[t * 32 + i];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int rr = rbase + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[t][r] + bv;
            if (ACT) v = elu_f(v);
            if (rr < M) Y[(size_t)rr * N + t * 32 + i] = v;
        }
    }
}

extern "C" int bg_mlp_layer_forward(int32_t M, int32_t K, int32_t N, const float* X, const float* W, const float* bias, float* Y, int32_t elu,
                                    void* stream) {
    if (M <= 0 || !X || !W || !bias || !Y) return bg_set_error(-1, "bg_mlp_layer_forward: bad argument");
    if ((((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y) & 15) != 0) return bg_set_error(-1, "bg_mlp_layer_forward: pointers must be 16-byte aligned");
    dim3 grid((M + FW_BM - 1) / FW_BM), block(256);
    hipStream_t st = (hipStream_t)stream;
#define BG_FWD(KK, NN)                                                                                                         \
    if (K == KK && N == NN) {                                                                                                  \
        if (elu) hipLaunchKernelGGL((mlp_fwd_kernel<KK, NN, true>), grid, block, 0, st, M, X, W, bias, Y);                     \
        else hipLaunchKernelGGL((mlp_fwd_kernel<KK, NN, false>), grid, block, 0, st, M, X, W, bias, Y);                        \
        HIP_OK(hipGetLastError());                                                                                             \
        return 0;                                                                                                              \
    }
    BG_FWD(256, 256)
    BG_FWD(256, 128)
    BG_FWD(128, 128)
#undef BG_FWD
    return bg_set_error(-4, "bg_mlp_layer_forward: unsupported (K, N); supported: (256,256) (256,128) (128,128)");
}
