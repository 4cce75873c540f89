// Probe: the forward chain of one network (three Linear+ELU layers) as ONE kernel per 128-row slab, activations handed from layer to layer in
// REGISTERS.  The products are computed transposed (D = W X^T: the weights are the A operand, read from LDS; the activations are the B operand):
// the accumulator layout of v_mfma_f32_32x32x2_f32 then gives a lane ONE sample and 16 features per tile, feature 32 t + (r & 3) + 8 (r >> 2) + 4 h
// in register r of tile t (h = lane >> 5) -- which is exactly the B operand of k-step 16 t + r of the next layer when that layer walks its k in the
// same permuted order (the weights are staged in LDS, so their order is free).  No transposition, no LDS round trip, no re-read of the activations;
// the HBM store of every layer's activations (the backward pass needs them) is fire-and-forget.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;    // k-chunk staged in LDS
constexpr int LDW = 36;   // LDS row stride (floats)
constexpr int NMAX = 256;

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

// W [N][K] row-major; chunk kc = columns kc*32 .. +31 of all N rows -> sW[n * LDW + kk]
template <int K, int N>
__device__ __forceinline__ void load_w_chunk(f32x4 (&wreg)[N / 32], const float* __restrict__ W, int kc) {
#pragma unroll
    for (int u = 0; u < N / 32; u++) {
        const int idx = threadIdx.x + u * 256, n = idx >> 3, c4 = idx & 7;
        wreg[u] = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + kc * KC + 4 * c4);
    }
}
template <int N>
__device__ __forceinline__ void store_w_chunk(const f32x4 (&wreg)[N / 32], float* sWbuf) {
#pragma unroll
    for (int u = 0; u < N / 32; u++) {
        const int idx = threadIdx.x + u * 256, n = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<f32x4*>(&sWbuf[n * LDW + 4 * c4]) = wreg[u];
    }
}

// One layer for this wave's 32 samples: x[s] = input feature (s & 3) + 8 (s >> 2) + 4 h of the lane's sample (K / 2 registers), out: y[16 t + r] =
// elu(acc + bias) in the same order (N / 2 registers), stored to Y [M][N] row-major as 16-byte pieces.
template <int K, int N>
__device__ __forceinline__ void chain_layer(const float (&x)[K / 2], float (&y)[N / 2], const float* __restrict__ W, const float* __restrict__ bias,
                                            float* __restrict__ Y, int row, bool live, float* sW, int i, int h) {
    constexpr int NT = N / 32, CH = K / KC;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    f32x4 wreg[N / 32];
    load_w_chunk<K, N>(wreg, W, 0);
    __syncthreads();  // the previous layer's last chunk is no longer read
    store_w_chunk<N>(wreg, sW);
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < CH; kc++) {
        float* cur = sW + (kc & 1) * (NMAX * LDW);
        float* nxt = sW + ((kc + 1) & 1) * (NMAX * LDW);
        if (kc + 1 < CH) load_w_chunk<K, N>(wreg, W, kc + 1);
        const float* sw = cur + i * LDW + 4 * h;
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(sw + t * 32 * LDW + j * 8);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, x[16 * kc + 4 * j + 0], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, x[16 * kc + 4 * j + 1], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, x[16 * kc + 4 * j + 2], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, x[16 * kc + 4 * j + 3], acc[t], 0, 0, 0);
            }
        }
        if (kc + 1 < CH) {
            store_w_chunk<N>(wreg, nxt);
            __syncthreads();
        }
    }
    // epilogue: bias + ELU in the accumulator layout, 16-byte stores (features 32 t + 8 g + 4 h .. + 3 of the lane's sample)
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 32 * t + 8 * g + 4 * h);
            f32x4 v;
            v.x = elu_f(acc[t][4 * g + 0] + b4.x); v.y = elu_f(acc[t][4 * g + 1] + b4.y);
            v.z = elu_f(acc[t][4 * g + 2] + b4.z); v.w = elu_f(acc[t][4 * g + 3] + b4.w);
            y[16 * t + 4 * g + 0] = v.x; y[16 * t + 4 * g + 1] = v.y; y[16 * t + 4 * g + 2] = v.z; y[16 * t + 4 * g + 3] = v.w;
            if (live) *reinterpret_cast<f32x4*>(Y + (size_t)row * N + 32 * t + 8 * g + 4 * h) = v;
        }
}

template <int K0, int N1, int N2, int N3>
__global__ __launch_bounds__(256) void mlp_chain_fwd_kernel(int M, const float* __restrict__ X, const float* __restrict__ W1, const float* __restrict__ b1,
                                                            const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ W3,
                                                            const float* __restrict__ b3, float* __restrict__ Y1, float* __restrict__ Y2,
                                                            float* __restrict__ Y3) {
    __shared__ __attribute__((aligned(16))) float sW[2 * NMAX * LDW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * 128 + wave * 32 + i;
    const bool live = row < M;
    const float* xrow = X + (size_t)(live ? row : M - 1) * K0 + 4 * h;
    float x0[K0 / 2];
#pragma unroll
    for (int j = 0; j < K0 / 8; j++) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + 8 * j);
        x0[4 * j + 0] = v.x; x0[4 * j + 1] = v.y; x0[4 * j + 2] = v.z; x0[4 * j + 3] = v.w;
    }
    float y1[N1 / 2], y2[N2 / 2], y3[N3 / 2];
    chain_layer<K0, N1>(x0, y1, W1, b1, Y1, row, live, sW, i, h);
    chain_layer<N1, N2>(y1, y2, W2, b2, Y2, row, live, sW, i, h);
    chain_layer<N2, N3>(y2, y3, W3, b3, Y3, row, live, sW, i, h);
}

extern "C" int bg_mlp_chain_forward(int32_t M, int32_t K0, int32_t N1, int32_t N2, int32_t N3, const float* X, const float* W1, const float* b1,
                                    const float* W2, const float* b2, const float* W3, const float* b3, float* Y1, float* Y2, float* Y3, void* stream) {
    dim3 grid((M + 127) / 128), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (K0 == 64 && N1 == 256 && N2 == 128 && N3 == 128)
        hipLaunchKernelGGL((mlp_chain_fwd_kernel<64, 256, 128, 128>), grid, block, 0, st, M, X, W1, b1, W2, b2, W3, b3, Y1, Y2, Y3);
    else if (K0 == 64 && N1 == 256 && N2 == 256 && N3 == 128)
        hipLaunchKernelGGL((mlp_chain_fwd_kernel<64, 256, 256, 128>), grid, block, 0, st, M, X, W1, b1, W2, b2, W3, b3, Y1, Y2, Y3);
    else
        return -4;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
