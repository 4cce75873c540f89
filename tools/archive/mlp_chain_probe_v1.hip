// Probe: the forward chain of one network (three Linear+ELU layers) as ONE kernel per 128-row slab, activations handed from layer to layer in
// REGISTERS.  The products are computed transposed (D = W X^T: the weights are the A operand, read from LDS; the activations are the B operand):
// the accumulator layout of v_mfma_f32_32x32x2_f32 then gives a lane ONE sample and 16 features per tile, feature 32 t + (r & 3) + 8 (r >> 2) + 4 h
// in register r of tile t (h = lane >> 5) -- which is exactly the B operand of k-step 16 t + r of the next layer when that layer walks its k in the
// same permuted order (the weights are staged in LDS, so their order is free).  No transposition, no LDS round trip, no re-read of the activations;
// the HBM store of every layer's activations (the backward pass needs them) is fire-and-forget.
//
// One wave per SIMD (x: K/2 registers, accumulators: N/2), so nothing but the wave's own instruction stream hides latency:
//   * the weights of all three layers are ONE stream of 32-wide k-chunks through three LDS buffers: chunk c + 2 is written and chunk c + 3 fetched
//     while chunk c is multiplied, across layer boundaries too; one barrier per chunk, at the top, and the first operands of a chunk are read
//     BEFORE it (they were made visible by the previous barrier);
//   * the last chunk of a layer runs tile by tile, and bias + ELU + store of tile t are issued under the MFMAs of tile t + 1.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;    // k-chunk staged in LDS (one row of a chunk = 128 bytes = 8 units of 16 bytes)
constexpr int NMAX = 256;
constexpr int BUF = NMAX * KC;  // floats per LDS buffer
constexpr int NBUF = 4, AHEAD = 3;

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0, "");
    constexpr int n = N > 63 ? 63 : N;
    __builtin_amdgcn_s_waitcnt((n & 15) | ((n >> 4) << 14) | 0x0F70);
}

// One k-chunk (32 columns) of W [N][K] global -> LDS with no register stop (global_load_lds_dwordx4: the LDS side of one wave-instruction is 64
// consecutive 16-byte units = 8 rows of the chunk, the global side is per lane).  Unit u of row n is kept at unit u ^ ((n >> 1) & 7) of its row: the
// 16 lanes of one pass of the operand reads (rows 128 bytes apart) then fall into 16 different 16-byte bank groups.  N / 32 instructions per wave.
template <int K, int N>
__device__ __forceinline__ void dma_rows(const float* __restrict__ W, int kc, float* sWbuf, int wave, const unsigned (&lane_ofs)[1]) {
#pragma unroll
    for (int u = 0; u < N / 32; u++) {
        const int q = u * 4 + wave;  // wave-uniform: rows 8 q .. 8 q + 7
        // inline asm, not __builtin_amdgcn_global_load_lds: the compiler orders LDS reads behind a DMA it knows about with vmcnt(0) (it cannot tell
        // the buffers apart) and drains vmcnt at every workgroup fence; the bookkeeping of these copies is explicit here (wait_vm / behind()).
        // Scalar base + 32-bit lane offset: no 64-bit address registers per copy.
        const float* base = W + (size_t)q * 8 * K + kc * KC;
        const unsigned lds = (unsigned)(uintptr_t)(sWbuf + q * 256);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_ofs[0]), "s"(base), "s"(lds) : "memory");
    }
}
// byte offset of this lane's 16 bytes inside an 8-row group of a chunk of a [.][K] matrix: row (lane >> 3), unit (lane & 7) ^ ((n >> 1) & 7) with
// n = 8 q + (lane >> 3): (n >> 1) & 7 = (4 (q & 1) + (lane >> 4)) & 7 (the parity of q = 4 u + wave is the wave's)
template <int K>
__device__ __forceinline__ void dma_lane_offsets(unsigned (&ofs)[1], int lane, int wave) {  // q = 4 u + wave: its parity is the wave's
    ofs[0] = (unsigned)(((lane >> 3) * K + 4 * ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7))) * 4);
}

// Y1 / Y2 / Y3 hold ceil(M / 128) * 128 rows (the stores of a slab are unconditional: their number is part of the vmcnt bookkeeping below).
template <int K0, int N1, int N2, int N3>
__global__ __launch_bounds__(256) void mlp_chain_fwd_kernel(int M, const float* __restrict__ X, const float* __restrict__ W1, const float* __restrict__ b1,
                                                            const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ W3,
                                                            const float* __restrict__ b3, float* __restrict__ Y1, float* __restrict__ Y2,
                                                            float* __restrict__ Y3) {
    constexpr int C0 = K0 / KC, C1 = N1 / KC, C2 = N2 / KC, C = C0 + C1 + C2;
    __shared__ __attribute__((aligned(16))) float sW[NBUF * BUF];
    __shared__ __attribute__((aligned(16))) float sB[N1 + N2 + N3];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    unsigned lo1[1], lo2[1], lo3[1];
    dma_lane_offsets<K0>(lo1, lane, wave);
    dma_lane_offsets<N1>(lo2, lane, wave);
    dma_lane_offsets<N2>(lo3, lane, wave);
    // chunk cc of the stream: DMA instructions per wave; vector-memory operations of the lane issued in iteration cc besides the DMA: the activation
    // stores of a layer's last chunk, and the loads of the NEXT slab's input rows in the first chunk of the last layer
    struct S {
        static constexpr int ndma(int cc) { return (cc % C < C0 ? N1 : cc % C < C0 + C1 ? N2 : N3) / 32; }
        static constexpr int extra(int cc) { return (cc == C0 - 1 ? N1 / 8 : cc == C0 + C1 - 1 ? N2 / 8 : cc == C - 1 ? N3 / 8 : 0) + (cc == C0 + C1 ? K0 / 8 : 0); }
        // operations issued behind the DMA of chunk cc when iteration cc begins: the DMAs of chunks cc + 1 .. cc + AHEAD - 1 and the extras of the
        // iterations since its issue (top of iteration cc - AHEAD; for the first chunks of a slab that was in the previous slab, or in the prologue:
        // counting only this slab's iterations is right for the first slab and asks for a little more than necessary in the others)
        static constexpr int behind(int cc) {
            int n = 0;
            for (int k = cc + 1; k < cc + AHEAD; k++) n += ndma(k);
            for (int it = (cc - AHEAD > 0 ? cc - AHEAD : 0); it < cc; it++) n += extra(it);
            return n;
        }
    };
    int rot = 0;  // chunk c of this slab lives in buffer (c + rot) % NBUF
    auto dma = [&](int cc) {  // cc may run into the next slab (same weights)
        float* dst = sW + ((cc + rot) % NBUF) * BUF;
        const int k = cc % C;
        if (k < C0) dma_rows<K0, N1>(W1, k, dst, wave, lo1);
        else if (k < C0 + C1) dma_rows<N1, N2>(W2, k - C0, dst, wave, lo2);
        else dma_rows<N2, N3>(W3, k - C0 - C1, dst, wave, lo3);
    };
    for (int j = threadIdx.x; j < N1 + N2 + N3; j += 256) sB[j] = j < N1 ? b1[j] : j < N1 + N2 ? b2[j - N1] : b3[j - N1 - N2];
    const int nslabs = (M + 127) / 128;
    float x0[K0 / 2], x0n[K0 / 2];
    auto load_x = [&](float (&x)[K0 / 2], int slab) {
        const int r = slab * 128 + wave * 32 + i;
        const float* xrow = X + (size_t)(r < M ? r : M - 1) * K0 + 4 * h;
#pragma unroll
        for (int j = 0; j < K0 / 8; j++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + 8 * j);
            x[4 * j + 0] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
        }
    };
    load_x(x0, blockIdx.x);
    dma(0);
    dma(1);
    dma(2);
    __builtin_amdgcn_sched_barrier(0);
    const int sx = (i >> 1) & 7;  // this lane's unit swizzle (rows 32 t + i: the tile offset does not change it)
    const int lofs = i * KC;

    for (int slab = blockIdx.x; slab < nslabs; slab += gridDim.x) {
        const int row = slab * 128 + wave * 32 + i;
        f32x16 a1[N1 / 32], a2[N2 / 32], a3[N3 / 32];
        // one layer: chunks base .. base + CH - 1 of the stream; xin(s) = B operand of k-step s
        auto layer = [&](auto& acc, auto xin, auto K_, auto N_, auto base_, int bias_ofs, float* __restrict__ Y) {
            constexpr int K = decltype(K_)::value, N = decltype(N_)::value, NT = N / 32, CH = K / KC, base = decltype(base_)::value;
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
            static_for<CH>([&](auto kc_) {
                constexpr int kc = decltype(kc_)::value, c = base + kc;
                const float* sw = sW + ((c + rot) % NBUF) * BUF + lofs;
                // chunk c complete in LDS (this wave's part), then published by the barrier; everything issued behind its DMA may stay in flight
                __builtin_amdgcn_sched_barrier(0);
                wait_vm<S::behind(c)>();
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): (first iteration) the bias values written to sB above
                asm volatile("s_barrier" ::: "memory");  // no fence: a workgroup fence would drain vmcnt (stores and younger copies included)
                __builtin_amdgcn_sched_barrier(0);
                dma(c + AHEAD);
                __builtin_amdgcn_sched_barrier(0);
                if (c == C0 + C1) load_x(x0n, slab + gridDim.x);  // the next slab's rows (clamped: loaded and ignored behind the last slab)
                if (kc + 1 < CH) {
                    // consecutive MFMAs go to different accumulators (one wave per SIMD: nobody else fills the gap behind a dependent MFMA)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        f32x4 w4[NT];
#pragma unroll
                        for (int t = 0; t < NT; t++) w4[t] = *reinterpret_cast<const f32x4*>(sw + t * 32 * KC + (((2 * j + h) ^ sx) << 2));
#pragma unroll
                        for (int q = 0; q < 4; q++)
#pragma unroll
                            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[t][q], xin(16 * kc + 4 * j + q), acc[t], 0, 0, 0);
                    }
                } else {
                    // last chunk: two tiles at a time; bias + ELU + store of a finished pair go out under the next pair's MFMAs (the accumulators are
                    // converted in place: they are the next layer's B operands)
#pragma unroll
                    for (int t = 0; t < NT; t += 2) {
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const f32x4 wa = *reinterpret_cast<const f32x4*>(sw + t * 32 * KC + (((2 * j + h) ^ sx) << 2));
                            const f32x4 wb = *reinterpret_cast<const f32x4*>(sw + (t + 1) * 32 * KC + (((2 * j + h) ^ sx) << 2));
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[q], xin(16 * kc + 4 * j + q), acc[t], 0, 0, 0);
                                acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[q], xin(16 * kc + 4 * j + q), acc[t + 1], 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int tt = t; tt < t + 2; tt++)
#pragma unroll
                            for (int g = 0; g < 4; g++) {
                                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sB[bias_ofs + 32 * tt + 8 * g + 4 * h]);
                                f32x4 v;
                                v.x = elu_f(acc[tt][4 * g + 0] + b4.x); v.y = elu_f(acc[tt][4 * g + 1] + b4.y);
                                v.z = elu_f(acc[tt][4 * g + 2] + b4.z); v.w = elu_f(acc[tt][4 * g + 3] + b4.w);
                                acc[tt][4 * g + 0] = v.x; acc[tt][4 * g + 1] = v.y; acc[tt][4 * g + 2] = v.z; acc[tt][4 * g + 3] = v.w;
                                *reinterpret_cast<f32x4*>(Y + (size_t)row * N + 32 * tt + 8 * g + 4 * h) = v;
                            }
                    }
                }
            });
        };
        layer(a1, [&](int s) { return x0[s]; }, std::integral_constant<int, K0>{}, std::integral_constant<int, N1>{}, std::integral_constant<int, 0>{}, 0, Y1);
        layer(a2, [&](int s) { return a1[s >> 4][s & 15]; }, std::integral_constant<int, N1>{}, std::integral_constant<int, N2>{}, std::integral_constant<int, C0>{}, N1, Y2);
        layer(a3, [&](int s) { return a2[s >> 4][s & 15]; }, std::integral_constant<int, N2>{}, std::integral_constant<int, N3>{}, std::integral_constant<int, C0 + C1>{}, N1 + N2, Y3);
        rot = (rot + C) % NBUF;
#pragma unroll
        for (int j = 0; j < K0 / 2; j++) x0[j] = x0n[j];
    }
    wait_vm<0>();  // the copies issued for a slab that does not exist must have landed before the LDS is handed to another workgroup
}

extern "C" int bg_mlp_chain_forward(int32_t M, int32_t K0, int32_t N1, int32_t N2, int32_t N3, const float* X, const float* W1, const float* b1,
                                    const float* W2, const float* b2, const float* W3, const float* b3, float* Y1, float* Y2, float* Y3, int32_t wg, void* stream) {
    const int nslabs = (M + 127) / 128;
    dim3 grid(nslabs < wg || wg <= 0 ? nslabs : wg), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (K0 == 64 && N1 == 256 && N2 == 128 && N3 == 128)
        hipLaunchKernelGGL((mlp_chain_fwd_kernel<64, 256, 128, 128>), grid, block, 0, st, M, X, W1, b1, W2, b2, W3, b3, Y1, Y2, Y3);
    else if (K0 == 64 && N1 == 256 && N2 == 256 && N3 == 128)
        hipLaunchKernelGGL((mlp_chain_fwd_kernel<64, 256, 256, 128>), grid, block, 0, st, M, X, W1, b1, W2, b2, W3, b3, Y1, Y2, Y3);
    else
        return -4;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
