// Pieces shared by the layer kernels of bg_mlp.hip (fp32 MFMA) and bg_mlp_split.hip (split bf16 MFMA): tile constants, the A-operand load and
// the epilogues.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in registers (HIP's float4 struct blocked SROA here)

// exp(x) - 1 through v_exp_f32: absolute error ~1e-7 on (-1, 0], far below fp32 activation noise; expm1f costs ~20 VALU per element
__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

constexpr int FW_BM = 128;   // rows per workgroup (4 waves x 32 rows)
constexpr int FW_KC = 32;    // k-chunk staged in LDS
constexpr int FW_LDW = 36;   // LDS row stride (floats): 16-byte aligned rows, spreads the 16-byte reads over the banks

__device__ __forceinline__ void load_a_chunk(f32x4 (&a4)[4], const float* xrow, int kc) {
#pragma unroll
    for (int s = 0; s < 4; s++) a4[s] = *reinterpret_cast<const f32x4*>(xrow + kc * FW_KC + s * 8);
}

#ifndef BG_EPI_STAMP  // timeline probe builds define it
#define BG_EPI_STAMP(SLOT) do { } while (0)
#endif
// Epilogue shared by the fp32-MFMA kernel and the split-bf16 kernel below (both leave the 32 x 32 tiles in the same C layout).
// csum: LDS the caller no longer needs: 4 x 32 x 36 floats (18 KB) for the transposition + 4 x 128 floats for the column sums of EPI 2.
template <int EPI, int NT>
__device__ __forceinline__ void layer_epilogue(f32x16 (&acc)[NT], const f32x4 (&auxq)[EPI == 2 ? NT : 1][4], int M, int ldy, int bx, int by, int wave,
                                               int lane, int i, int h, const float* __restrict__ bias, float* __restrict__ Y,
                                               float* __restrict__ colpart, float* csum) {
    constexpr int N = 128;
    // epilogue: C layout of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int rbase = bx * FW_BM + wave * 32;
    if constexpr (EPI <= 1) {
        // The C layout gives a lane ONE column and 4 consecutive rows per register group; the stores want one row and 4 consecutive columns per lane
        // (16 bytes per lane, 8 full 128-byte lines per instruction: with 64 dword stores per lane the store tail was issue-bound).  The tile goes
        // through a wave-private 32 x 32 block of LDS (the weight staging buffer, free after the loop's last barrier): 16 ds_write_b32 in the C
        // layout, 4 ds_read_b128 by rows -- 20 LDS instructions where the 4 x 4 quad transposes (DPP) of the first version cost 64 VALU per tile in
        // an epilogue that is VALU-issue bound (tools/archive/mlp_timeline_probe.py).  Row stride 36 floats: 16-byte aligned, spreads both access patterns.
        constexpr int LS = 36;
        float* wl = csum + wave * (32 * LS);
        const int r8 = lane >> 3, c8 = (lane & 7) * 4;  // read side: row r8 + 8 k of the block, columns c8 .. c8 + 3
        // all bias values first, complete before the first store (loads and stores share vmcnt and return out of order against each other: a load
        // issued behind stores is only known done when the stores are)
        f32x4 b4[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) b4[t] = *reinterpret_cast<const f32x4*>(bias + t * 32 + c8);
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) wl[((r & 3) + 8 * (r >> 2) + 4 * h) * LS + i] = acc[t][r];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&wl[(r8 + 8 * k) * LS + c8]) + b4[t];
                if (EPI == 1) { v.x = elu_f(v.x); v.y = elu_f(v.y); v.z = elu_f(v.z); v.w = elu_f(v.w); }
                const int rr = rbase + r8 + 8 * k;
#ifdef BG_PROBE_NO_STORE  // tools/archive/mlp_nostore_probe.py: how much of the kernel is its store tail?  (never defined in the product build)
                if (rr < M && v.x == 12345.678f) *reinterpret_cast<f32x4*>(Y + (size_t)rr * ldy + t * 32 + c8) = v;
#else
                if (rr < M) *reinterpret_cast<f32x4*>(Y + (size_t)rr * ldy + t * 32 + c8) = v;
#endif
            }
            if (t < 3) BG_EPI_STAMP(12 + t);
        }
    } else {
        // backward: the same transposition through the wave's LDS block, then x elu'(aux) (aux arrives in the row layout: auxq[t][k] = 4 columns c8 .. of
        // row r8 + 8 k, fetched by the caller under its last MFMAs), the store, and the column sums: over the 4 rows of a lane, then over the 8 lanes
        // that share its columns (lane bits 3..5)
        constexpr int LS = 36;
        float* wl = csum + wave * (32 * LS);
        float* cpart = csum + 4 * 32 * LS;  // [4 waves][128 columns], behind the transposition blocks
        const int r8 = lane >> 3, c8 = (lane & 7) * 4;
#pragma unroll
        for (int t = 0; t < NT; t++) {
#pragma unroll
            for (int r = 0; r < 16; r++) wl[((r & 3) + 8 * (r >> 2) + 4 * h) * LS + i] = acc[t][r];
            f32x4 cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f32x4 a = auxq[t][k];
                f32x4 v = *reinterpret_cast<const f32x4*>(&wl[(r8 + 8 * k) * LS + c8]);
                const int rr = rbase + r8 + 8 * k;
                if (rr < M) {
                    v = f32x4{v.x * (a.x > 0.f ? 1.0f : a.x + 1.0f), v.y * (a.y > 0.f ? 1.0f : a.y + 1.0f),
                              v.z * (a.z > 0.f ? 1.0f : a.z + 1.0f), v.w * (a.w > 0.f ? 1.0f : a.w + 1.0f)};
                    *reinterpret_cast<f32x4*>(Y + (size_t)rr * ldy + t * 32 + c8) = v;
                    cs += v;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float x = cs[k];
                x += __shfl_xor(x, 8);
                x += __shfl_xor(x, 16);
                x += __shfl_xor(x, 32);
                cs[k] = x;
            }
            if (r8 == 0) *reinterpret_cast<f32x4*>(&cpart[wave * N + t * 32 + c8]) = cs;
        }
        __syncthreads();
        if (threadIdx.x < N)
            colpart[(size_t)bx * ldy + by * N + threadIdx.x] =
                cpart[threadIdx.x] + cpart[N + threadIdx.x] + cpart[2 * N + threadIdx.x] + cpart[3 * N + threadIdx.x];
    }
}

