// Split-bf16 form of the fused MLP layer kernels (opt-in; see the comment below and include/booster_gym_amd.h), gfx950 only.  Own translation
// unit: the default scheduler keeps it at three waves per SIMD without spills, max-ilp does not.
#include <hip/hip_runtime.h>

#include "../../include/booster_gym_amd.h"
#include "bg_mlp_tile.h"

extern int bg_set_error(int code, const char* msg);
extern int bg_colsum_finish_launch(int nb, int C, const float* partial, float* out, hipStream_t st);
#define HIP_OK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return bg_set_error(-2, hipGetErrorString(_e));               \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------------------------
// Split form of the same layer (opt-in, `terms` = 9 or 6): fp32 x fp32 products on the bf16 matrix pipe, which on gfx950 is 16 x the fp32 one.
// Every fp32 number is EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significant bits: hi = the top 16 bits of x, mid = the top 16 bits of
// x - hi, lo = x - hi - mid, each subtraction exact), the product of two bf16 numbers is exact in the fp32 accumulator of
// v_mfma_f32_32x32x16_bf16, so with all 9 cross terms x * w is accumulated WITHOUT the rounding of the product -- the same arithmetic as an
// fp32 FMA chain; 9 MFMAs of 32 cycles for a 32 x 32 x 16 block against 8 x 64 cycles of v_mfma_f32_32x32x2_f32: 1.78 x the fp32 MFMA rate.
// terms = 6 drops mid*lo, lo*mid, lo*lo (each <= 2^-24 of the product: one more rounding error per product, 2.67 x).
//   * weights: split once per optimiser step by split_planes_kernel into [n][k-chunk][plane][32] bf16, k permuted inside a chunk to the order
//     the A side produces (below), staged per chunk in LDS (3 planes x 64 B per column, row stride 208 B);
//   * activations: fp32 from HBM exactly as in the fp32 kernel (16-byte loads, lane (i, h) takes floats s * 8 + 4 h + 0..3 of its row), split in
//     registers (5.5 VALU per element, hidden in the MFMA shadow: an MFMA holds the issue port 8 of its 32 cycles).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SP_ROW = 48;  // dwords per weight column and k-chunk in LDS: 3 planes x 4 slots of 16 bytes (slot = MFMA step j x lane half h)

// (x0, x1) -> packed bf16 pairs (low half = x0) of the three planes
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hp, unsigned& mp, unsigned& lp) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    hp = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    mp = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    lp = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
// one 16-deep MFMA step's A operand: 8 floats (two 16-byte loads) -> three bf16x8
__device__ __forceinline__ void split_a8(const f32x4& lo4, const f32x4& hi4, u32x4 (&pl)[3]) {
    unsigned hp[4], mp[4], lp[4];
    split_pair(lo4.x, lo4.y, hp[0], mp[0], lp[0]);
    split_pair(lo4.z, lo4.w, hp[1], mp[1], lp[1]);
    split_pair(hi4.x, hi4.y, hp[2], mp[2], lp[2]);
    split_pair(hi4.z, hi4.w, hp[3], mp[3], lp[3]);
    pl[0] = u32x4{hp[0], hp[1], hp[2], hp[3]};
    pl[1] = u32x4{mp[0], mp[1], mp[2], mp[3]};
    pl[2] = u32x4{lp[0], lp[1], lp[2], lp[3]};
}
#define BG_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)
#define BG_PIN() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void read_b3(u32x4 (&bp)[3], const unsigned* sw) {
#pragma unroll
    for (int q = 0; q < 3; q++) bp[q] = *reinterpret_cast<const u32x4*>(sw + q * 16);
}
// One tile's MFMAs (small terms first) with the split of ONE pair of the next step's A operand placed in their shadow: an MFMA holds the issue
// port 8 of its 32 cycles, so the two or three VALU instructions behind each one cost nothing.  The order is pinned (sched_barrier): left alone
// the scheduler puts the 44 VALU instructions of a step in one clump in front of the first MFMA.
template <int TERMS, bool SPLIT>
__device__ __forceinline__ void tile_step(f32x16& acc, const u32x4 (&ap)[3], const u32x4 (&bp)[3], float x0, float x1, unsigned& hp, unsigned& mp,
                                          unsigned& lp) {
    unsigned u0 = 0, u1 = 0, v0 = 0, v1 = 0;
    float r0 = 0.f, r1 = 0.f, s0, s1;
    if (TERMS == 9) {
        BG_MFMA(acc, ap[2], bp[2]);
        BG_PIN();
        BG_MFMA(acc, ap[1], bp[2]);
        BG_PIN();
        BG_MFMA(acc, ap[2], bp[1]);
        BG_PIN();
    }
    BG_MFMA(acc, ap[0], bp[2]);
    if (SPLIT) { u0 = __float_as_uint(x0); u1 = __float_as_uint(x1); hp = __builtin_amdgcn_perm(u1, u0, 0x07060302u); }
    BG_PIN();
    BG_MFMA(acc, ap[1], bp[1]);
    if (SPLIT) { r0 = x0 - __uint_as_float(u0 & 0xffff0000u); r1 = x1 - __uint_as_float(u1 & 0xffff0000u); }
    BG_PIN();
    BG_MFMA(acc, ap[2], bp[0]);
    if (SPLIT) { v0 = __float_as_uint(r0); v1 = __float_as_uint(r1); mp = __builtin_amdgcn_perm(v1, v0, 0x07060302u); }
    BG_PIN();
    BG_MFMA(acc, ap[0], bp[1]);
    if (SPLIT) { s0 = r0 - __uint_as_float(v0 & 0xffff0000u); s1 = r1 - __uint_as_float(v1 & 0xffff0000u); lp = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u); }
    BG_PIN();
    BG_MFMA(acc, ap[1], bp[0]);
    BG_PIN();
    BG_MFMA(acc, ap[0], bp[0]);
    BG_PIN();
}
// One 16-deep MFMA step over the wave's 4 column tiles.  ap: this step's A planes; (n0, n1): the 8 floats of the NEXT step's A operand, split into
// apn meanwhile (SPLIT); sw: this lane's LDS row for this step; b0: the fragments of tile 0, already read; swn: LDS row of the next step, whose
// tile-0 fragments are read into b0 during tile 3 (nullptr: the next step reads a buffer that is not ready yet).
template <int TERMS, bool SPLIT, int NT>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[NT], const u32x4 (&ap)[3], const f32x4& n0, const f32x4& n1, u32x4 (&apn)[3], const unsigned* sw,
                                          u32x4 (&b0)[3], const unsigned* swn) {
    static_assert(NT == 4, "4 column tiles per wave");
    u32x4 b1[3];
    unsigned hp[4], mp[4], lp[4];
    read_b3(b1, sw + 1 * 32 * SP_ROW);
    BG_PIN();
    tile_step<TERMS, SPLIT>(acc[0], ap, b0, n0.x, n0.y, hp[0], mp[0], lp[0]);
    read_b3(b0, sw + 2 * 32 * SP_ROW);
    BG_PIN();
    tile_step<TERMS, SPLIT>(acc[1], ap, b1, n0.z, n0.w, hp[1], mp[1], lp[1]);
    read_b3(b1, sw + 3 * 32 * SP_ROW);
    BG_PIN();
    tile_step<TERMS, SPLIT>(acc[2], ap, b0, n1.x, n1.y, hp[2], mp[2], lp[2]);
    if (swn) read_b3(b0, swn);
    BG_PIN();
    tile_step<TERMS, SPLIT>(acc[3], ap, b1, n1.z, n1.w, hp[3], mp[3], lp[3]);
    if (SPLIT) {
        apn[0] = u32x4{hp[0], hp[1], hp[2], hp[3]};
        apn[1] = u32x4{mp[0], mp[1], mp[2], mp[3]};
        apn[2] = u32x4{lp[0], lp[1], lp[2], lp[3]};
    }
}
// One k-chunk of the weight planes (128 columns x 192 bytes) global -> LDS with no register stop (global_load_lds_dwordx4: the LDS side of
// one wave-instruction is 64 consecutive 16-byte slots, the global side is per lane).  Slot s holds column n = s / 12, piece
// (s % 12) ^ ((n >> 2) & 3) in its low two bits (piece = plane * 4 + step * 2 + lane half): the XOR spreads the 16 lanes of one read pass,
// whose rows are 192 bytes apart, over all 16 bank groups.  24 wave-instructions, 6 per wave.
__device__ __forceinline__ void stage_p_chunk(const unsigned* __restrict__ P, int CH, int kc, unsigned* sBbuf, int wave, int lane) {
#pragma unroll
    for (int u = 0; u < 6; u++) {
        const int q = u * 4 + wave, s = q * 64 + lane, n = s / 12, sig = s % 12;
        const int piece = (sig & ~3) | ((sig & 3) ^ ((n >> 2) & 3));
        const unsigned* g = P + ((size_t)n * CH + kc) * SP_ROW + piece * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(sBbuf + q * 256), 16, 0, 0);
    }
}

template <int K, int EPI, int NB, int TERMS>
__global__ __launch_bounds__(256, 2) void mlp_split_kernel(int M, int ldy, const float* __restrict__ X, const unsigned* __restrict__ Pfull,
                                                           const float* __restrict__ biasfull, float* __restrict__ Yfull,
                                                           const float* __restrict__ auxfull, float* __restrict__ colpart) {
    constexpr int N = 128;
    constexpr int NT = N / 32, CH = K / FW_KC;
    const int ncb = ldy / N, grp = blockIdx.x / (8 * ncb), rem = blockIdx.x % (8 * ncb);
    const int bx = grp * 8 + (rem & 7), by = rem >> 3;  // row slab, column block (XCD-aware, as in mlp_fwd_kernel)
    if (bx * FW_BM >= M) return;
    const unsigned* __restrict__ P = Pfull + (size_t)by * N * CH * SP_ROW;
    const float* __restrict__ bias = EPI <= 1 ? biasfull + by * N : nullptr;
    float* __restrict__ Y = Yfull + by * N;
    static_assert(CH % 2 == 0, "K must be a multiple of 64");
    __shared__ __attribute__((aligned(16))) unsigned sB[2][N * SP_ROW];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int row = bx * FW_BM + wave * 32 + i;
    const float* xrow = X + (size_t)(row < M ? row : M - 1) * K + 4 * h;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    f32x4 aA[4], aB[4];
    f32x4 auxq[EPI == 2 ? NT : 1][4];
    stage_p_chunk(P, CH, 0, sB[0], wave, lane);
    load_a_chunk(aA, xrow, 0);
    const int sx = (i >> 2) & 3;  // this lane's slot swizzle
    const unsigned* s00 = &sB[0][i * SP_ROW + ((0 + h) ^ sx) * 4];  // buffer 0, MFMA step 0
    const unsigned* s01 = &sB[0][i * SP_ROW + ((2 + h) ^ sx) * 4];  // buffer 0, step 1
    const unsigned* s10 = s00 + N * SP_ROW;
    const unsigned* s11 = s01 + N * SP_ROW;
    u32x4 ap0[3], ap1[3], b0[3];
    split_a8(aA[0], aA[1], ap0);  // the only split outside an MFMA shadow
    __syncthreads();
    // In-wave software pipeline: while the MFMAs of step (c, j) run, the A operand of the next step is split and the next tile's B fragments are
    // read; the global loads (A fp32, weight planes by LDS-DMA) of chunk c + 1 are issued at the top of chunk c.
    for (int kc = 0; kc < CH - 2; kc += 2) {
        stage_p_chunk(P, CH, kc + 1, sB[1], wave, lane);
        load_a_chunk(aB, xrow, kc + 1);
        read_b3(b0, s00);
        mfma_step<TERMS, true, NT>(acc, ap0, aA[2], aA[3], ap1, s00, b0, s01);
        mfma_step<TERMS, true, NT>(acc, ap1, aB[0], aB[1], ap0, s01, b0, nullptr);
        __syncthreads();
        stage_p_chunk(P, CH, kc + 2, sB[0], wave, lane);
        load_a_chunk(aA, xrow, kc + 2);
        read_b3(b0, s10);
        mfma_step<TERMS, true, NT>(acc, ap0, aB[2], aB[3], ap1, s10, b0, s11);
        mfma_step<TERMS, true, NT>(acc, ap1, aA[0], aA[1], ap0, s11, b0, nullptr);
        __syncthreads();
    }
    // last two chunks, peeled: the backward epilogue's elu' operand is fetched under the final MFMAs and occupies registers only from here on
    stage_p_chunk(P, CH, CH - 1, sB[1], wave, lane);
    load_a_chunk(aB, xrow, CH - 1);
    read_b3(b0, s00);
    mfma_step<TERMS, true, NT>(acc, ap0, aA[2], aA[3], ap1, s00, b0, s01);
    mfma_step<TERMS, true, NT>(acc, ap1, aB[0], aB[1], ap0, s01, b0, nullptr);
    __syncthreads();
    if constexpr (EPI == 2) {
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
    read_b3(b0, s10);
    mfma_step<TERMS, true, NT>(acc, ap0, aB[2], aB[3], ap1, s10, b0, s11);
    mfma_step<TERMS, false, NT>(acc, ap1, aB[0], aB[1], ap0, s11, b0, nullptr);
    __syncthreads();
    layer_epilogue<EPI, NT>(acc, auxq, M, ldy, bx, by, wave, lane, i, h, bias, Y, colpart, reinterpret_cast<float*>(&sB[0][0]));
}

// W [src_rows][ldw] fp32 (transpose: element (n, k) = W[k][n]) -> planes [n_out][k_out / 32][3][32] bf16; (n, k) outside the source = 0 (padded
// first layers).  Inside a chunk, k = s * 8 + 4 h + q (s 0..3, h 0..1, q 0..3) sits at position (s / 2) * 16 + h * 8 + (s % 2) * 4 + q: the
// 8 values lane (., h) of MFMA step j = s / 2 needs are 16 contiguous bytes.
__global__ __launch_bounds__(256) void split_planes_kernel(int n_out, int k_out, const float* __restrict__ W, int ldw, int src_rows, int src_cols,
                                                           int transpose, unsigned short* __restrict__ planes, int negated_copy) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_out * k_out) return;
    const int n = idx / k_out, k = idx % k_out;
    float x = 0.f;
    if (!transpose) { if (n < src_rows && k < src_cols) x = W[(size_t)n * ldw + k]; }
    else if (k < src_rows && n < src_cols) x = W[(size_t)k * ldw + n];
    const unsigned u = __float_as_uint(x);
    const float r = x - __uint_as_float(u & 0xffff0000u);
    const unsigned v = __float_as_uint(r);
    const float s2 = r - __uint_as_float(v & 0xffff0000u);
    const int kin = k & 31, s = kin >> 3, hh = (kin >> 2) & 1, q = kin & 3;
    const int pos = (s >> 1) * 16 + hh * 8 + (s & 1) * 4 + q;
    unsigned short* dst = planes + ((size_t)n * (k_out / 32) + (k >> 5)) * 96 + pos;
    dst[0] = (unsigned short)(u >> 16);
    dst[32] = (unsigned short)(v >> 16);
    dst[64] = (unsigned short)(__float_as_uint(s2) >> 16);
    if (negated_copy) {  // the planes of -W (the split is symmetric in the sign: every plane's sign bit flipped), behind the planes of W
        unsigned short* neg = dst + (size_t)n_out * k_out * 3;
        neg[0] = (unsigned short)((u >> 16) ^ 0x8000u);
        neg[32] = (unsigned short)((v >> 16) ^ 0x8000u);
        neg[64] = (unsigned short)((__float_as_uint(s2) >> 16) ^ 0x8000u);
    }
}

extern "C" int bg_mlp_split_weights(int32_t n_out, int32_t k_out, const float* W, int32_t ldw, int32_t src_rows, int32_t src_cols, int32_t transpose,
                                    uint16_t* planes, void* stream) {
    if (n_out <= 0 || k_out <= 0 || k_out % 32 || !W || !planes || ldw <= 0 || src_rows <= 0 || src_cols <= 0)
        return bg_set_error(-1, "bg_mlp_split_weights: bad argument");
    if (((uintptr_t)planes & 15) != 0) return bg_set_error(-1, "bg_mlp_split_weights: planes must be 16-byte aligned");
    hipLaunchKernelGGL(split_planes_kernel, dim3((n_out * k_out + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_out, k_out, W, ldw, src_rows, src_cols,
                       transpose, planes, 0);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_mlp_split_weights_pm(int32_t n_out, int32_t k_out, const float* W, int32_t ldw, int32_t src_rows, int32_t src_cols, int32_t transpose,
                                       uint16_t* planes, void* stream) {
    if (n_out <= 0 || k_out <= 0 || k_out % 32 || !W || !planes || ldw <= 0 || src_rows <= 0 || src_cols <= 0)
        return bg_set_error(-1, "bg_mlp_split_weights_pm: bad argument");
    if (((uintptr_t)planes & 15) != 0) return bg_set_error(-1, "bg_mlp_split_weights_pm: planes must be 16-byte aligned");
    hipLaunchKernelGGL(split_planes_kernel, dim3((n_out * k_out + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_out, k_out, W, ldw, src_rows, src_cols,
                       transpose, planes, 1);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_mlp_layer_forward_split(int32_t M, int32_t K, int32_t N, const float* X, const uint16_t* planes, const float* bias, float* Y,
                                          int32_t elu, int32_t terms, void* stream) {
    if (M <= 0 || !X || !planes || !bias || !Y) return bg_set_error(-1, "bg_mlp_layer_forward_split: bad argument");
    if ((((uintptr_t)X | (uintptr_t)planes | (uintptr_t)Y) & 15) != 0) return bg_set_error(-1, "bg_mlp_layer_forward_split: pointers must be 16-byte aligned");
    if (N % 128 != 0 || N > 1024) return bg_set_error(-4, "bg_mlp_layer_forward_split: unsupported N (multiples of 128 up to 1024)");
#ifndef BG_PROBE_TERMS
    if (terms != 9 && terms != 6) return bg_set_error(-4, "bg_mlp_layer_forward_split: terms must be 9 or 6");
#endif
    dim3 grid((((M + FW_BM - 1) / FW_BM + 7) / 8) * 8 * (N / 128)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const unsigned* P = reinterpret_cast<const unsigned*>(planes);
#define BG_FWD(KK, TT)                                                                                                            \
    if (K == KK && terms == TT) {                                                                                                 \
        if (elu) hipLaunchKernelGGL((mlp_split_kernel<KK, 1, 0, TT>), grid, block, 0, st, M, N, X, P, bias, Y, nullptr, nullptr);  \
        else hipLaunchKernelGGL((mlp_split_kernel<KK, 0, 0, TT>), grid, block, 0, st, M, N, X, P, bias, Y, nullptr, nullptr);      \
        HIP_OK(hipGetLastError());                                                                                                \
        return 0;                                                                                                                 \
    }
    BG_FWD(256, 9) BG_FWD(128, 9) BG_FWD(64, 9) BG_FWD(256, 6) BG_FWD(128, 6) BG_FWD(64, 6)
#ifdef BG_PROBE_TERMS
    BG_FWD(256, 1) BG_FWD(128, 1) BG_FWD(64, 1)
#endif
#undef BG_FWD
    return bg_set_error(-4, "bg_mlp_layer_forward_split: unsupported K (64, 128, 256)");
}

extern "C" int bg_mlp_layer_backward_split(int32_t M, int32_t K, int32_t N, const float* G, const uint16_t* planes_t, const float* act_below,
                                           float* Gout, float* bias_grad_below, float* scratch, int32_t terms, void* stream) {
    if (M <= 0 || !G || !planes_t || !act_below || !Gout || !bias_grad_below || !scratch) return bg_set_error(-1, "bg_mlp_layer_backward_split: bad argument");
    if ((((uintptr_t)G | (uintptr_t)planes_t | (uintptr_t)Gout | (uintptr_t)act_below) & 15) != 0)
        return bg_set_error(-1, "bg_mlp_layer_backward_split: pointers must be 16-byte aligned");
    if (N % 128 != 0 || N > 1024) return bg_set_error(-4, "bg_mlp_layer_backward_split: unsupported N (multiples of 128 up to 1024)");
    if (terms != 9 && terms != 6) return bg_set_error(-4, "bg_mlp_layer_backward_split: terms must be 9 or 6");
    const int nb = (M + FW_BM - 1) / FW_BM;
    dim3 grid(((nb + 7) / 8) * 8 * (N / 128)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const unsigned* P = reinterpret_cast<const unsigned*>(planes_t);
#define BG_BWD(KK, TT)                                                                                                            \
    if (K == KK && terms == TT) {                                                                                                 \
        hipLaunchKernelGGL((mlp_split_kernel<KK, 2, 0, TT>), grid, block, 0, st, M, N, G, P, nullptr, Gout, act_below, scratch);   \
        if (bg_colsum_finish_launch(nb, N, scratch, bias_grad_below, st)) return bg_set_error(-2, "bg_mlp_layer_backward_split: launch failed");  \
        HIP_OK(hipGetLastError());                                                                                                \
        return 0;                                                                                                                 \
    }
    BG_BWD(256, 9) BG_BWD(128, 9) BG_BWD(256, 6) BG_BWD(128, 6)
#undef BG_BWD
    return bg_set_error(-4, "bg_mlp_layer_backward_split: unsupported K (128, 256)");
}
