// Pieces shared by the chained kernels on the bf16 matrix pipe with fp32 semantics (bg_mlp_chain_split.hip: forward; bg_mlp_chain_split_bwd.hip:
// backward-data): the exact three-way split of an fp32 operand in MFMA-gap-sized pieces, the nine products of a tile, the weight planes' copy into LDS
// and their fragment reads.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SP_ROW = 48;              // dwords per weight row and 32-deep chunk: 3 planes x 2 steps x 2 lane halves x 16 bytes
constexpr int NMAX = 256;               // widest layer
constexpr int BUFDW = NMAX * SP_ROW;    // dwords per LDS buffer (48 KB)
constexpr int NBUF = 3, AHEAD = 2;
constexpr int CHAIN_MAX = 4;

template <int V> using IC = std::integral_constant<int, V>;
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

#define BG_PIN() __builtin_amdgcn_sched_barrier(0)
#define BG_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)

// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0, "");
    constexpr int n = N > 63 ? 63 : N;
    __builtin_amdgcn_s_waitcnt((n & 15) | ((n >> 4) << 14) | 0x0F70);
}

// ELU through v_exp_f32 (as bg_mlp_tile.h)
__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : __expf(x) - 1.0f; }

// The split of one pair of fp32 values into the three planes' packed bf16 pairs (low half = x0), in four pieces that ride behind four MFMAs:
// 3 + 3 + 3 + 2 VALU instructions (an MFMA gap takes four for free).
struct SplitTmp { unsigned m0, m1, n0, n1; float r0, r1, s0; };
template <int PH>
__device__ __forceinline__ void split_phase(float x0, float x1, SplitTmp& s, unsigned& hp, unsigned& mp, unsigned& lp) {
#ifndef BG_ABL_NOSPLIT
    if constexpr (PH == 0) {
        const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
        hp = __builtin_amdgcn_perm(u1, u0, 0x07060302u); s.m0 = u0 & 0xffff0000u; s.m1 = u1 & 0xffff0000u;
    }
    if constexpr (PH == 1) {
        s.r0 = x0 - __uint_as_float(s.m0); s.r1 = x1 - __uint_as_float(s.m1);
        mp = __builtin_amdgcn_perm(__float_as_uint(s.r1), __float_as_uint(s.r0), 0x07060302u);
    }
    if constexpr (PH == 2) { s.n0 = __float_as_uint(s.r0) & 0xffff0000u; s.n1 = __float_as_uint(s.r1) & 0xffff0000u; s.s0 = s.r0 - __uint_as_float(s.n0); }
    if constexpr (PH == 3) { const float s1 = s.r1 - __uint_as_float(s.n1); lp = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s.s0), 0x07060302u); }
#endif
}
// ELU of an accumulator element in two pieces (3 + 1 issue slots: the exponential counts double; then 3)
struct FinTmp { float v, e; };
__device__ __forceinline__ void fin_a(FinTmp& f, float x) {
#ifndef BG_ABL_NOFIN
    f.v = x; f.e = __expf(x);
#endif
}
__device__ __forceinline__ float fin_b(const FinTmp& f, float x) {
#ifndef BG_ABL_NOFIN
    return f.v > 0.f ? f.v : f.e - 1.0f;
#else
    return x;
#endif
}

// One tile's nine products (small terms first); fill(gap) runs behind MFMA number gap, pinned there.
template <class F>
__device__ __forceinline__ void mfma9(f32x16& acc, const u32x4 (&w)[3], const u32x4 (&x)[3], F&& fill) {
    BG_MFMA(acc, w[2], x[2]); fill(IC<0>{}); BG_PIN();
    BG_MFMA(acc, w[1], x[2]); fill(IC<1>{}); BG_PIN();
    BG_MFMA(acc, w[2], x[1]); fill(IC<2>{}); BG_PIN();
    BG_MFMA(acc, w[0], x[2]); fill(IC<3>{}); BG_PIN();
    BG_MFMA(acc, w[1], x[1]); fill(IC<4>{}); BG_PIN();
    BG_MFMA(acc, w[2], x[0]); fill(IC<5>{}); BG_PIN();
    BG_MFMA(acc, w[0], x[1]); fill(IC<6>{}); BG_PIN();
    BG_MFMA(acc, w[1], x[0]); fill(IC<7>{}); BG_PIN();
    BG_MFMA(acc, w[0], x[0]); fill(IC<8>{}); BG_PIN();
}

// One wave-instruction (64 slots of 16 bytes = 1 KiB) of the copy of a 32-deep k-chunk of a layer's weight planes, [n][K / 32][48 dwords] in global
// memory, -> LDS with no register stop.  Slot s = 12 n + sig of the chunk's LDS image holds piece (sig & ~3) | ((sig & 3) ^ ((n >> 2) & 3)) of row n
// (piece = plane * 4 + step * 2 + lane half): the XOR spreads the 16 lanes of one read pass, whose rows are 192 bytes apart, over all bank groups.
// 16 rows = 3 wave-instructions; piece q of a wave: rows 16 (4 (q / 3) + wave) .., instruction q % 3 of them; rowpart / piecepart: this lane's row
// (x 192 bytes) and piece (x 16 bytes) in each of the three.  N * 3 / 64 pieces per wave and chunk.
template <int Q>
__device__ __forceinline__ void dma_piece(const unsigned* __restrict__ P, int CH, int kc, unsigned lds_chunk_bytes, int wave, const unsigned (&rowpart)[3],
                                          const unsigned (&piecepart)[3]) {
    constexpr int u = Q / 3, m = Q % 3;
    const int g = u * 4 + wave;  // wave-uniform: rows 16 g .. 16 g + 15
    const unsigned* base = P + ((size_t)(16 * g) * CH + kc) * SP_ROW;
    const unsigned lds = lds_chunk_bytes + (unsigned)((g * 192 + m * 64) * 16);
    const unsigned lofs = rowpart[m] * (unsigned)CH + piecepart[m];
    // inline asm: the copies' bookkeeping is explicit (wait_vm), the compiler must not drain vmcnt for them; M0 cannot be named as a clobber
    // (reserved), the backend never keeps a value of its own live in M0 across an inline asm (see bg_mlp_chain.hip)
#ifndef BG_ABL_NODMA
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lofs), "s"(base), "s"(lds) : "memory");
#endif
}

// The same for the tile-major stream of the backward chain: the chunk is ONE 32-row tile of a layer's planes over ALL its CH k-chunks (a contiguous
// block of the [n][CH][48 dwords] array), imaged in LDS as [k-chunk][32 rows][192 bytes] with the same slot swizzle.  Piece q of a wave: group
// G = 4 (q / 3) + wave of 16 rows of one k-chunk (k-chunk G / 2, rows 16 (G % 2) ..), instruction q % 3 of it.  CH * 3 / 2 pieces per wave and chunk.
template <int Q>
__device__ __forceinline__ void dma_tile_piece(const unsigned* __restrict__ P, int CH, int tile, unsigned lds_chunk_bytes, int wave, const unsigned (&rowpart)[3],
                                               const unsigned (&piecepart)[3]) {
    constexpr int u = Q / 3, m = Q % 3;
    const int G = u * 4 + wave;  // wave-uniform
    const unsigned* base = P + ((size_t)(32 * tile + 16 * (G & 1)) * CH + (G >> 1)) * SP_ROW;
    const unsigned lds = lds_chunk_bytes + (unsigned)((G * 192 + m * 64) * 16);
    const unsigned lofs = rowpart[m] * (unsigned)CH + piecepart[m];
#ifndef BG_ABL_NODMA
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lofs), "s"(base), "s"(lds) : "memory");
#endif
}

// mfma9 for the first k-step of a tile: the accumulator starts from zero (C = 0 in the first product: no register initialisation)
template <class F>
__device__ __forceinline__ void mfma9_first(f32x16& acc, const u32x4 (&w)[3], const u32x4 (&x)[3], F&& fill) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[2]), __builtin_bit_cast(bf16x8, x[2]), zero, 0, 0, 0); fill(IC<0>{}); BG_PIN();
    BG_MFMA(acc, w[1], x[2]); fill(IC<1>{}); BG_PIN();
    BG_MFMA(acc, w[2], x[1]); fill(IC<2>{}); BG_PIN();
    BG_MFMA(acc, w[0], x[2]); fill(IC<3>{}); BG_PIN();
    BG_MFMA(acc, w[1], x[1]); fill(IC<4>{}); BG_PIN();
    BG_MFMA(acc, w[2], x[0]); fill(IC<5>{}); BG_PIN();
    BG_MFMA(acc, w[0], x[1]); fill(IC<6>{}); BG_PIN();
    BG_MFMA(acc, w[1], x[0]); fill(IC<7>{}); BG_PIN();
    BG_MFMA(acc, w[0], x[0]); fill(IC<8>{}); BG_PIN();
}

struct Frag { u32x4 p[3]; };
__device__ __forceinline__ void read_w(Frag& f, const unsigned* sw) {
#pragma unroll
    for (int q = 0; q < 3; q++) f.p[q] = *reinterpret_cast<const u32x4*>(sw + q * 16);
}

}  // namespace
