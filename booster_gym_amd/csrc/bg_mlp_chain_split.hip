// The forward chain of one network of the PPO update (reference utils/model.py:9-26: three Linear + ELU layers in front of the output layer; called
// from utils/runner.py:132,147) as ONE kernel per 128-row slab on the bf16 matrix pipe with fp32 semantics, gfx950 only.
//   * Arithmetic: every fp32 operand is EXACTLY the sum of three bf16 numbers (hi / mid / lo, 8 + 8 + 8 significant bits), a bf16 x bf16 product is
//     exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, and all 9 cross products are accumulated: x * w enters the fp32 sum unrounded, as in
//     an fp32 FMA chain (see bg_mlp_split.hip).  9 x 32 cycles per 32 x 32 x 16 block against 8 x 64 for v_mfma_f32_32x32x2_f32.
//   * Shape of the computation as in bg_mlp_chain.hip: the products are computed transposed (D = W X^T: the weight planes are the A operand, read from
//     LDS; the activations are the B operand), so the accumulator layout gives a lane ONE sample and 16 features per tile -- feature
//     32 t + (r & 3) + 8 (r >> 2) + 4 h in register r of tile t (h = lane >> 5) -- and registers 8 jj .. 8 jj + 7 of tile t ARE the 8 k-values lane (., h)
//     supplies to 16-deep MFMA step 2 t + jj of the next layer when the weights are laid out in that k order (bg_mlp_split_weights does: position
//     (s >> 1) * 16 + h * 8 + (s & 1) * 4 + q of k = 8 s + 4 h + q inside a 32-chunk).  Activations go from layer to layer in REGISTERS as fp32 and are
//     split into their three planes just in front of the step that multiplies them (11 VALU per pair, in the shadow of the MFMAs).
//   * One wave per SIMD (inputs K/2 registers + accumulators N/2), one workgroup per CU, persistent over its share of the slabs.  The weight planes of
//     all three layers are ONE stream of 32-deep k-chunks (192 bytes per output row) through three 48 KB LDS buffers, copied by
//     global_load_lds_dwordx4 two chunks ahead; one barrier per chunk; copies, loads and stores counted by hand (every wait asks for the vector-memory
//     operations of the chunk before last, no more).
//   * Nothing but MFMAs on the critical path where it can be helped: bias = the accumulators' initial value; ELU + store of tile T of a layer ride
//     in the MFMA gaps of chunk T - 1 of the NEXT layer (which needs only tiles < T), 8 elements per k-step; tile 0 of a layer is finished under the
//     other tiles of the layer's last k-step; an MFMA gap takes up to 4-5 VALU instructions for free (tools/probe/mfma_fillers.hip).
//   Results: fp32-exact products, another summation order than bg_mlp_chain.hip (not bit-identical to it; both are compared with float64 in the tests).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

#include "../../include/booster_gym_amd.h"

extern int bg_set_error(int code, const char* msg);

#ifdef BG_CHAIN_PROBE_STAMPS  // tools/chain_split_stamps.py: shader-clock stamps of every wave around every chunk barrier (never defined in the product build)
__device__ long long bg_split_stamp_buf[2 * 256 * 4 * 64];  // [N2 == 256][workgroup][wave][stamp]
extern "C" int bg_probe_read_split_stamps(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(bg_split_stamp_buf), bytes); }
#define BG_STAMP(K) stamps[K] = clock64()
#else
#define BG_STAMP(K) do { } while (0)
#endif

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
struct SplitGroup { int n; int begin[CHAIN_MAX + 1]; bg_mlp_chain_split net[CHAIN_MAX]; };

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
// 1 + 4 + 1 + 5 VALU instructions.
struct SplitTmp { unsigned u0, u1, v0, v1; float r0, r1; };
template <int PH>
__device__ __forceinline__ void split_phase(float x0, float x1, SplitTmp& s, unsigned& hp, unsigned& mp, unsigned& lp) {
    if constexpr (PH == 0) { s.u0 = __float_as_uint(x0); s.u1 = __float_as_uint(x1); hp = __builtin_amdgcn_perm(s.u1, s.u0, 0x07060302u); }
    if constexpr (PH == 1) { s.r0 = x0 - __uint_as_float(s.u0 & 0xffff0000u); s.r1 = x1 - __uint_as_float(s.u1 & 0xffff0000u); }
    if constexpr (PH == 2) { s.v0 = __float_as_uint(s.r0); s.v1 = __float_as_uint(s.r1); mp = __builtin_amdgcn_perm(s.v1, s.v0, 0x07060302u); }
    if constexpr (PH == 3) {
        const float s0 = s.r0 - __uint_as_float(s.v0 & 0xffff0000u), s1 = s.r1 - __uint_as_float(s.v1 & 0xffff0000u);
        lp = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
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

// One 32-deep k-chunk of a layer's weight planes, [n][K / 32][48 dwords] in global memory, -> LDS with no register stop.  The LDS side of one
// wave-instruction is 64 consecutive 16-byte slots; slot s = 12 n + sig holds piece (sig & ~3) | ((sig & 3) ^ ((n >> 2) & 3)) of row n
// (piece = plane * 4 + step * 2 + lane half): the XOR spreads the 16 lanes of one read pass, whose rows are 192 bytes apart, over all bank groups.
// 16 rows = 3 wave-instructions; rowpart / piecepart: this lane's row (x 192 bytes) and piece (x 16 bytes) in each of the three.
template <int N>
__device__ __forceinline__ void dma_chunk(const unsigned* __restrict__ P, int CH, int kc, unsigned* sbuf, int wave, const unsigned (&rowpart)[3],
                                          const unsigned (&piecepart)[3]) {
#pragma unroll
    for (int u = 0; u < N / 64; u++) {
        const int g = u * 4 + wave;  // wave-uniform: rows 16 g .. 16 g + 15
        const unsigned* base = P + ((size_t)(16 * g) * CH + kc) * SP_ROW;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const unsigned lds = (unsigned)(uintptr_t)(sbuf + (g * 192 + m * 64) * 4);
            const unsigned lofs = rowpart[m] * (unsigned)CH + piecepart[m];
            // inline asm: the copies' bookkeeping is explicit (wait_vm), the compiler must not drain vmcnt for them; M0 cannot be named as a clobber
            // (reserved), the backend never keeps a value of its own live in M0 across an inline asm (see bg_mlp_chain.hip)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lofs), "s"(base), "s"(lds) : "memory");
        }
    }
}

struct Frag { u32x4 p[3]; };
__device__ __forceinline__ void read_w(Frag& f, const unsigned* sw) {
#pragma unroll
    for (int q = 0; q < 3; q++) f.p[q] = *reinterpret_cast<const u32x4*>(sw + q * 16);
}

// One 128-row slab of one network.  Y1 / Y2 / Y3 hold whole slabs (every store is unconditional: their number is part of the vmcnt bookkeeping).
// sB: the three bias vectors, staged once per workgroup.
template <int K0, int N1, int N2, int N3>
__device__ __forceinline__ void split_slab(const bg_mlp_chain_split& a, int slab, unsigned* sW, const float* sB) {
    constexpr int C0 = K0 / 32, C1 = N1 / 32, C2 = N2 / 32, C = C0 + C1 + C2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    unsigned rowpart[3], piecepart[3];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const int sp = m * 64 + lane, nl = sp / 12, sig = sp % 12;
        rowpart[m] = (unsigned)nl * 192u;
        piecepart[m] = (unsigned)(((sig & ~3) | ((sig & 3) ^ ((nl >> 2) & 3))) * 16);
    }
    const unsigned* __restrict__ P1 = reinterpret_cast<const unsigned*>(a.P1);
    const unsigned* __restrict__ P2 = reinterpret_cast<const unsigned*>(a.P2);
    const unsigned* __restrict__ P3 = reinterpret_cast<const unsigned*>(a.P3);
    // chunk cc of the stream: copies per wave, and the stores a lane issues during the chunk
    struct S {
        static constexpr int ndma(int cc) { return (cc < 0 ? 0 : cc < C0 ? N1 : cc < C0 + C1 ? N2 : cc < C ? N3 : 0) * 3 / 64; }
        // tile 0 of a layer is stored in the layer's last chunk, tile T >= 1 in chunk T - 1 of the next layer (the last layer's: behind the loop)
        static constexpr int stores_in(int cc) {
            int n = 0;
            if (cc == C0 - 1 || cc == C0 + C1 - 1 || cc == C - 1) n += 4;
            if (cc >= C0 && cc < C0 + C1 && cc - C0 + 1 < N1 / 32) n += 4;
            if (cc >= C0 + C1 && cc < C && cc - C0 - C1 + 1 < N2 / 32) n += 4;
            return n;
        }
        // what a wave has issued behind the copies of chunk cc when it arrives at the top of iteration cc: everything of iteration cc - 1 (the copies
        // of chunk cc + 1 among it); before iteration 0: the copies of chunk 1
        static constexpr int behind(int cc) { return ndma(cc + 1) + (cc > 0 ? stores_in(cc - 1) : 0); }
    };
    auto dma = [&](auto cc_) {
        constexpr int cc = decltype(cc_)::value;
        unsigned* dst = sW + (cc % NBUF) * BUFDW;
        if constexpr (cc < C0) dma_chunk<N1>(P1, C0, cc, dst, wave, rowpart, piecepart);
        else if constexpr (cc < C0 + C1) dma_chunk<N2>(P2, C1, cc - C0, dst, wave, rowpart, piecepart);
        else if constexpr (cc < C) dma_chunk<N3>(P3, C2, cc - C0 - C1, dst, wave, rowpart, piecepart);
    };
#ifdef BG_CHAIN_PROBE_STAMPS
    long long stamps[64];
    stamps[62] = wall_clock64();
#endif
    BG_STAMP(0);
    const int row = slab * 128 + wave * 32 + i;
    const float* xrow = a.X + (size_t)(row < a.M ? row : a.M - 1) * K0 + 4 * h;
    float x0[K0 / 2];
#pragma unroll
    for (int j = 0; j < K0 / 8; j++) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + 8 * j);
        x0[4 * j + 0] = v.x; x0[4 * j + 1] = v.y; x0[4 * j + 2] = v.z; x0[4 * j + 3] = v.w;
    }
    BG_PIN();
    dma(IC<0>{});
    dma(IC<1>{});
    BG_PIN();
    const int sx = (i >> 2) & 3;                      // this lane's slot swizzle (rows 32 t + i: the tile offset does not change it)
    const unsigned* swl = sW + i * SP_ROW;            // + buffer, + tile * 32 rows, + slot
    const int s0 = ((0 + h) ^ sx) * 4, s1 = ((2 + h) ^ sx) * 4;  // step 0 / step 1 of a chunk
    f32x16 a1[N1 / 32], a2[N2 / 32], a3[N3 / 32];
    u32x4 xp[3];
    unsigned xn[3][4];
    // planes of the first k-step of the first layer (the only split outside an MFMA shadow)
    {
        SplitTmp st;
        static_for<4>([&](auto p_) {
            constexpr int p = decltype(p_)::value;
            unsigned hp, mp, lp;
            split_phase<0>(x0[2 * p], x0[2 * p + 1], st, hp, mp, lp);
            split_phase<1>(x0[2 * p], x0[2 * p + 1], st, hp, mp, lp);
            split_phase<2>(x0[2 * p], x0[2 * p + 1], st, hp, mp, lp);
            split_phase<3>(x0[2 * p], x0[2 * p + 1], st, hp, mp, lp);
            xp[0][p] = hp; xp[1][p] = mp; xp[2][p] = lp;
        });
    }
    const size_t rowofs = (size_t)row;
    // ELU of element r of tile t in place; store of 4 finished elements
    auto fin = [&](auto& A, auto t_, auto r_) { constexpr int t = decltype(t_)::value, r = decltype(r_)::value; A[t][r] = elu_f(A[t][r]); };
    auto store4 = [&](auto& A, float* __restrict__ Y, auto N_, auto t_, auto g_) {
        constexpr int N = decltype(N_)::value, t = decltype(t_)::value, g = decltype(g_)::value;
        const f32x4 v = {A[t][4 * g + 0], A[t][4 * g + 1], A[t][4 * g + 2], A[t][4 * g + 3]};
        *reinterpret_cast<f32x4*>(Y + rowofs * N + 32 * t + 8 * g + 4 * h) = v;
    };
    // One layer: chunks base .. base + K / 32 - 1 of the stream.  xin(s): the lane's s'th input value (k-step J takes values 8 J .. 8 J + 7);
    // prev / Yprev / NP: the layer below, whose tiles >= 1 are finished here (NP = 0: none); Y: this layer's activations; LAST: nothing follows.
    auto layer = [&](auto& acc, auto xin, auto& prev, float* __restrict__ Yprev, auto NP_, auto K_, auto N_, auto base_, int bias_ofs, float* __restrict__ Y,
                     auto LAST_) {
        constexpr int K = decltype(K_)::value, N = decltype(N_)::value, NT = N / 32, CH = K / 32, base = decltype(base_)::value;
        constexpr int NP = decltype(NP_)::value, NTP = NP / 32;
        constexpr bool LAST = decltype(LAST_)::value;
        constexpr int EPT = 8 / NT;  // elements of the tile below finished per tile-step (8 per k-step)
        static_assert(NT == 4 || NT == 8, "4 or 8 tiles per layer");
        // bias = initial value of the accumulators (feature 32 t + 8 g + 4 h + q in register 4 g + q)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sB[bias_ofs + 32 * t + 8 * g + 4 * h]);
                acc[t][4 * g + 0] = b4.x; acc[t][4 * g + 1] = b4.y; acc[t][4 * g + 2] = b4.z; acc[t][4 * g + 3] = b4.w;
            }
        static_for<CH>([&](auto kc_) {
            constexpr int kc = decltype(kc_)::value, c = base + kc;
            const unsigned* sw = swl + (c % NBUF) * BUFDW;
            // chunk c complete in LDS (this wave's part), then published by the barrier; what iteration c - 1 issued may stay in flight
            BG_PIN();
            BG_STAMP(1 + 3 * c);
            wait_vm<S::behind(c)>();
            asm volatile("s_barrier" ::: "memory");  // no fence: a workgroup fence would drain vmcnt (stores and younger copies included)
            BG_STAMP(2 + 3 * c);
            BG_PIN();
            dma(IC<c + AHEAD>{});
            BG_PIN();
            BG_STAMP(3 + 3 * c);
            Frag fr[2];
            read_w(fr[0], sw + s0);
            static_for<2>([&](auto j_) {
                constexpr int j = decltype(j_)::value, J = 2 * kc + j;
                constexpr bool lastk = (J == K / 16 - 1);
                static_for<NT>([&](auto t_) {
                    constexpr int t = decltype(t_)::value, ts = j * NT + t;
                    SplitTmp st0, st1;
                    mfma9(acc[t], fr[ts & 1].p, xp, [&](auto g_) {
                        constexpr int g = decltype(g_)::value;
                        // (A) the next tile-step's weight fragments
                        if constexpr (g == 0 && ts + 1 < 2 * NT) read_w(fr[(ts + 1) & 1], sw + ((ts + 1) / NT ? s1 : s0) + ((ts + 1) % NT) * 32 * SP_ROW);
                        if constexpr (!lastk) {
                            // (B) the planes of k-step J + 1: one pair per NT / 4 tile-steps, pieces behind MFMAs 3 .. 6
                            if constexpr (t % (NT / 4) == 0 && g >= 3 && g <= 6) {
                                constexpr int p = t / (NT / 4), s = 8 * (J + 1) + 2 * p;
                                split_phase<g - 3>(xin(IC<s>{}), xin(IC<s + 1>{}), st0, xn[0][p], xn[1][p], xn[2][p]);
                            }
                            // (C) tile kc + 1 of the layer below: elements 8 j .. 8 j + 7 during this k-step, stored by fours
                            if constexpr (NTP > 0 && kc + 1 < NTP) {
                                if constexpr (EPT == 1) {
                                    if constexpr (g == 1) fin(prev, IC<kc + 1>{}, IC<8 * j + t>{});
                                    if constexpr (g == 8 && (t & 3) == 3) store4(prev, Yprev, IC<NP>{}, IC<kc + 1>{}, IC<(8 * j + t) / 4>{});
                                } else {
                                    if constexpr (g == 1) fin(prev, IC<kc + 1>{}, IC<8 * j + 2 * t>{});
                                    if constexpr (g == 7) fin(prev, IC<kc + 1>{}, IC<8 * j + 2 * t + 1>{});
                                    if constexpr (g == 8 && (t & 1) == 1) store4(prev, Yprev, IC<NP>{}, IC<kc + 1>{}, IC<(8 * j + 2 * t) / 4>{});
                                }
                            }
                        } else if constexpr (!LAST) {
                            // the layer's last k-step: tile 0 is complete behind tile-step 0.  It is finished under the other tiles, and the planes of the
                            // NEXT layer's first k-step (its elements 0 .. 7) are split at the end.
                            if constexpr (NT == 8) {
                                if constexpr (t >= 1 && t <= 4) {
                                    if constexpr (g == 1 || g == 3 || g == 5 || g == 7) fin(acc, IC<0>{}, IC<4 * (t - 1) + (g - 1) / 2>{});
                                    if constexpr (g == 8) store4(acc, Y, IC<N>{}, IC<0>{}, IC<t - 1>{});
                                }
                                if constexpr (t == 5 || t == 6) {
                                    constexpr int p = 2 * (t - 5);
                                    if constexpr (g <= 3) split_phase<g>(acc[0][2 * p], acc[0][2 * p + 1], st0, xn[0][p], xn[1][p], xn[2][p]);
                                    else if constexpr (g <= 7) split_phase<g - 4>(acc[0][2 * p + 2], acc[0][2 * p + 3], st1, xn[0][p + 1], xn[1][p + 1], xn[2][p + 1]);
                                }
                            } else {
                                if constexpr (t == 1 || t == 2) {
                                    if constexpr (g <= 7) fin(acc, IC<0>{}, IC<8 * (t - 1) + g>{});
                                    if constexpr (g == 8) { store4(acc, Y, IC<N>{}, IC<0>{}, IC<2 * (t - 1)>{}); store4(acc, Y, IC<N>{}, IC<0>{}, IC<2 * (t - 1) + 1>{}); }
                                }
                                if constexpr (t == 3 && g <= 7) {
                                    constexpr int p = g / 2;
                                    if constexpr ((g & 1) == 0) {
                                        split_phase<0>(acc[0][2 * p], acc[0][2 * p + 1], st0, xn[0][p], xn[1][p], xn[2][p]);
                                        split_phase<1>(acc[0][2 * p], acc[0][2 * p + 1], st0, xn[0][p], xn[1][p], xn[2][p]);
                                    } else {
                                        split_phase<2>(acc[0][2 * p], acc[0][2 * p + 1], st0, xn[0][p], xn[1][p], xn[2][p]);
                                        split_phase<3>(acc[0][2 * p], acc[0][2 * p + 1], st0, xn[0][p], xn[1][p], xn[2][p]);
                                    }
                                }
                            }
                        } else {
                            // the last layer's last k-step: tile t - 1 whole under tile t (tile NT - 1: behind the loop)
                            if constexpr (t >= 1 && g <= 7) {
                                fin(acc, IC<t - 1>{}, IC<2 * g>{});
                                fin(acc, IC<t - 1>{}, IC<2 * g + 1>{});
                                if constexpr (g & 1) store4(acc, Y, IC<N>{}, IC<t - 1>{}, IC<g / 2>{});
                            }
                        }
                    });
                });
                if constexpr (!(lastk && LAST)) {
#pragma unroll
                    for (int q = 0; q < 3; q++) xp[q] = u32x4{xn[q][0], xn[q][1], xn[q][2], xn[q][3]};
                }
            });
        });
    };
    auto x0in = [&](auto s_) { return x0[decltype(s_)::value]; };
    auto a1in = [&](auto s_) { constexpr int s = decltype(s_)::value; return a1[s >> 4][s & 15]; };
    auto a2in = [&](auto s_) { constexpr int s = decltype(s_)::value; return a2[s >> 4][s & 15]; };
    layer(a1, x0in, a1, nullptr, IC<0>{}, IC<K0>{}, IC<N1>{}, IC<0>{}, 0, a.Y1, std::false_type{});
    layer(a2, a1in, a1, a.Y1, IC<N1>{}, IC<N1>{}, IC<N2>{}, IC<C0>{}, N1, a.Y2, std::false_type{});
    layer(a3, a2in, a2, a.Y2, IC<N2>{}, IC<N2>{}, IC<N3>{}, IC<C0 + C1>{}, N1 + N2, a.Y3, std::true_type{});
    // the last tile of the last layer
    {
        constexpr int t = N3 / 32 - 1;
        static_for<16>([&](auto r_) { fin(a3, IC<t>{}, r_); });
        static_for<4>([&](auto g_) { store4(a3, a.Y3, IC<N3>{}, IC<t>{}, g_); });
    }
    if (a.v_out) {
        // scalar output layer on the last activations, straight from the registers that hold them: the lane has 64 of its sample's 128 features
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < N3 / 32; t++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.v_w + 32 * t + 8 * g + 4 * h);
                part = fmaf(a3[t][4 * g + 0], w4.x, part); part = fmaf(a3[t][4 * g + 1], w4.y, part);
                part = fmaf(a3[t][4 * g + 2], w4.z, part); part = fmaf(a3[t][4 * g + 3], w4.w, part);
            }
        part += __shfl_xor(part, 32);
        if (h == 0 && row < a.M) a.v_out[row] = part + a.v_b[0];
    }
#ifdef BG_CHAIN_PROBE_STAMPS
    stamps[1 + 3 * C] = clock64();
    stamps[63] = wall_clock64();
    if (lane == 0 && blockIdx.x < 256)
        for (int k = 0; k < 64; k++) bg_split_stamp_buf[(((size_t)(N2 == 256) * 256 + blockIdx.x) * 4 + wave) * 64 + k] = (k <= 1 + 3 * C || k >= 62) ? stamps[k] : 0;
#endif
}

// TAG: 1 / 2 = one network with N2 = 128 / 256 (one kernel symbol per shape: a profiler's per-kernel average is the average of ONE shape),
// 0 = a group, shapes looked up per workgroup.
template <int TAG>
__global__ __launch_bounds__(256) void mlp_chain_split_fwd_kernel(SplitGroup grp) {
    __shared__ __attribute__((aligned(16))) unsigned sW[NBUF * BUFDW];
    __shared__ __attribute__((aligned(16))) float sB[3 * NMAX];
    int k = 0;
    if constexpr (TAG == 0) {
#pragma unroll
        for (int j = 1; j < CHAIN_MAX; j++)
            if (j < grp.n && (int)blockIdx.x >= grp.begin[j]) k = j;
    }
    const bg_mlp_chain_split& a = grp.net[k];
    for (int j = threadIdx.x; j < a.N1 + a.N2 + a.N3; j += 256) sB[j] = j < a.N1 ? a.b1[j] : j < a.N1 + a.N2 ? a.b2[j - a.N1] : a.b3[j - a.N1 - a.N2];
    __syncthreads();
    // One slab per workgroup, or (workgroups > 0) that many workgroups walking the network's slabs (see bg_mlp_chain.hip: two launches side by side
    // share the chip by CUs; counts that are multiples of the 8 XCDs)
    const int nslabs = (a.M + 127) / 128, stride = grp.begin[k + 1] - grp.begin[k];
    for (int slab = blockIdx.x - grp.begin[k]; slab < nslabs; slab += stride) {
        if (TAG == 2 || (TAG == 0 && a.N2 == 256)) split_slab<64, 256, 256, 128>(a, slab, sW, sB);
        else split_slab<64, 256, 128, 128>(a, slab, sW, sB);
        // every wave has read the last chunk before anybody's copies of the next slab land in the buffers
        asm volatile("s_barrier" ::: "memory");
    }
}

int split_check(const bg_mlp_chain_split& q) {
    if (q.M <= 0 || !q.X || !q.P1 || !q.b1 || !q.P2 || !q.b2 || !q.P3 || !q.b3 || !q.Y1 || !q.Y2 || !q.Y3) return bg_set_error(-1, "bg_mlp_chain_forward_split: bad argument");
    if ((((uintptr_t)q.X | (uintptr_t)q.P1 | (uintptr_t)q.P2 | (uintptr_t)q.P3 | (uintptr_t)q.Y1 | (uintptr_t)q.Y2 | (uintptr_t)q.Y3 | (uintptr_t)q.b1 |
          (uintptr_t)q.b2 | (uintptr_t)q.b3) & 15) != 0)
        return bg_set_error(-1, "bg_mlp_chain_forward_split: pointers must be 16-byte aligned");
    if ((q.v_w || q.v_b || q.v_out) && (!q.v_w || !q.v_b || !q.v_out || ((uintptr_t)q.v_w & 15) != 0))
        return bg_set_error(-1, "bg_mlp_chain_forward_split: value head needs v_w (16-byte aligned), v_b and v_out");
    if (!(q.K0 == 64 && q.N1 == 256 && (q.N2 == 128 || q.N2 == 256) && q.N3 == 128))
        return bg_set_error(-4, "bg_mlp_chain_forward_split: unsupported widths (64-256-128-128 and 64-256-256-128)");
    if (q.workgroups < 0) return bg_set_error(-1, "bg_mlp_chain_forward_split: workgroups < 0");
    return 0;
}

}  // namespace

extern "C" int bg_mlp_chain_forward_split(const bg_mlp_chain_split* nets, int32_t count, void* stream) {
    if (!nets || count <= 0 || count > CHAIN_MAX) return bg_set_error(-1, "bg_mlp_chain_forward_split: 1 to 4 networks");
    SplitGroup grp;
    grp.n = count;
    int blocks = 0;
    for (int k = 0; k < count; k++) {
        const int rc = split_check(nets[k]);
        if (rc) return rc;
        grp.begin[k] = blocks;
        grp.net[k] = nets[k];
        const int slabs = (nets[k].M + 127) / 128;
        blocks += nets[k].workgroups > 0 && nets[k].workgroups < slabs ? nets[k].workgroups : slabs;
    }
    grp.begin[count] = blocks;
    if (count > 1) hipLaunchKernelGGL(mlp_chain_split_fwd_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    else if (nets[0].N2 == 256) hipLaunchKernelGGL(mlp_chain_split_fwd_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    else hipLaunchKernelGGL(mlp_chain_split_fwd_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    if (hipGetLastError() != hipSuccess) return bg_set_error(-2, "bg_mlp_chain_forward_split: launch failed");
    return 0;
}
