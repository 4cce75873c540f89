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

#include "bg_chain_split.h"

namespace {

struct SplitGroup { int n; int begin[CHAIN_MAX + 1]; bg_mlp_chain_split net[CHAIN_MAX]; };
constexpr int SBNEG = 3 * NMAX + 128;   // floats between a bias in LDS and its negated copy

// All slabs first, first + stride, ... < nslabs of one network: ONE continuous stream of weight chunks, the slabs' own prologues and epilogues folded
// into their neighbours' MFMA shadows:
//   * the copies of the next slab's first two chunks go out during this slab's last two chunks; the next slab's input rows are loaded during layer 3
//     (the registers of the layer-1 input are free from layer 2 on) and the planes of its first k-step are split under layer 3's last MFMAs;
//   * the accumulators start as the bias, read straight from LDS into the accumulator registers while the layer before runs (a1: during layer 3 of
//     the slab before);
//   * ELU + store + value head of ALL of layer 3's tiles ride in the gaps of layer 1 of the NEXT slab (behind the loop for the last slab).
// Y1 / Y2 / Y3 hold whole slabs (every store is unconditional: their number is part of the vmcnt bookkeeping).
// sB: the three bias vectors, then v_w [N3] (zeros without a value head), staged once per workgroup.
template <int K0, int N1, int N2, int N3>
__device__ __forceinline__ void split_net(const bg_mlp_chain_split& a, int first, int stride, int nslabs, unsigned* sW, const float* sB) {
    constexpr int C0 = K0 / 32, C1 = N1 / 32, C2 = N2 / 32, C = C0 + C1 + C2;
    constexpr int NT1 = N1 / 32, NT2 = N2 / 32, NT3 = N3 / 32;
    static_assert(K0 == 64 && NT1 == 8 && NT3 == 4 && (NT2 == 4 || NT2 == 8) && C2 >= 2, "the reference's widths");
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
    // The stream: chunk cc (cc >= C: chunk cc - C of the next slab).  Per chunk and wave: the copies issued during it (those of chunk cc + 2), the
    // stores and loads every lane issues during it.
    struct S {
        static constexpr int rows(int cc) { return (cc % C) < C0 ? N1 : (cc % C) < C0 + C1 ? N2 : N3; }
        static constexpr int ndma(int cc) { return rows(cc) * 3 / 64; }
        static constexpr int tilesteps(int cc) { return 2 * rows(cc) / 32; }
        // copies per tile-step of chunk cc (those of chunk cc + 2, dealt from the chunk's first tile-step on)
        static constexpr int pp(int cc) { return (ndma(cc + 2) + tilesteps(cc) - 1) / tilesteps(cc); }
        // layer 3 of the slab before is finished two elements per tile-step during k-steps 0 .. 2 of layer 1 (48 elements) and one per tile-step during
        // the first 16 tile-steps of layer 2 (chunk 0 at 8 tiles, chunks 0 and 1 at 4)
        static constexpr int stores_in(int cc) {
            if (cc == 0) return 8;                        // layer 3 of the slab before: elements 0 .. 31
            if (cc == 1) return 4 + 4;                    // ... 32 .. 47; tile 0 of layer 1
            if (cc < C0 + C1) {                           // tile cc - C0 + 1 of layer 1 (the last chunk: tile 0 of layer 2) ...
                const int kc = cc - C0;
                return 4 + (NT2 == 8 ? (kc == 0 ? 4 : 0) : (kc <= 1 ? 2 : 0));  // ... and elements 48 .. 63 of layer 3
            }
            return cc - C0 - C1 + 1 < NT2 ? 4 : 0;        // tile cc - C0 - C1 + 1 of layer 2
        }
        static constexpr int loads_in(int cc) { return cc == C0 + C1 ? K0 / 8 : 0; }  // the next slab's input rows
        static constexpr int ops_in(int cc) { return ndma(cc + 2) + stores_in(cc) + loads_in(cc); }
        // The barrier that publishes chunk cc stands in front of the LAST tile-step of chunk cc - 1 (whose MFMAs then cover the latency of the first
        // fragment read of chunk cc).  What a wave has issued behind the last copy of chunk cc (issued during chunk cc - 2) when it arrives there: at
        // least everything of chunk cc - 1 but what its last tile-step issues (never more than 3 operations: a store or two, a load)
        static constexpr int behind(int cc) { const int n = ops_in((cc + C - 1) % C) - 3; return n > 0 ? n : 0; }
    };
    // LDS image of stream chunk cc: buffer (cc + phase) % 3, the phase advancing by C per slab; bb[r]: byte offset of the buffer of chunks cc % 3 == r
    unsigned bb[3] = {0u, (unsigned)BUFDW * 4u, 2u * (unsigned)BUFDW * 4u};
    const unsigned sWbase = (unsigned)(uintptr_t)sW;
    // Odd slabs accumulate the NEGATED sums when a.alternate is set (planes of -W, -bias as the accumulators' start; include/booster_gym_amd.h: the
    // rounding bias of the MFMA accumulator then cancels in what is summed over rows); sgn / sgnp put a finished tile of this slab / the one before right
    bool negc = false, negn = false;
    float sgn = 1.0f, sgnp = 1.0f;
    auto dma = [&](auto cc_, auto q_) {
        constexpr int cc = decltype(cc_)::value, q = decltype(q_)::value, c = cc % C;
        if constexpr (q < S::ndma(cc)) {
            const unsigned dst = sWbase + bb[cc % 3];
            const bool neg = cc < C ? negc : negn;   // (the slab the chunk belongs to: this one or the next; odd slabs read the planes of -W)
            if constexpr (c < C0) dma_piece<q>(P1 + (neg ? (size_t)N1 * K0 * 3 / 2 : 0), C0, c, dst, wave, rowpart, piecepart);
            else if constexpr (c < C0 + C1) dma_piece<q>(P2 + (neg ? (size_t)N2 * N1 * 3 / 2 : 0), C1, c - C0, dst, wave, rowpart, piecepart);
            else dma_piece<q>(P3 + (neg ? (size_t)N3 * N2 * 3 / 2 : 0), C2, c - C0 - C1, dst, wave, rowpart, piecepart);
        }
    };
#ifdef BG_CHAIN_PROBE_STAMPS
    long long stamps[64];
#endif
    const int sx = (i >> 2) & 3;                                 // this lane's slot swizzle (rows 32 t + i: the tile offset does not change it)
    const int s0 = ((0 + h) ^ sx) * 4, s1 = ((2 + h) ^ sx) * 4;  // step 0 / step 1 of a chunk
    const unsigned* swl = sW + i * SP_ROW;                       // + buffer, + tile * 32 rows, + slot
    f32x16 a1[NT1], a2[NT2], a3[NT3];
    float x0[K0 / 2];
    u32x4 xp[3];
    unsigned xn[3][4] = {};
    Frag fr[2];
    f32x4 wv[3];   // value-head weights of the group of four layer-3 elements in work, of the one before (elements may still be pending) and of the next
    float part = 0.f;
    const float vbias = a.v_out ? a.v_b[0] : 0.f;  // (read once: a load in the middle of the stream would be waited for with everything else)
    // bias -> accumulators: feature 32 t + 8 g + 4 h + q in register 4 g + q of tile t (one 16-byte LDS read, straight into the accumulator registers)
    auto init4 = [&](auto& A, int ofs, bool neg, auto t_, auto g_) {   // (neg: the negated biases, SBNEG floats further)
        constexpr int t = decltype(t_)::value, g = decltype(g_)::value;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sB[(neg ? SBNEG : 0) + ofs + 32 * t + 8 * g + 4 * h]);
        A[t][4 * g + 0] = b4.x; A[t][4 * g + 1] = b4.y; A[t][4 * g + 2] = b4.z; A[t][4 * g + 3] = b4.w;
    };
    auto loadx = [&](int r, auto j_) {
        constexpr int j = decltype(j_)::value;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.X + (size_t)(r < a.M ? r : a.M - 1) * K0 + 4 * h + 8 * j);
        x0[4 * j + 0] = v.x; x0[4 * j + 1] = v.y; x0[4 * j + 2] = v.z; x0[4 * j + 3] = v.w;
    };
    // ELU of element r of tile t in place, in two pieces (fin_a / fin_b)
    auto fa = [&](FinTmp& f, auto& A, auto t_, auto r_, float sg) { fin_a(f, A[decltype(t_)::value][decltype(r_)::value] * sg); };
    auto fb = [&](const FinTmp& f, auto& A, auto t_, auto r_) { constexpr int t = decltype(t_)::value, r = decltype(r_)::value; A[t][r] = fin_b(f, A[t][r]); };
    auto store4 = [&](auto& A, float* __restrict__ Y, int r, auto N_, auto t_, auto g_) {
#ifndef BG_ABL_NOSTORE
        constexpr int N = decltype(N_)::value, t = decltype(t_)::value, g = decltype(g_)::value;
        const f32x4 v = {A[t][4 * g + 0], A[t][4 * g + 1], A[t][4 * g + 2], A[t][4 * g + 3]};
        *reinterpret_cast<f32x4*>(Y + (size_t)r * N + 32 * t + 8 * g + 4 * h) = v;
#endif
    };
    auto vw4 = [&](auto G_) { constexpr int G = decltype(G_)::value; wv[G % 3] = *reinterpret_cast<const f32x4*>(&sB[N1 + N2 + N3 + 32 * (G / 4) + 8 * (G % 4) + 4 * h]); };
    // element e of layer 3 (tile e / 16, register e % 16): ELU in place (two pieces) + its term of the value head; the weights of group e / 4 + 1 are
    // fetched with the first element of group e / 4
    auto l3_a = [&](FinTmp& f, auto e_) {
        constexpr int e = decltype(e_)::value;
        if constexpr (e % 4 == 0 && e / 4 + 1 < 16) vw4(IC<e / 4 + 1>{});
        fa(f, a3, IC<e / 16>{}, IC<e % 16>{}, sgnp);
    };
    auto l3_b = [&](const FinTmp& f, auto e_, int rowp) {
        constexpr int e = decltype(e_)::value;
        fb(f, a3, IC<e / 16>{}, IC<e % 16>{});
        part = fmaf(a3[e / 16][e % 16], wv[(e / 4) % 3][e % 4], part);
        // ONE chain of 64 fused multiply-adds in element order, wherever this runs: in the stream the pieces are pinned between MFMAs anyway; in the tail
        // (straight-line code) -fassociative-math would re-associate the chain into several partial sums, and a slab's value would depend on whether it
        // is a workgroup's last one -- that is on how the slabs are dealt to workgroups (tests: "...however the slabs are dealt...")
        asm volatile("" : "+v"(part));
        if constexpr (e % 4 == 3) store4(a3, a.Y3, rowp, IC<N3>{}, IC<e / 16>{}, IC<(e / 4) % 4>{});
    };
    auto split_pair_at = [&](auto ph_, float v0, float v1, SplitTmp& st, auto p_) {
        constexpr int p = decltype(p_)::value;
        split_phase<decltype(ph_)::value>(v0, v1, st, xn[0][p], xn[1][p], xn[2][p]);
    };
    // The top of chunk cc: its copies have landed (this wave's part), the barrier publishes them and tells that everybody has read chunk cc - 1's
    // fragments (the copies of chunk cc + 2 will overwrite them); then the first fragment read of the chunk.
    auto chunk_top = [&](auto cc_) {
        constexpr int cc = decltype(cc_)::value;
        BG_PIN();
        BG_STAMP(1 + 2 * (cc % C));
        wait_vm<S::behind(cc)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's fragment reads of the chunk before are in its registers
        asm volatile("s_barrier" ::: "memory");  // no fence: a workgroup fence would drain vmcnt (stores and younger copies included)
        BG_STAMP(2 + 2 * (cc % C));
        BG_PIN();
        read_w(fr[0], swl + bb[cc % 3] / 4 + s0);
        BG_PIN();
    };

    // ---- prologue of the first slab
    int slab = first;
    int row = slab * 128 + wave * 32 + i;
    negc = a.alternate && (slab & 1);
    sgn = sgnp = negc ? -1.0f : 1.0f;
    BG_STAMP(0);
#ifdef BG_CHAIN_PROBE_STAMPS
    stamps[62] = wall_clock64();
#endif
    static_for<K0 / 8>([&](auto j_) { loadx(row, j_); });
    BG_PIN();
    static_for<S::ndma(0)>([&](auto q_) { dma(IC<0>{}, q_); });
    static_for<S::ndma(1)>([&](auto q_) { dma(IC<1>{}, q_); });
    BG_PIN();
    static_for<NT1>([&](auto t_) { static_for<4>([&](auto g_) { init4(a1, 0, negc, t_, g_); }); });
#pragma unroll
    for (int t = 0; t < NT3; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) a3[t][r] = 0.f;
    vw4(IC<0>{});
    wait_vm<S::ndma(1)>();  // chunk 0 has landed (later slabs: the count of the chunk before covers it)
    {
        SplitTmp st;
        static_for<4>([&](auto p_) {
            constexpr int p = decltype(p_)::value;
            static_for<4>([&](auto ph_) { split_pair_at(ph_, x0[2 * p], x0[2 * p + 1], st, p_); });
        });
#pragma unroll
        for (int q = 0; q < 3; q++) xp[q] = u32x4{xn[q][0], xn[q][1], xn[q][2], xn[q][3]};
    }
    chunk_top(IC<0>{});
    int rowp = row;        // the slab whose layer 3 is finished during this slab's layers 1 / 2 (first slab: itself -- rewritten later with the real values)
    bool has_prev = false;

    // One layer: chunks base .. base + K / 32 - 1 of the stream.  xin(s): the lane's s'th input value (k-step J takes values 8 J .. 8 J + 7).
    // What rides behind MFMA g of a tile-step: 0: the next fragments, a copy; 1, 2 and 7, 8: one element of an epilogue each (ELU in two pieces),
    // stores behind 8; 3 .. 6: the split of one pair; 5: a load; 6: a second copy.  Four issue slots per gap are free.
    auto layer = [&](auto L_, auto& acc, auto xin, auto K_, auto N_, auto base_, int rown) {
        constexpr int L = decltype(L_)::value, K = decltype(K_)::value, N = decltype(N_)::value, NT = N / 32, CH = K / 32, base = decltype(base_)::value;
        constexpr int EPT = 8 / NT;  // elements of the tile below finished per tile-step (8 per k-step)
        static_for<CH>([&](auto kc_) {
            constexpr int kc = decltype(kc_)::value, c = base + kc, PP = S::pp(c);
            const unsigned* sw = swl + bb[c % 3] / 4;
            static_for<2>([&](auto j_) {
                constexpr int j = decltype(j_)::value, J = 2 * kc + j;
                constexpr bool lastk = (J == K / 16 - 1);
                static_for<NT>([&](auto t_) {
                    constexpr int t = decltype(t_)::value, ts = j * NT + t;
                    SplitTmp st0, st1;
                    FinTmp f0, f1;
                    if constexpr (ts == 2 * NT - 1) chunk_top(IC<c + 1>{});  // (reads the next chunk's first fragments into fr[0]; this tile-step's are in fr[1])
                    mfma9(acc[t], fr[ts & 1].p, xp, [&](auto g_) {
                        constexpr int g = decltype(g_)::value;
                        // the next tile-step's weight fragments
                        if constexpr (g == 0 && ts + 1 < 2 * NT) read_w(fr[(ts + 1) & 1], sw + ((ts + 1) / NT ? s1 : s0) + ((ts + 1) % NT) * 32 * SP_ROW);
                        // the copies of chunk c + 2: PP per tile-step from the chunk's first tile-step on
                        if constexpr (g == 0) dma(IC<c + AHEAD>{}, IC<ts * PP>{});
                        if constexpr (g == 6 && PP >= 2) dma(IC<c + AHEAD>{}, IC<ts * PP + 1>{});
                        static_assert(PP <= 2, "");
                        if constexpr (!lastk) {
                            // the planes of k-step J + 1: one pair per NT / 4 tile-steps, pieces behind MFMAs 3 .. 6
                            if constexpr (t % (NT / 4) == 0 && g >= 3 && g <= 6) {
                                constexpr int p = t / (NT / 4), s = 8 * (J + 1) + 2 * p;
                                split_pair_at(IC<g - 3>{}, xin(IC<s>{}), xin(IC<s + 1>{}), st0, IC<p>{});
                            }
                        }
                        if constexpr (L == 1 && J <= 2) {
                            // layer 3 of the slab before: elements 0 .. 47, two per tile-step
                            constexpr int e0 = 2 * (J * NT + t);
                            if constexpr (g == 1) l3_a(f0, IC<e0>{});
                            if constexpr (g == 2) l3_b(f0, IC<e0>{}, rowp);
                            if constexpr (g == 7) l3_a(f1, IC<e0 + 1>{});
                            if constexpr (g == 8) l3_b(f1, IC<e0 + 1>{}, rowp);
                        }
                        if constexpr (L == 2 && kc * 2 * NT + ts < 16) {
                            // ... elements 48 .. 63, one per tile-step, behind MFMAs the split leaves lightly used
                            constexpr int e = 48 + kc * 2 * NT + ts;
                            if constexpr (g == 0) l3_a(f1, IC<e>{});
                            if constexpr (g == 6) l3_b(f1, IC<e>{}, rowp);
                        }
                        if constexpr (L >= 2 && !lastk) {
                            // tile kc + 1 of the layer below: elements 8 j .. 8 j + 7 during this k-step, stored by fours
                            constexpr int NTP = K / 32;
                            auto& prev = [&]() -> auto& { if constexpr (L == 2) return a1; else return a2; }();
                            float* __restrict__ Yprev = L == 2 ? a.Y1 : a.Y2;
                            if constexpr (kc + 1 < NTP) {
                                if constexpr (EPT == 1) {
                                    if constexpr (g == 1) fa(f0, prev, IC<kc + 1>{}, IC<8 * j + t>{}, sgn);
                                    if constexpr (g == 2) fb(f0, prev, IC<kc + 1>{}, IC<8 * j + t>{});
                                    if constexpr (g == 8 && (t & 3) == 3) store4(prev, Yprev, row, IC<K>{}, IC<kc + 1>{}, IC<(8 * j + t) / 4>{});
                                } else {
                                    if constexpr (g == 1) fa(f0, prev, IC<kc + 1>{}, IC<8 * j + 2 * t>{}, sgn);
                                    if constexpr (g == 2) fb(f0, prev, IC<kc + 1>{}, IC<8 * j + 2 * t>{});
                                    if constexpr (g == 7) fa(f1, prev, IC<kc + 1>{}, IC<8 * j + 2 * t + 1>{}, sgn);
                                    if constexpr (g == 8) fb(f1, prev, IC<kc + 1>{}, IC<8 * j + 2 * t + 1>{});
                                    if constexpr (g == 8 && (t & 1) == 1) store4(prev, Yprev, row, IC<K>{}, IC<kc + 1>{}, IC<(8 * j + 2 * t) / 4>{});
                                }
                            }
                        }
                        if constexpr (L == 3) {
                            // the next slab's input rows (the registers of the layer-1 input are free), one 16-byte load per tile-step of the first chunk
                            if constexpr (kc == 0 && g == 5) loadx(rown, IC<ts>{});
                            // ... and its layer-1 accumulators = bias, over the last two chunks
                            if constexpr (kc >= CH - 2 && (g == 3 || g == 4)) {
                                constexpr int G = ((kc - (CH - 2)) * 2 * NT + ts) * 2 + (g == 4);
                                init4(a1, 0, negn, IC<G / 4>{}, IC<G % 4>{});
                            }
                        }
                        if constexpr (lastk && L < 3) {
                            // the layer's last k-step: tile 0 is complete behind tile-step 0; it is finished under the other tiles, then the planes of the
                            // NEXT layer's first k-step (its elements 0 .. 7) are split; the next layer's accumulators = bias
                            float* __restrict__ Y = L == 1 ? a.Y1 : a.Y2;
                            if constexpr (L == 1) { if constexpr (g == 2 || g == 4 || g == 6 || g == 8) { constexpr int G = 4 * t + (g - 2) / 2; if constexpr (G < 4 * NT2) init4(a2, N1, negc, IC<G / 4>{}, IC<G % 4>{}); } }
                            if constexpr (L == 2) { if constexpr (g == 2 || g == 4 || g == 6 || g == 8) { constexpr int G = 4 * t + (g - 2) / 2; if constexpr (G < 4 * NT3) init4(a3, N1 + N2, negc, IC<G / 4>{}, IC<G % 4>{}); } }
                            if constexpr (NT == 8) {
                                if constexpr (t >= 1 && t <= 4 && g >= 1) {
                                    constexpr int r = 4 * (t - 1) + (g - 1) / 2;
                                    if constexpr (g & 1) fa(f0, acc, IC<0>{}, IC<r>{}, sgn); else fb(f0, acc, IC<0>{}, IC<r>{});
                                    if constexpr (g == 8) store4(acc, Y, row, IC<N>{}, IC<0>{}, IC<t - 1>{});
                                }
                                if constexpr (t == 5 || t == 6) {
                                    constexpr int p = 2 * (t - 5);
                                    if constexpr (g <= 3) split_pair_at(IC<g>{}, acc[0][2 * p], acc[0][2 * p + 1], st0, IC<p>{});
                                    else if constexpr (g <= 7) split_pair_at(IC<g - 4>{}, acc[0][2 * p + 2], acc[0][2 * p + 3], st1, IC<p + 1>{});
                                }
                            } else {
                                if constexpr (t == 1 || t == 2) {
                                    // eight elements: piece a of element g behind MFMA g, piece b behind the next (two temporaries in turn)
                                    if constexpr (g >= 1) { if constexpr ((g - 1) & 1) fb(f1, acc, IC<0>{}, IC<8 * (t - 1) + g - 1>{}); else fb(f0, acc, IC<0>{}, IC<8 * (t - 1) + g - 1>{}); }
                                    if constexpr (g <= 7) { if constexpr (g & 1) fa(f1, acc, IC<0>{}, IC<8 * (t - 1) + g>{}, sgn); else fa(f0, acc, IC<0>{}, IC<8 * (t - 1) + g>{}, sgn); }
                                    if constexpr (g == 4) store4(acc, Y, row, IC<N>{}, IC<0>{}, IC<2 * (t - 1)>{});
                                    if constexpr (g == 8) store4(acc, Y, row, IC<N>{}, IC<0>{}, IC<2 * (t - 1) + 1>{});
                                }
                                if constexpr (t == 3 && g <= 7) {
                                    constexpr int p = g / 2;
                                    split_pair_at(IC<2 * (g & 1)>{}, acc[0][2 * p], acc[0][2 * p + 1], st0, IC<p>{});
                                    split_pair_at(IC<2 * (g & 1) + 1>{}, acc[0][2 * p], acc[0][2 * p + 1], st0, IC<p>{});
                                }
                            }
                        }
                        if constexpr (lastk && L == 3) {
                            // the planes of the next slab's first k-step, from its input rows
                            if constexpr (t == NT - 1 && g <= 7) {
                                constexpr int p = g / 2;
                                split_pair_at(IC<2 * (g & 1)>{}, x0[2 * p], x0[2 * p + 1], st0, IC<p>{});
                                split_pair_at(IC<2 * (g & 1) + 1>{}, x0[2 * p], x0[2 * p + 1], st0, IC<p>{});
                            }
                        }
                    });
                    if constexpr (L == 2 && kc * 2 * NT + ts == 15) {
                        // the slab before is complete: its value (the two halves of a sample's features sit in lanes i and i + 32)
                        part += __shfl_xor(part, 32);
                        if (a.v_out && has_prev && h == 0 && rowp < a.M) a.v_out[rowp] = part + vbias;
                        part = 0.f;
                        vw4(IC<0>{});
                    }
                });
#pragma unroll
                for (int q = 0; q < 3; q++) xp[q] = u32x4{xn[q][0], xn[q][1], xn[q][2], xn[q][3]};
            });
        });
    };
    auto x0in = [&](auto s_) { return x0[decltype(s_)::value]; };
    auto a1in = [&](auto s_) { constexpr int s = decltype(s_)::value; return a1[s >> 4][s & 15]; };
    auto a2in = [&](auto s_) { constexpr int s = decltype(s_)::value; return a2[s >> 4][s & 15]; };
    for (;;) {
        const int next = slab + stride;
        const bool has_next = next < nslabs;
        const int rown = (has_next ? next : slab) * 128 + wave * 32 + i;
        negn = a.alternate && ((has_next ? next : slab) & 1);
        layer(IC<1>{}, a1, x0in, IC<K0>{}, IC<N1>{}, IC<0>{}, rown);
        layer(IC<2>{}, a2, a1in, IC<N1>{}, IC<N2>{}, IC<C0>{}, rown);
        layer(IC<3>{}, a3, a2in, IC<N2>{}, IC<N3>{}, IC<C0 + C1>{}, rown);
        // the stream's phase in the three LDS buffers advances by C chunks
        if constexpr (C % 3 == 1) { const unsigned b0 = bb[0]; bb[0] = bb[1]; bb[1] = bb[2]; bb[2] = b0; }
        if constexpr (C % 3 == 2) { const unsigned b0 = bb[0]; bb[0] = bb[2]; bb[2] = bb[1]; bb[1] = b0; }
        rowp = row;
        sgnp = sgn;
        has_prev = true;
        if (!has_next) break;
        slab = next;
        row = rown;
        negc = negn;
        sgn = negc ? -1.0f : 1.0f;
    }
    // layer 3 of the last slab
    static_for<16 * NT3>([&](auto e_) {
        FinTmp f;
        l3_a(f, e_);
        l3_b(f, e_, rowp);
    });
    part += __shfl_xor(part, 32);
    if (a.v_out && h == 0 && rowp < a.M) a.v_out[rowp] = part + vbias;
    wait_vm<0>();  // the copies issued for a slab that does not exist must have landed before the workgroup's LDS is handed on
#ifdef BG_CHAIN_PROBE_STAMPS
    stamps[1 + 2 * C] = clock64();
    stamps[63] = wall_clock64();
    if (lane == 0 && blockIdx.x < 256)
        for (int k = 0; k < 64; k++) bg_split_stamp_buf[(((size_t)(N2 == 256) * 256 + blockIdx.x) * 4 + wave) * 64 + k] = (k <= 1 + 2 * C || k >= 62) ? stamps[k] : 0;
#endif
}

// TAG: 1 / 2 = one network with N2 = 128 / 256 (one kernel symbol per shape: a profiler's per-kernel average is the average of ONE shape),
// 0 = a group, shapes looked up per workgroup; 3 = the same code as 0 under another symbol, for groups WITHOUT persistent workgroups (the few slabs per
// network the rollout evaluates as they appear: 40-60 us launches that would otherwise be averaged with the update's 260 us launches).
template <int TAG>
__global__ __launch_bounds__(256) void mlp_chain_split_fwd_kernel(SplitGroup grp) {
    __shared__ __attribute__((aligned(16))) unsigned sW[NBUF * BUFDW];
    __shared__ __attribute__((aligned(16))) float sB[2 * SBNEG];   // biases [N1 | N2 | N3], value-head weights [N3]; SBNEG further: the negated biases
    int k = 0;
    if constexpr (TAG == 0 || TAG == 3) {
#pragma unroll
        for (int j = 1; j < CHAIN_MAX; j++)
            if (j < grp.n && (int)blockIdx.x >= grp.begin[j]) k = j;
    }
    const bg_mlp_chain_split& a = grp.net[k];
    const int nb = a.N1 + a.N2 + a.N3;
    for (int j = threadIdx.x; j < nb + a.N3; j += 256) {
        const float v = j < a.N1 ? a.b1[j] : j < a.N1 + a.N2 ? a.b2[j - a.N1] : j < nb ? a.b3[j - a.N1 - a.N2] : a.v_w ? a.v_w[j - nb] : 0.f;
        sB[j] = v;
        if (j < nb) sB[SBNEG + j] = -v;
    }
    __syncthreads();
    // One slab per workgroup, or (workgroups > 0) that many workgroups walking the network's slabs (see bg_mlp_chain.hip: two launches side by side
    // share the chip by CUs; counts that are multiples of the 8 XCDs)
    const int nslabs = (a.M + 127) / 128, stride = grp.begin[k + 1] - grp.begin[k], first = blockIdx.x - grp.begin[k];
    if (first >= nslabs) return;
    if (TAG == 2 || ((TAG == 0 || TAG == 3) && a.N2 == 256)) split_net<64, 256, 256, 128>(a, first, stride, nslabs, sW, sB);
    else split_net<64, 256, 128, 128>(a, first, stride, nslabs, sW, sB);
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
    bool persistent = false;
    for (int k = 0; k < count; k++) {
        const int rc = split_check(nets[k]);
        if (rc) return rc;
        grp.begin[k] = blocks;
        grp.net[k] = nets[k];
        const int slabs = (nets[k].M + 127) / 128;
        persistent = persistent || (nets[k].workgroups > 0 && nets[k].workgroups < slabs);
        blocks += nets[k].workgroups > 0 && nets[k].workgroups < slabs ? nets[k].workgroups : slabs;
    }
    grp.begin[count] = blocks;
    if (count > 1 && persistent) hipLaunchKernelGGL(mlp_chain_split_fwd_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    else if (count > 1) hipLaunchKernelGGL(mlp_chain_split_fwd_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    else if (nets[0].N2 == 256) hipLaunchKernelGGL(mlp_chain_split_fwd_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    else hipLaunchKernelGGL(mlp_chain_split_fwd_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    if (hipGetLastError() != hipSuccess) return bg_set_error(-2, "bg_mlp_chain_forward_split: launch failed");
    return 0;
}
