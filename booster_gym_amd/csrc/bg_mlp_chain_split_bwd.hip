// The backward-data chain of one network of the PPO update (reference utils/model.py:9-26 under `loss.backward()`, utils/runner.py:163): from dL/dz of
// the last hidden layer down to dL/dz of the first, as ONE kernel per 128-row slab on the bf16 matrix pipe with fp32 semantics, gfx950 only:
//     G2 = (G3 W3) * elu'(A2),   G1 = (G2 W2) * elu'(A1),   column sums of G2 and G1 (the bias gradients of layers 2 and 1)
// (W3: [N3][N2], W2: [N2][N1] row-major as torch's Linear; A2 / A1: the layers' stored outputs; elu'(a) = 1 for a > 0, a + 1 otherwise).
//   * Arithmetic as bg_mlp_chain_split.hip: every fp32 operand the exact sum of three bf16 numbers, all 9 cross products in the fp32 accumulator.
//   * Transposed products as there (D = W^T G^T: the planes of the TRANSPOSED weights are the A operand, one lane = one sample), but TILE-MAJOR: the
//     inputs of a layer are kept in registers already split (three planes: 1.5 registers per value -- the backward chain has two layers, so this fits
//     where the forward's three would not), a chunk of the weight stream is ONE 32-wide output tile over the whole k range, and a tile is complete when
//     its chunk ends.  Everything that follows a tile -- x elu'(activation), store, column sum, and for G2 the split into the next layer's planes --
//     rides in the MFMA gaps of the NEXT tile's chunk, across layers and across slabs (the last tile of a slab under the first of the next).  The
//     activations a tile's epilogue needs are loaded during the tile's own chunk; the next slab's input rows are loaded and split during layer B.
//   * One wave per SIMD, one workgroup per CU, persistent; weight planes through three 48 KB LDS buffers by global_load_lds_dwordx4 two chunks ahead,
//     one barrier per chunk (in front of the last k-step of the chunk before), copies / loads / stores counted by hand.
//   * The CU's vector-memory pipeline is what this kernel runs against (tools/chain_split_bwd_stamps.py: without its loads and stores a slab takes
//     1.14-1.16 x its MFMA time, with them 2.3 x in the first version and 1.4-1.7 x now): a 16-byte-per-lane access in the accumulator layout touches 32 lines a quarter each.  So the activations
//     under a tile come in as FOUR coalesced LDS-DMA pieces (8 rows x 128 bytes each, full lines, no register stop, swizzled at the source) into a
//     4 KB corner of LDS per wave and are read from there in the accumulator layout a chunk later; the input rows of the next slab are counted asm
//     loads; nothing the compiler would wait for by draining the copies is left in the stream.
//   * Column sums: across the 32 samples of a wave by a transposing DPP butterfly (38 instructions per tile), one record per (slab, wave) in global
//     memory, summed in a fixed order by the reduction the descriptor describes (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/booster_gym_amd.h"

extern int bg_set_error(int code, const char* msg);

#include "bg_chain_split.h"

#ifdef BG_CHAIN_PROBE_STAMPS  // tools/chain_split_bwd_stamps.py: shader-clock stamps of every wave around every chunk barrier (never defined in the product build)
__device__ long long bg_bwd_stamp_buf[2 * 256 * 4 * 64];  // [N2 == 256][workgroup][wave][stamp]
extern "C" int bg_probe_read_bwd_stamps(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(bg_bwd_stamp_buf), bytes); }
#define BG_STAMP(K) stamps[K] = clock64()
#else
#define BG_STAMP(K) do { } while (0)
#endif

namespace {

struct BwdGroup { int n; int begin[CHAIN_MAX + 1]; bg_mlp_chain_split_bwd net[CHAIN_MAX]; };

// DPP helpers: x + (x of the partner lane) for partner maps that pair lanes across one bit of the lane number (both lanes of a pair get the sum), and
// the choice between two registers by 4-lane bank (a v_mov_b32_dpp with a bank mask)
template <int CTRL>
__device__ __forceinline__ float add_partner(float x) {
    return x + __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xF, 0xF, true));
}
template <int BANKS>
__device__ __forceinline__ float pick_banks(float keep, float take) {  // `take` in the lanes of the banks in BANKS, `keep` elsewhere
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp((int)__float_as_uint(keep), (int)__float_as_uint(take), 0xE4, 0xF, BANKS, false));
}
// Column sums of a finished tile: 16 registers (register r of lane (i, h): feature (r & 3) + 8 (r >> 2) + 4 h of sample i) -> ONE register in which lane
// i holds the sum over the 32 samples of the half-wave of register R(i) = 8 b4 + 4 b3 + 2 b2 + b1 (b_k: bit k of i; lanes i and i ^ 1 hold the same sum).
// A transposing butterfly: each stage adds across one bit of the lane number and halves the number of registers, the lanes with the bit set keeping
// the upper half -- 16 + 12 + 6 + 3 + 1 = 38 vector instructions for the tile (five DPP adds per ELEMENT otherwise).  OP = 0 .. 38: one instruction each,
// so that they can be dealt over the MFMA gaps.
struct Bfly { float u[8], t1, t2, w; };
template <int OP>
__device__ __forceinline__ void bfly_op(Bfly& b, const f32x16& v, bool odd_pair) {
    if constexpr (OP < 16) {  // lanes 16 .. 31 of each half <-> lanes 0 .. 15: v_permlane16_swap, then one add
        constexpr int r = OP / 2;
        if constexpr ((OP & 1) == 0) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[r]), __float_as_uint(v[r + 8]), false, false);
            b.t1 = __uint_as_float(sw[0]); b.t2 = __uint_as_float(sw[1]);
        } else {
            b.u[r] = b.t1 + b.t2;
        }
    } else if constexpr (OP < 28) {  // bit 3: row_mirror (lane l <-> 15 - l), banks 2, 3 keep the upper registers
        constexpr int k = OP - 16, r = k / 3;
        if constexpr (k % 3 == 0) b.t1 = add_partner<0x140>(b.u[r]);
        if constexpr (k % 3 == 1) b.t2 = add_partner<0x140>(b.u[r + 4]);
        if constexpr (k % 3 == 2) b.u[r] = pick_banks<0xC>(b.t1, b.t2);
    } else if constexpr (OP < 34) {  // bit 2: row_half_mirror (l <-> 7 - l inside 8 lanes), banks 1, 3
        constexpr int k = OP - 28, r = k / 3;
        if constexpr (k % 3 == 0) b.t1 = add_partner<0x141>(b.u[r]);
        if constexpr (k % 3 == 1) b.t2 = add_partner<0x141>(b.u[r + 2]);
        if constexpr (k % 3 == 2) b.u[r] = pick_banks<0xA>(b.t1, b.t2);
    } else if constexpr (OP == 34) b.t1 = add_partner<0x1B>(b.u[0]);   // bit 1: quad_perm [3, 2, 1, 0]
    else if constexpr (OP == 35) b.t2 = add_partner<0x1B>(b.u[1]);
    else if constexpr (OP == 36) b.w = odd_pair ? b.t2 : b.t1;         // (lanes with bit 1 set)
    else if constexpr (OP == 37) b.w = add_partner<0xB1>(b.w);         // bit 0: quad_perm [1, 0, 3, 2]
}

// The products with operands where the register allocator would not put them by itself: layer B's input planes live in ACCUMULATOR registers (192 of
// them: together with layer A's they do not fit the 256 architectural ones, and handed to the compiler's MFMA they are copied to VGPRs before every use)
// and are read from there by the MFMA (gfx90a+: A / B operands may be AGPRs).  The compiler does not know that the statement is an MFMA: where it
// moves an operand into place with a v_accvgpr_write / v_accvgpr_mov just in front of it (it does, 61 times in the critic's kernel), the two wait
// states a vector write needs before an MFMA reads the register are missing.  An s_nop inside every statement costs 12 of a gap's 32 cycles with the
// compiler's own s_nop between adjacent statements (20 k cycles per slab: the gaps are issue-bound where they carry copies or loads); instead the planes
// are pinned to their register class where they are ASSIGNED (BG_PLANES_IN_PLACE), so that no move is left for the compiler to place, and
// tools/isa_hazard_scan.py checks the shipped assembly statement by statement (tests/test_host_logic.py).
#define BG_HAZARD_NOP ""
// BG_PLANES_IN_PLACE(constraint, x): the planes are put into the register class the MFMAs read them from AT THEIR ASSIGNMENT (an empty statement with
// an "a" / "v" operand), far from the products -- then no operand move stands in front of an MFMA (BG_ABL_NOPIN: without, for the test of the check)
#ifdef BG_ABL_NOPIN
#define BG_PLANES_IN_PLACE(C, X) do { } while (0)
#else
#define BG_PLANES_IN_PLACE(C, X) asm volatile("" : C(X))
#endif
template <bool XA>
__device__ __forceinline__ void mfma_asm(f32x16& acc, const u32x4& w, const u32x4& x) {
    if constexpr (XA) asm volatile(BG_HAZARD_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "a"(x));
    else asm volatile(BG_HAZARD_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));
}
template <bool XA>
__device__ __forceinline__ void mfma_asm_first(f32x16& acc, const u32x4& w, const u32x4& x) {
    if constexpr (XA) asm volatile(BG_HAZARD_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "a"(x));
    else asm volatile(BG_HAZARD_NOP "v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(x));
}
// one tile's nine products of a k-step (small terms first), fill(gap) behind MFMA number gap
template <bool XA, bool FIRST, class F>
__device__ __forceinline__ void mfma9_asm(f32x16& acc, const u32x4 (&w)[3], const u32x4 (&x)[3], F&& fill) {
    if constexpr (FIRST) mfma_asm_first<XA>(acc, w[2], x[2]); else mfma_asm<XA>(acc, w[2], x[2]);
    fill(IC<0>{}); BG_PIN();
    mfma_asm<XA>(acc, w[1], x[2]); fill(IC<1>{}); BG_PIN();
    mfma_asm<XA>(acc, w[2], x[1]); fill(IC<2>{}); BG_PIN();
    mfma_asm<XA>(acc, w[0], x[2]); fill(IC<3>{}); BG_PIN();
    mfma_asm<XA>(acc, w[1], x[1]); fill(IC<4>{}); BG_PIN();
    mfma_asm<XA>(acc, w[2], x[0]); fill(IC<5>{}); BG_PIN();
    mfma_asm<XA>(acc, w[0], x[1]); fill(IC<6>{}); BG_PIN();
    mfma_asm<XA>(acc, w[1], x[0]); fill(IC<7>{}); BG_PIN();
    mfma_asm<XA>(acc, w[0], x[0]); fill(IC<8>{}); BG_PIN();
}

// All slabs first, first + stride, ... < nslabs of one network.  Stream: chunk c < TA: tile c of layer A (G3 -> G2), TA <= c < C: tile c - TA of
// layer B (G2 -> G1).  sT: 4 KB of LDS per wave.
template <int N1, int N2, int N3>
__device__ __forceinline__ void bwd_net(const bg_mlp_chain_split_bwd& a, int first, int stride, int nslabs, unsigned* sW, float* sT) {
    constexpr int KA = N3, KB = N2, TA = N2 / 32, TB = N1 / 32, C = TA + TB;
    constexpr int JA = KA / 16, JB = KB / 16, CHA = KA / 32, CHB = KB / 32;
    static_assert(N3 == 128 && N1 == 256 && (N2 == 128 || N2 == 256), "the reference's widths");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    unsigned rowpart[3], piecepart[3];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const int sp = m * 64 + lane, nl = sp / 12, sig = sp % 12;
        rowpart[m] = (unsigned)nl * 192u;
        piecepart[m] = (unsigned)(((sig & ~3) | ((sig & 3) ^ ((nl >> 2) & 3))) * 16);
    }
    const unsigned* __restrict__ PA = reinterpret_cast<const unsigned*>(a.PT3);
    const unsigned* __restrict__ PB = reinterpret_cast<const unsigned*>(a.PT2);
    // Per chunk and wave: the copies issued during it (those of chunk cc + 2), and what every lane issues during it: the four stores of the tile
    // before, the four loads of this tile's activations, and during eight chunks of layer B two loads of the next slab's input rows.
    struct S {
        static constexpr bool isA(int cc) { return (cc % C) < TA; }
        static constexpr int ndma(int cc) { return (isA(cc) ? CHA : CHB) * 3 / 2; }
        static constexpr int ksteps(int cc) { return isA(cc) ? JA : JB; }
        // copies per k-step of chunk cc (those of chunk cc + 2), dealt from the chunk's first k-step on, none in the last (the next chunk's top sits there)
        static constexpr int pp(int cc) { return (ndma(cc + 2) + ksteps(cc) - 2) / (ksteps(cc) - 1); }
        // the next slab's 16 input loads: two per chunk over the 8 chunks that end with the slab's last but one (planes J are split in the chunk after)
        static constexpr int xfirst() { return C - 1 - JA; }
        static constexpr int xloads_in(int cc) { return (cc >= xfirst() && cc < xfirst() + JA) ? 2 : 0; }
        static constexpr int ops_in(int cc) { return ndma(cc + 2) + 4 + 4 + xloads_in(cc); }
        // Where a chunk's vector-memory operations go -- at most two or three per k-step, none in the last (the CU's four waves run in step: eight in
        // one k-step fill the memory pipeline's queue and all four stall at their next one): the copies of chunk cc + 2, pp() per k-step from the
        // first; two pieces of the tile's activations in each of k-steps 0 and 1 (under load a piece takes ~2,000 cycles to land and to retire); the two loads of the next slab's input in k-step 0; the four stores of
        // the tile before in k-steps ksteps / 2 - 1 .. ksteps / 2 + 2 (group k of its elements is finished by then).
        static constexpr int store_first(int cc) { return ksteps(cc) / 2 - 1; }
        // operations of chunk cc issued BEHIND the last one the next chunk's top must wait for (the activations' pieces of k-step 1): the copies of
        // k-steps >= 2 and the stores
        static constexpr int late_ops(int cc) {
            int n = 4;   // (the stores: k-steps >= 3)
            for (int J = 2; J < ksteps(cc) - 1; J++)
                for (int k = 0; k < pp(cc); k++) n += (J * pp(cc) + k < ndma(cc + 2)) ? 1 : 0;
            return n;
        }
        // the barrier that publishes chunk cc stands in front of the LAST k-step of chunk cc - 1, which issues nothing: exactly late_ops(cc - 1)
        // operations are younger than the last piece of the activations (and than the copies of chunk cc, issued during chunk cc - 2); one for safety
        static constexpr int behind(int cc) { const int n = late_ops((cc + C - 1) % C) - 1; return n > 0 ? n : 0; }
    };
    static_assert(S::xfirst() >= TA - 1 && JA == 8, "the next slab's planes are written when layer A of this slab is over");
    unsigned bb[3] = {0u, (unsigned)BUFDW * 4u, 2u * (unsigned)BUFDW * 4u};
    const unsigned sWbase = (unsigned)(uintptr_t)sW;
    // Odd slabs accumulate the negated sums when a.alternate is set (see include/booster_gym_amd.h: the MFMA accumulator's rounding bias then cancels in
    // everything summed over rows); sgn / sgnp: the sign that puts a finished tile of the slab in work / of the slab before right again
    bool negc = false, negn = false;
    float sgn = 1.0f, sgnp = 1.0f;
    auto dma = [&](auto cc_, auto q_) {
        constexpr int cc = decltype(cc_)::value, q = decltype(q_)::value, c = cc % C;
        if constexpr (q < S::ndma(cc)) {
            const unsigned dst = sWbase + bb[cc % 3];
            const bool neg = cc < C ? negc : negn;   // (the slab the chunk belongs to: this one or the next; odd slabs read the planes of -W^T)
            if constexpr (c < TA) dma_tile_piece<q>(PA + (neg ? (size_t)N2 * N3 * 3 / 2 : 0), CHA, c, dst, wave, rowpart, piecepart);
            else dma_tile_piece<q>(PB + (neg ? (size_t)N1 * N2 * 3 / 2 : 0), CHB, c - TA, dst, wave, rowpart, piecepart);
        }
    };
    const int sx = (i >> 2) & 3;              // this lane's slot swizzle
    const unsigned* swl = sW + i * SP_ROW;    // + buffer, + k-chunk * 32 rows, + slot
    // this wave's 4 KB of LDS for the activations under a tile: [32 rows][8 pieces of 16 bytes], piece P of row r at slot P ^ (r & 7)
    const unsigned sTw = (unsigned)(uintptr_t)sT + (unsigned)wave * 4096u;
    const float* sTr = sT + wave * 1024 + i * 32;   // this lane's row
    // lane l of an activation piece: row l >> 3 of the piece's eight, source piece (l & 7) ^ (row & 7): byte offset in rows of ld floats
    const unsigned aofsA = (unsigned)((lane >> 3) * N2 * 4 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));
    const unsigned aofsB = (unsigned)((lane >> 3) * N1 * 4 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));
    // the feature of a tile whose column sum the butterfly leaves in this lane: register R = 8 b4 + 4 b3 + 2 b2 + b1 of the lane number's bits
    const int csR = ((i >> 4) & 1) * 8 + ((i >> 3) & 1) * 4 + ((i >> 2) & 1) * 2 + ((i >> 1) & 1), csofs = (csR & 3) + 8 * (csR >> 2) + 4 * h;

    u32x4 gp[JA][3];   // planes of the slab's input rows (G3), k-step J: values 8 J .. 8 J + 7 of the lane
    u32x4 hp[JB][3];   // planes of G2, filled as layer A's tiles are finished
    f32x16 acc[2];     // the tile in work and the tile being finished
    f32x4 aux[4];      // the activations (layer outputs) under the tile being finished, read from this wave's LDS corner at the top of the chunk
    f32x4 xin[2][2] = {};  // two 16-byte pieces of the next slab's input row on their way into gp: loaded in one chunk, split in the next
    Frag fr[2];
    unsigned pn[3][4] = {}, pn2[3][4] = {};
#ifdef BG_CHAIN_PROBE_STAMPS
    long long stamps[64] = {};
#endif
    // input rows: 16-byte piece j of the lane = floats 8 j + 4 h .. + 3; rows >= M are ZERO (then so is everything computed from them: G2, G1 rows and
    // their share of the column sums)
    // Prologue loads of the first slab's input rows (plain loads: the compiler's waits are fine there).  (The select for rows >= M is applied where
    // the values are USED, in the split: a select directly behind a load makes the compiler wait for the load in place.)
    auto loadx = [&](int r, auto j_, f32x4& dst) {
        constexpr int j = decltype(j_)::value;
        dst = *reinterpret_cast<const f32x4*>(a.G3 + (size_t)(r < a.M ? r : a.M - 1) * N3 + 4 * h + 8 * j);
    };
    // ... inside the stream: a counted asm load (the compiler does not know it, does not wait for it -- and does not drain the copies for it)
    auto loadx_asm = [&](const float* rowptr, auto j_, f32x4& dst) {
        constexpr int j = decltype(j_)::value;
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(rowptr), "i"(32 * j) : "memory");
    };
    // the activations under tile `tile` of rows row0 .. row0 + 31 (this wave's) -> this wave's LDS corner: four pieces of eight full 128-byte rows
    auto aux_dma = [&](const float* __restrict__ A, int ld, unsigned lofs, int row0, int tile, auto k_) {
        constexpr int k = decltype(k_)::value;
#ifndef BG_ABL_NOAUX
        const float* base = A + (size_t)(row0 + 8 * k) * ld + 32 * tile;
        const unsigned lds = sTw + 1024u * k;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lofs), "s"(base), "s"(lds) : "memory");
#endif
    };
    auto store4 = [&](const f32x16& A, float* __restrict__ Y, int ld, int r, int tile, auto g_) {
#ifndef BG_ABL_NOSTORE
        constexpr int g = decltype(g_)::value;
        const f32x4 v = {A[4 * g + 0], A[4 * g + 1], A[4 * g + 2], A[4 * g + 3]};
        *reinterpret_cast<f32x4*>(Y + (size_t)r * ld + 32 * tile + 8 * g + 4 * h) = v;
#endif
    };
    auto chunk_top = [&](auto cc_) {
        constexpr int cc = decltype(cc_)::value;
        BG_PIN();
        BG_STAMP(1 + 3 * (cc % C));
        wait_vm<S::behind(cc)>();
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): this wave's fragment reads of the chunk before are in its registers
        asm volatile("s_barrier" ::: "memory");  // no fence (it would drain vmcnt)
        BG_STAMP(2 + 3 * (cc % C));
        BG_PIN();
        read_w(fr[0], swl + bb[cc % 3] / 4 + ((0 + h) ^ sx) * 4);
        // the activations under the tile that chunk cc finishes (copied during chunk cc - 1): piece 2 g + h of this lane's row
#pragma unroll
        for (int g = 0; g < 4; g++) aux[g] = *reinterpret_cast<const f32x4*>(sTr + (((2 * g + h) ^ (i & 7)) << 2));
        // the asm loads of chunk cc - 1 have landed (counted wait above): from here on their registers hold the values
        asm volatile("" : "+v"(xin[(cc + 1) & 1][0]), "+v"(xin[(cc + 1) & 1][1]));
        BG_STAMP(3 + 3 * (cc % C));
        BG_PIN();
    };

#ifdef BG_CHAIN_PROBE_STAMPS
    stamps[62] = wall_clock64();
#endif
    BG_STAMP(0);
    // ---- prologue of the first slab: its input rows, split; the first two chunks' copies
    int slab = first;
    int row = slab * 128 + wave * 32 + i;
    negc = a.alternate && (slab & 1);
    sgn = sgnp = negc ? -1.0f : 1.0f;
    static_for<S::ndma(0)>([&](auto q_) { dma(IC<0>{}, q_); });
    static_for<S::ndma(1)>([&](auto q_) { dma(IC<1>{}, q_); });
    BG_PIN();
    static_for<JA>([&](auto J_) {
        constexpr int J = decltype(J_)::value;
        f32x4 x0, x1;
        loadx(row, IC<2 * J>{}, x0);
        loadx(row, IC<2 * J + 1>{}, x1);
        SplitTmp st;
        const bool ok = row < a.M;
        const float v[8] = {ok ? x0.x : 0.f, ok ? x0.y : 0.f, ok ? x0.z : 0.f, ok ? x0.w : 0.f, ok ? x1.x : 0.f, ok ? x1.y : 0.f, ok ? x1.z : 0.f, ok ? x1.w : 0.f};
        static_for<4>([&](auto p_) {
            constexpr int p = decltype(p_)::value;
            static_for<4>([&](auto ph_) { split_phase<decltype(ph_)::value>(v[2 * p], v[2 * p + 1], st, pn[0][p], pn[1][p], pn[2][p]); });
        });
#pragma unroll
        for (int q = 0; q < 3; q++) { gp[J][q] = u32x4{pn[q][0], pn[q][1], pn[q][2], pn[q][3]}; BG_PLANES_IN_PLACE("+v", gp[J][q]); }
    });
    wait_vm<0>();  // (once per workgroup)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[1][r] = 0.f;   // "the tile before" of the first chunk: nothing (zeros: stored to rows that are rewritten, added as zeros)
    // (the first chunk's "tile before" reads this wave's LDS corner before anything was copied there: zeros x whatever is finite -- make it so)
    for (int j = lane; j < 1024; j += 64) sT[wave * 1024 + j] = 1.0f;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    chunk_top(IC<0>{});
    int rowp = row;   // the slab whose last tile is finished under this slab's first chunk (first slab: itself -- zeros, rewritten later)
    const float* xnext = a.G3;   // this lane's part of the next slab's input row
    int row0w = slab * 128 + wave * 32;   // first row of this wave in the slab in work (wave-uniform: the base of a copy is a scalar)
    int slabp = slab;

    // One chunk = one tile: its MFMAs (acc[c & 1]), and in their gaps the tile BEFORE (acc[(c - 1) & 1], aux[(c - 1) & 1]): its 16 elements x
    // elu'(activation) during the FIRST half of the chunk's k-steps (layer B's first tile needs all of G2 in its last k-steps), stored and, for G2, split
    // into layer B's planes there; its column sums (the butterfly above) during the second half.
    // What rides behind MFMA g of a k-step: 0: the next fragments, a copy; 1 .. 4: elements (first half) or butterfly instructions (second half);
    // 1 .. 4 also: the split of a pair of the next slab's input; 4: a second copy; 5: a store; 5 .. 8: the split of G2's pairs; 8: loads.
    auto chunk = [&](auto c_, int rown) {
        constexpr int c = decltype(c_)::value, cp = (c + C - 1) % C;   // cp: the tile being finished
        constexpr bool A = c < TA, pA = cp < TA;
        constexpr int T = A ? c : c - TA, Tp = pA ? cp : cp - TA;       // tile numbers inside their layers
        constexpr int JX = A ? JA : JB, PP = S::pp(c);
        constexpr int JE = JX / 2, EPK = 16 / JE;                       // k-steps that carry the elements; elements per k-step (2 or 4)
        static_assert((C & 1) == 0 && (JX & 1) == 0 && (EPK == 2 || EPK == 4) && S::store_first(c) + 4 <= JX - 1, "two accumulators / fragment buffers in turn");
        const unsigned* sw = swl + bb[c % 3] / 4;
        f32x16& cur = acc[c & 1];
        f32x16& prv = acc[(c + 1) & 1];
        f32x4 (&pa)[4] = aux;   // the activations under the tile being finished (read from LDS at the top of this chunk)
        const int prow = c == 0 ? rowp : row;                      // rows of the tile being finished
        const int prow0w = (c == 0 ? slabp : slab) * 128 + wave * 32;   // ... its first row in this wave (wave-uniform; record (slab, wave) = row / 32)
        float* __restrict__ Yp = pA ? a.G2 : a.G1;
        constexpr int ldp = pA ? N2 : N1;
        float* __restrict__ csp = a.colsum_partial + ((size_t)(prow0w >> 5)) * (N2 + N1) + (pA ? 0 : N2) + 32 * Tp + csofs;   // record (slab, wave)
        Bfly bf;
        static_for<JX>([&](auto J_) {
            constexpr int J = decltype(J_)::value;
            SplitTmp st0, st1, st2;
            float e[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (J == JX - 1) chunk_top(IC<c + 1>{});   // (the next chunk's first fragments -> fr[0]; this k-step's are in fr[1]: JX is even)
            auto fill = [&](auto g_) {
                constexpr int g = decltype(g_)::value;
                // the next k-step's weight fragments; a copy of chunk c + 2
                if constexpr (g == 0 && J + 1 < JX) read_w(fr[(J + 1) & 1], sw + ((J + 1) >> 1) * 32 * SP_ROW + ((2 * ((J + 1) & 1) + h) ^ sx) * 4);
                if constexpr (g == 0) dma(IC<c + AHEAD>{}, IC<J * PP>{});
                if constexpr (g == 4 && PP >= 2) dma(IC<c + AHEAD>{}, IC<J * PP + 1>{});
                if constexpr (g == 7 && PP >= 3) dma(IC<c + AHEAD>{}, IC<J * PP + 2>{});
                static_assert(PP <= 3, "");
                // the activations under this tile (for its epilogue during the next chunk): four coalesced copies into this wave's LDS corner, one in each
                // of the chunk's first two k-steps, two each (behind the reads of the corner at the chunk's top, whose values MFMA 1's element has already used)
                if constexpr (J < 2 && (g == 6 || g == 8)) aux_dma(A ? a.A2 : a.A1, A ? N2 : N1, A ? aofsA : aofsB, row0w, T, IC<2 * J + (g == 8)>{});
                // the next slab's input rows: two 16-byte loads per chunk, split into planes during the chunk after
                // the next slab's input rows: two 16-byte asm loads per chunk (first k-step), split into planes during the chunk after
                if constexpr (c >= S::xfirst() && c < S::xfirst() + JA && J == 0 && (g == 2 || g == 3))
                    loadx_asm(xnext, IC<2 * (c - S::xfirst()) + (g - 2)>{}, xin[c & 1][g - 2]);
                if constexpr (c > S::xfirst() && c <= S::xfirst() + JA && J < 4) {   // (pair J of the two pieces the chunk before loaded)
                    constexpr int Jn = c - S::xfirst() - 1, q4 = J;  // planes gp[Jn] of the next slab (layer A of this slab is over), pair q4
                    static_assert(c <= S::xfirst() || c >= TA, "");
                    const bool okn = rown < a.M;
                    const f32x4 (&xq)[2] = xin[(c + 1) & 1];
                    const float v0 = okn ? xq[q4 >> 1][2 * (q4 & 1)] : 0.f, v1 = okn ? xq[q4 >> 1][2 * (q4 & 1) + 1] : 0.f;
                    if constexpr (g >= 5) split_phase<g - 5>(v0, v1, st2, pn2[0][q4], pn2[1][q4], pn2[2][q4]);
                    if constexpr (q4 == 3 && g == 8) {
#pragma unroll
                        for (int q = 0; q < 3; q++) { gp[Jn][q] = u32x4{pn2[q][0], pn2[q][1], pn2[q][2], pn2[q][3]}; BG_PLANES_IN_PLACE("+v", gp[Jn][q]); }
                    }
                }
                if constexpr (J < JE) {
                    constexpr int r0 = EPK * J;
                    // x elu'(activation): one element behind each of MFMAs 1 .. EPK
                    if constexpr (g >= 1 && g <= EPK) {
                        constexpr int k = g - 1, r = r0 + k;
                        const float av = pa[r / 4][r % 4];
                        const float sg = c == 0 ? sgnp : sgn;
                        e[k] = prv[r] * (av > 0.f ? sg : fmaf(av, sg, sg));
                        prv[r] = e[k];
                    }
                    // G2's elements become planes of layer B: pair p of the tile (elements 2 p, 2 p + 1) is slot p % 4 of k-step 2 Tp + p / 4
                    if constexpr (pA && g >= 5) {
                        split_phase<g - 5>(e[0], e[1], st0, pn[0][(r0 / 2) & 3], pn[1][(r0 / 2) & 3], pn[2][(r0 / 2) & 3]);
                        if constexpr (EPK == 4) split_phase<g - 5>(e[2], e[3], st1, pn[0][(r0 / 2 + 1) & 3], pn[1][(r0 / 2 + 1) & 3], pn[2][(r0 / 2 + 1) & 3]);
                        constexpr int plast = (r0 + EPK - 1) / 2;   // the last pair this k-step completes
                        if constexpr (g == 8 && (plast & 3) == 3) {
#pragma unroll
                            for (int q = 0; q < 3; q++) { hp[2 * Tp + plast / 4][q] = u32x4{pn[q][0], pn[q][1], pn[q][2], pn[q][3]}; BG_PLANES_IN_PLACE("+a", hp[2 * Tp + plast / 4][q]); }
                        }
                    }
                }
                // the finished groups of four go to memory, one per k-step of the second half
                if constexpr (J >= S::store_first(c) && J < S::store_first(c) + 4 && g == 6) store4(prv, Yp, ldp, prow, Tp, IC<J - S::store_first(c)>{});
                if constexpr (J >= JE && g >= 1 && g <= 4) {
                    // the tile's column sums: 38 butterfly instructions over the second half's gaps 1 .. 4, then ONE ds_add per lane pair (lanes i and
                    // i ^ 1 hold the same sum: the odd one adds zero)
                    constexpr int slots = 4 * (JX - JE), per = (39 + slots - 1) / slots, first = ((J - JE) * 4 + (g - 1)) * per;
                    static_for<per>([&](auto k_) {
                        constexpr int op = first + decltype(k_)::value;
                        if constexpr (op < 38) bfly_op<op>(bf, prv, (i & 2) != 0);
                        if constexpr (op == 38) { if ((i & 1) == 0) *csp = bf.w; }   // (lanes i and i ^ 1 hold the same sum)
                    });
                }
            };
#ifdef BG_CHAIN_PROBE_STAMPS   // one chunk of each layer k-step by k-step: stamps 40 .. (layer A, chunk 1), 48 .. (layer B, chunk TA + 2)
            if constexpr (c == 1) BG_STAMP(40 + J);
            if constexpr (c == TA + 2 && J < 14) BG_STAMP(48 + J);
#endif
            const u32x4 (&xpl)[3] = [&]() -> const u32x4 (&)[3] { if constexpr (A) return gp[J]; else return hp[J]; }();
            mfma9_asm<!A, J == 0>(cur, fr[J & 1].p, xpl, fill);
        });
    };
    for (;;) {
        const int next = slab + stride;
        const bool has_next = next < nslabs;
        const int rown = (has_next ? next : slab) * 128 + wave * 32 + i;
        negn = a.alternate && ((has_next ? next : slab) & 1);
        xnext = a.G3 + (size_t)(rown < a.M ? rown : a.M - 1) * N3 + 4 * h;
        row0w = slab * 128 + wave * 32;
        static_for<C>([&](auto c_) { chunk(c_, rown); });
        if constexpr (C % 3 == 1) { const unsigned b0 = bb[0]; bb[0] = bb[1]; bb[1] = bb[2]; bb[2] = b0; }
        if constexpr (C % 3 == 2) { const unsigned b0 = bb[0]; bb[0] = bb[2]; bb[2] = bb[1]; bb[1] = b0; }
        rowp = row;
        slabp = slab;
        sgnp = sgn;
        if (!has_next) break;
        slab = next;
        row = rown;
        negc = negn;
        sgn = negc ? -1.0f : 1.0f;
    }
    // the last tile of the last slab (tile TB - 1 of layer B: acc[(C - 1) & 1], aux[(C - 1) & 1]).  Its last MFMA has just been issued, and the compiler
    // does not know that the statement was one: the wait states a vector read of an MFMA's result needs, by hand (in the loop a whole MFMA and more
    // lie between a tile's last product and the first read of its accumulator).
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[(C - 1) & 1]));  // (tied to the accumulator: a plain statement would not keep the reads behind it)
    {
        f32x16& prv = acc[(C - 1) & 1];
        // its activations: copied into the LDS corner during the last chunk; the chunk_top of the slab that does not exist has read them into aux
        f32x4 (&pa)[4] = aux;
        static_for<16>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const float av = pa[r / 4][r % 4];
            prv[r] = prv[r] * (av > 0.f ? sgnp : fmaf(av, sgnp, sgnp));
            if constexpr ((r & 3) == 3) store4(prv, a.G1, N1, rowp, TB - 1, IC<r / 4>{});
        });
        Bfly bf;
        static_for<38>([&](auto op_) { bfly_op<decltype(op_)::value>(bf, prv, (i & 2) != 0); });
        if ((i & 1) == 0) a.colsum_partial[((size_t)(slabp * 4 + wave)) * (N2 + N1) + N2 + 32 * (TB - 1) + csofs] = bf.w;
    }
    wait_vm<0>();  // the copies issued for a slab that does not exist must have landed before the workgroup's LDS is handed on
#ifdef BG_CHAIN_PROBE_STAMPS
    stamps[1 + 3 * C] = clock64();
    stamps[63] = wall_clock64();
    if (lane == 0 && blockIdx.x < 256)
        for (int k = 0; k < 64; k++) bg_bwd_stamp_buf[(((size_t)(N2 == 256) * 256 + blockIdx.x) * 4 + wave) * 64 + k] = stamps[k];
#endif
}

template <int TAG>
__global__ __launch_bounds__(256) void mlp_chain_split_bwd_kernel(BwdGroup grp) {
    __shared__ __attribute__((aligned(16))) unsigned sW[NBUF * BUFDW];
    __shared__ __attribute__((aligned(16))) float sT[4 * 1024];   // 4 KB per wave: the activations under the tile being finished
    int k = 0;
    if constexpr (TAG == 0) {
#pragma unroll
        for (int j = 1; j < CHAIN_MAX; j++)
            if (j < grp.n && (int)blockIdx.x >= grp.begin[j]) k = j;
    }
    const bg_mlp_chain_split_bwd& a = grp.net[k];
    const int nslabs = (a.M + 127) / 128, stride = grp.begin[k + 1] - grp.begin[k], first = blockIdx.x - grp.begin[k];
    if (first >= nslabs) return;
    if (TAG == 0 && a.N2 == 256) bwd_net<256, 256, 128>(a, first, stride, nslabs, sW, sT);
    else bwd_net<256, 128, 128>(a, first, stride, nslabs, sW, sT);
}

int bwd_check(const bg_mlp_chain_split_bwd& q) {
    if (q.M <= 0 || !q.G3 || !q.PT3 || !q.PT2 || !q.A2 || !q.A1 || !q.G2 || !q.G1 || !q.colsum_partial || !q.bias_grad2 || !q.bias_grad1)
        return bg_set_error(-1, "bg_mlp_chain_backward_split: bad argument");
    if ((((uintptr_t)q.G3 | (uintptr_t)q.PT3 | (uintptr_t)q.PT2 | (uintptr_t)q.A2 | (uintptr_t)q.A1 | (uintptr_t)q.G2 | (uintptr_t)q.G1) & 15) != 0)
        return bg_set_error(-1, "bg_mlp_chain_backward_split: pointers must be 16-byte aligned");
    if (!(q.N1 == 256 && (q.N2 == 128 || q.N2 == 256) && q.N3 == 128)) return bg_set_error(-4, "bg_mlp_chain_backward_split: unsupported widths (256-128-128 and 256-256-128)");
    if (q.workgroups < 0) return bg_set_error(-1, "bg_mlp_chain_backward_split: workgroups < 0");
    return 0;
}

}  // namespace

extern "C" int bg_mlp_chain_backward_split(const bg_mlp_chain_split_bwd* nets, int32_t count, bg_reduce_problem* finishes, void* stream) {
    if (!nets || !finishes || count <= 0 || count > CHAIN_MAX) return bg_set_error(-1, "bg_mlp_chain_backward_split: 1 to 4 networks and their reduction descriptors");
    BwdGroup grp;
    grp.n = count;
    int blocks = 0;
    for (int k = 0; k < count; k++) {
        const int rc = bwd_check(nets[k]);
        if (rc) return rc;
        grp.begin[k] = blocks;
        grp.net[k] = nets[k];
        const int slabs = (nets[k].M + 127) / 128;
        const int wg = nets[k].workgroups > 0 && nets[k].workgroups < slabs ? nets[k].workgroups : slabs;
        blocks += wg;
        memset(&finishes[k], 0, sizeof(finishes[k]));
        finishes[k].partial = nets[k].colsum_partial; finishes[k].groups = slabs * 4; finishes[k].record = nets[k].N2 + nets[k].N1; finishes[k].n_out = nets[k].N2 + nets[k].N1;
        finishes[k].out[0] = nets[k].bias_grad2; finishes[k].n[0] = nets[k].N2;
        finishes[k].out[1] = nets[k].bias_grad1; finishes[k].n[1] = nets[k].N1;
    }
    grp.begin[count] = blocks;
    // (one kernel symbol per shape: a profiler's per-kernel average is the average of ONE shape -- except that the register allocator spills six
    // registers in the 256-256-128 kernel compiled alone and none when both shapes share a kernel, and a spill reload is a vector-memory operation
    // that drains the copies in flight: that shape runs the shared kernel)
    if (count > 1 || nets[0].N2 == 256) hipLaunchKernelGGL(mlp_chain_split_bwd_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    else hipLaunchKernelGGL(mlp_chain_split_bwd_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    if (hipGetLastError() != hipSuccess) return bg_set_error(-2, "bg_mlp_chain_backward_split: launch failed");
    return 0;
}
