// Split-bf16 form of the grouped weight-gradient launch (opt-in, BG_GEMM_SPLIT; the fp32-MFMA form is bg_wgrad.hip), gfx950 only.
// dW[C_out][C_in] = G^T A over the batch (reference utils/runner.py:163 `loss.backward()` through utils/model.py:9-26's Linear layers) with the
// fp32 x fp32 products on the bf16 matrix pipe: every operand is the exact sum of three bf16 numbers (hi / mid / lo, see bg_mlp_split.hip), all 9
// (or the 6 largest) cross products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
//   * The reduction dimension is the batch: a 32x32x16 MFMA takes 16 batch rows per step, lane (i, h) supplies rows 8 h .. 8 h + 7 of output row /
//     column i.  At 9 x 32 cycles per 32 x 32 x 16 block a 128 x 128 tile per wave needs its rows 1.8 x faster than the fp32 form, which already
//     sat on the flop / byte ridge with every wave fetching its own rows: so here the waves of a workgroup that work on the SAME rows (all tiles of
//     a layer: 256 x 256 = 4 tiles, 128 x 256 = 2, ...) share them.  Each 16-row block of G and A (one contiguous span of the row-major arrays) is
//     copied ONCE per workgroup into LDS by global_load_lds_dwordx4, two blocks ahead of its use, and every wave picks its operands out of LDS.
//   * The split runs in the shadow of the MFMAs: while block b is multiplied, the planes of block b + 1 are built two floats at a time (one
//     row pair of one tile: 11 VALU, 2 ds_read_b32) between the MFMAs -- an MFMA holds the issue port 8 of its 32 cycles.  Planes are
//     double-buffered in registers (2 x 96), the 256 accumulators live in the AGPR half of the file.
//   * Split over the batch, in-workgroup reduction of waves that share a tile, one partial tile per workgroup and tile, fixed-order finish:
//     exactly as in bg_wgrad.hip (same scratch layout, same finish kernel).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/booster_gym_amd.h"
#include "bg_wgrad.h"

extern int bg_set_error(int code, const char* msg);
extern int bg_wgrad_group_fill(const bg_wgrad_problem* problems, int32_t count, WgradGroup& grp, int& wg, int& fin, const char* who);
extern int bg_wgrad_group_finish_launch(const WgradGroup& grp, int fin, hipStream_t st);
#define HIP_OK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return bg_set_error(-2, hipGetErrorString(_e));               \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BG_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)
#define BG_PIN() __builtin_amdgcn_sched_barrier(0)

constexpr int WS_BLOCK = 16;                  // batch rows per MFMA step
constexpr int WS_LDS_FLOATS = 36864;          // 144 KB: three buffers of row blocks where they fit (two for the 128 x 128 layer), reused by the reduction (128 KB)
// one block of zeros in device memory: the source of the copies a sub-range asks for past its end (sub-ranges of a workgroup may differ by one pair of
// blocks and all waves keep the same schedule of copies, barriers and MFMAs: the extra blocks multiply zeros)
__device__ float ws_zero_block[WS_BLOCK * 512];

// (x0, x1) -> one dword (low half = x0) of each of the three planes
struct Pair3 { unsigned h, m, l; };
__device__ __forceinline__ Pair3 split_pair3(float x0, float x1) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    Pair3 o;
    o.h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    o.m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    o.l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    return o;
}
__device__ __forceinline__ void set_dword(u32x4 (&pl)[3], int p, const Pair3& v) {  // p is a compile-time constant at every call site
    pl[0][p] = v.h; pl[1][p] = v.m; pl[2][p] = v.l;
}

// One output tile's MFMAs for one 16-row block (small terms first) with NU split units (x[k], y[k]) -> o[k] in their shadow.  The 11 VALU
// instructions of every unit are dealt EVENLY over the gaps behind the MFMAs, units interleaved (dependent instructions land in different gaps):
// measured (tools/probe/mfma_fillers.hip) up to 4 VALU instructions behind a v_mfma_f32_32x32x16_bf16 are free (33.5 cycles per MFMA against 32.8
// alone), 6 cost 52 cycles and 8 cost 62 -- a clump of 8 behind every other MFMA, as a first version had it, ran the loop at half speed.
struct SplitUnit { unsigned u0, u1, t0, t1; float r0, r1; };
// sgn: 0, or the sign bit for the units k < NG (the G operand's) of a sub-range that accumulates the negated sums: the unit then splits -x
template <int NU>
__device__ __forceinline__ void split_op(int o, SplitUnit (&s)[4], const float (&x)[4], const float (&y)[4], Pair3 (&out)[4], int NG, unsigned sgn) {
    const int k = o % NU;  // unit; o / NU: step (compile-time after unrolling)
    SplitUnit& q = s[k];
    switch (o / NU) {
        case 0: {
            const unsigned f = k < NG ? sgn : 0u;
            q.u0 = __float_as_uint(x[k]) ^ f; q.u1 = __float_as_uint(y[k]) ^ f; out[k].h = __builtin_amdgcn_perm(q.u1, q.u0, 0x07060302u);
        } break;
        case 1: q.t0 = q.u0 & 0xffff0000u; break;
        case 2: q.t1 = q.u1 & 0xffff0000u; break;
        case 3: q.r0 = __uint_as_float(q.u0) - __uint_as_float(q.t0); break;
        case 4: q.r1 = __uint_as_float(q.u1) - __uint_as_float(q.t1); break;
        case 5: out[k].m = __builtin_amdgcn_perm(__float_as_uint(q.r1), __float_as_uint(q.r0), 0x07060302u); break;
        case 6: q.t0 = __float_as_uint(q.r0) & 0xffff0000u; break;
        case 7: q.t1 = __float_as_uint(q.r1) & 0xffff0000u; break;
        case 8: q.r0 = q.r0 - __uint_as_float(q.t0); break;
        case 9: q.r1 = q.r1 - __uint_as_float(q.t1); break;
        default: out[k].l = __builtin_amdgcn_perm(__float_as_uint(q.r1), __float_as_uint(q.r0), 0x07060302u); break;
    }
}
template <int TERMS, int NU>
__device__ __forceinline__ void tile_mfmas(f32x16& acc, const u32x4 (&g)[3], const u32x4 (&a)[3], const float (&x)[4], const float (&y)[4], Pair3 (&o)[4], unsigned sgn) {
    // cross terms, small first: (g plane, a plane); the last 6 are the 6-term set
    constexpr int GP[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0}, AP[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};
    constexpr int NOPS = 11 * NU;
    SplitUnit s[4];
#ifdef WS_ABL_NOSPLIT
    for (int k = 0; k < NU; k++) { o[k].h = __float_as_uint(x[k]); o[k].m = __float_as_uint(y[k]); o[k].l = 0u; }
#endif
#pragma unroll
    for (int m = 0; m < TERMS; m++) {
        const int term = 9 - TERMS + m;
#ifndef WS_ABL_NOMFMA  // (probe builds: the loop's copies, reads and split without the matrix work)
        BG_MFMA(acc, g[GP[term]], a[AP[term]]);
#endif
#pragma unroll
        for (int op = (m * NOPS) / TERMS; op < ((m + 1) * NOPS) / TERMS; op++) {
#ifndef WS_ABL_NOSPLIT  // (tools/probe builds: the loop without the split's VALU work -- never defined in the product build)
            split_op<NU == 0 ? 1 : NU>(op, s, x, y, o, NU - 1, sgn);  // (units 0 .. NU - 2: G's; the last: A's)
#endif
        }
        BG_PIN();
    }
}

// A workgroup's 4 waves cover tw output tiles x ks = 4 / tw sub-ranges of the slice's rows (bg_wgrad.hip's wgrad_tile); the tw waves of one
// sub-range share its rows through LDS.  TCI: 32-column C_in tiles per wave (4: 128 input columns, 2: the 64-wide zero-padded first layers).
// Cout / Cin are compile-time: every LDS read of the split then has an immediate offset (with run-time widths the compiler kept one address
// register per (row pair, tile) and the kernel spilled).
template <int Cout, int Cin, int TERMS>
__device__ __forceinline__ void wgrad_split_tile(float* lds, int M, const float* __restrict__ G, const float* __restrict__ A,
                                                 float* __restrict__ P, int ntile_ci, int tile0, int tw, int slice, int slices) {
    constexpr int TCI = Cin == 64 ? 2 : 4;
    constexpr int UPG = 4 / TCI;  // row pairs handled per (t, u) group for each operand: 4 pairs per tile operand over TCI groups
    typedef float avec __attribute__((ext_vector_type(TCI)));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int ks = 4 / tw, tl = wave % tw, ksub = wave / tw, tile = tile0 + tl;
    const int tco = tile / ntile_ci, tci = tile % ntile_ci;
    const long NB2 = M / (2 * WS_BLOCK), W = (long)slices * ks, widx = (long)slice * ks + ksub;  // sub-ranges in units of two blocks
    const int b0 = 2 * (int)(NB2 * widx / W), b1 = 2 * (int)(NB2 * (widx + 1) / W), nb = b1 - b0;
    // Odd sub-ranges accumulate the NEGATED sums (the planes of -G) and hand over the negated accumulators: the bf16 MFMA's accumulator truncates, which
    // leaves ONE negative offset on every element after the thousands of updates of a sub-range (measured -1.27e-6 of an rms error of 1.78e-6:
    // profiles/r06_wgrad_split_error_parts.jsonl); with the sign alternating from sub-range to sub-range the offsets cancel in the sum over the batch
    // and what remains is the random part, which is below the fp32-MFMA kernel's error.  Same products, same exactness.
    const unsigned sgn = (widx & 1) ? 0x80000000u : 0u;
    auto sg = [&](float v) { return __uint_as_float(__float_as_uint(v) ^ sgn); };
    // LDS: [buffer 2][group ks][16 rows x Cout of G, then 16 rows x Cin of A]; a block of G (or A) is one contiguous span of the row-major array
    constexpr int gfl = WS_BLOCK * Cout, afl = WS_BLOCK * Cin, grp_fl = gfl + afl;
    const int buf_fl = ks * grp_fl;
    float* my0 = lds + ksub * grp_fl;  // this group's block in buffer 0 (buffer n: + n * buf_fl)
    // NBUF buffers: the copy of block c + NBUF is started at the top of block c, NBUF - 1 blocks before its first use
    constexpr int NT = (Cout / 128) * (Cin >= 128 ? Cin / 128 : 1), KS = 4 / NT, NBUF = 3 * KS * grp_fl <= WS_LDS_FLOATS ? 3 : 2;  // (four and five buffers in 160 KB: measured no faster, round 6)
    // this wave's share of the group's copy: wave-instructions (1 KB each) q = tl, tl + tw, ... of the block's (gfl + afl) / 256
    constexpr int n_inst = grp_fl / 256, N_MINE = n_inst / NT;
    static_assert(n_inst % NT == 0, "every wave of a group issues the same number of copies");
    auto stage = [&](int blk) {  // blk relative to b0; past the end of the sub-range: zeros
        const bool real = blk < nb;
        const float* gsrc = real ? G + (size_t)(b0 + blk) * gfl : ws_zero_block;
        const float* asrc = real ? A + (size_t)(b0 + blk) * afl : ws_zero_block;
        float* dst = my0 + (blk % NBUF) * buf_fl;
#pragma unroll
        for (int j = 0; j < N_MINE; j++) {
            const int f = (tl + j * NT) * 256;  // float offset of this instruction inside the group's block (wave-uniform)
            const float* src = f < gfl ? gsrc + f : asrc + (f - gfl);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane * 4), (__attribute__((address_space(3))) void*)(dst + f), 16, 0, 0);
        }
    };
    // LDS float index of this lane's operands inside a group's block: G row r, tile t: r * Cout + tco * 128 + 4 i + t; A row r, tile u: gfl + r * Cin + ...
    const int gofs = (8 * h) * Cout + tco * 128 + 4 * i, aofs = gfl + (8 * h) * Cin + tci * (32 * TCI) + TCI * i;

    f32x16 acc[4][TCI];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int u = 0; u < TCI; u++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][u][r] = 0.f;
    u32x4 gp[2][4][3], ap[2][TCI][3];  // [plane set][tile][hi / mid / lo]

#pragma unroll
    for (int c = 0; c < NBUF; c++) stage(c);
    // planes of block 0, not hidden (once per workgroup); counted wait: the copies of blocks 1 .. NBUF - 1 stay in flight
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((NBUF - 1) * N_MINE & 15) | (((NBUF - 1) * N_MINE >> 4) << 14));
    __builtin_amdgcn_s_barrier();
    {
        const float* src = my0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
#pragma unroll
            for (int t = 0; t < 4; t++) set_dword(gp[0][t], p, split_pair3(sg(src[gofs + (2 * p) * Cout + t]), sg(src[gofs + (2 * p + 1) * Cout + t])));
#pragma unroll
            for (int u = 0; u < TCI; u++) set_dword(ap[0][u], p, split_pair3(src[aofs + (2 * p) * Cin + u], src[aofs + (2 * p + 1) * Cin + u]));
        }
    }
    // One block: 4 x TCI tile groups of TERMS MFMAs; in the shadow of group (t, u) the planes of the NEXT block get G tile t, row pairs
    // u * UPG .. + UPG - 1, and (for t < TCI) A tile t, the same pairs.  CUR / NXT: plane sets; `src`: the next block in LDS (nullptr: none).
    // In row t of the tile grid the NEXT block's row pair t (rows 8 h + 2 t, + 1 of the lane's half) is split for all tiles: the lane's four G floats
    // (one per tile) and TCI A floats of a row come as ONE 16- / 8-byte LDS read each (4-byte reads of this pattern are 4-way bank conflicts and
    // four times as many), and are consumed within the row.
    auto block = [&](auto cur_c, const float* src) {
        constexpr int CUR = decltype(cur_c)::value, NXT = 1 - CUR;
        // the lane's four G floats and TCI A floats of a row pair, read ONE ROW PAIR AHEAD of their use (round 2 had no registers for this -- 16 more and
        // the kernel spilled; it fits since the split's first step carries its values in the unit's own registers: 232 of 256): 320 -> 308 us
        f32x4 gq[2][2]; avec aq[2][2];
        auto rd = [&](int t, int w) {
#ifdef WS_ABL_NOLDSREAD
            gq[w][0] = gq[w][1] = f32x4{1.f, 2.f, 3.f, 4.f}; aq[w][0] = aq[w][1] = avec(1.5f); return;
#endif
            gq[w][0] = *reinterpret_cast<const f32x4*>(src + gofs + (2 * t) * Cout); gq[w][1] = *reinterpret_cast<const f32x4*>(src + gofs + (2 * t + 1) * Cout);
            aq[w][0] = *reinterpret_cast<const avec*>(src + aofs + (2 * t) * Cin); aq[w][1] = *reinterpret_cast<const avec*>(src + aofs + (2 * t + 1) * Cin);
        };
        rd(0, 0);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (t < 3) rd(t + 1, (t + 1) & 1);  // the next row pair's four reads are in flight under this row's MFMAs
            BG_PIN();
            const f32x4 g0 = gq[t & 1][0], g1 = gq[t & 1][1];
            const avec a0 = aq[t & 1][0], a1 = aq[t & 1][1];
#pragma unroll
            for (int u = 0; u < TCI; u++) {
                float x[4], y[4];
                Pair3 o[4];
                // units of this group: G tiles u * UPG .. + UPG - 1, then A tile u
#pragma unroll
                for (int k = 0; k < UPG; k++) { x[k] = g0[u * UPG + k]; y[k] = g1[u * UPG + k]; }
                x[UPG] = a0[u]; y[UPG] = a1[u];
                BG_PIN();
                tile_mfmas<TERMS, UPG + 1>(acc[t][u], gp[CUR][t], ap[CUR][u], x, y, o, sgn);
#pragma unroll
                for (int k = 0; k < UPG; k++) set_dword(gp[NXT][u * UPG + k], t, o[k]);
                set_dword(ap[NXT][u], t, o[UPG]);
            }
        }
    };
    using c0 = std::integral_constant<int, 0>; using c1 = std::integral_constant<int, 1>;
    // Block b multiplies plane set b & 1 while the planes of block b + 1 are built from LDS buffer (b + 1) & 1; at the top of block b the copy of
    // block b + 2 into buffer b & 1 (whose rows were consumed during block b - 1) is started.  One barrier per block.  ONE code path holds all the
    // MFMAs (peeled tails made the register allocator pass accumulators through scratch): a sub-range is an even number of blocks, and the last
    // block of a sub-range builds planes nobody uses from whatever the other buffer holds.
    // top of block c: its successor's rows (copy started NBUF - 1 blocks ago) must have landed -- a COUNTED wait, the younger copies stay in flight
    // (raw s_barrier: __syncthreads() would drain them) -- then everybody is done with buffer c % NBUF and the copy of block c + NBUF goes there
    auto top = [&](int c) {
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((NBUF - 2) * N_MINE & 15) | (((NBUF - 2) * N_MINE >> 4) << 14));
#ifndef WS_ABL_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
#ifndef BG_PROBE_NO_STAGE  // tools/archive/wgrad_split_parts_probe.py: the loop without its copies (never defined in the product build)
        stage(c + NBUF);
#endif
    };
    int nbmax = 0;
    for (int c = 0; c < ks; c++) {
        const long wc = (long)slice * ks + c;
        const int len = 2 * (int)(NB2 * (wc + 1) / W) - 2 * (int)(NB2 * wc / W);
        nbmax = len > nbmax ? len : nbmax;
    }
    for (int b = 0; b < nbmax; b += 2) {
        top(b);
        block(c0{}, my0 + ((b + 1) % NBUF) * buf_fl);
        top(b + 1);
        block(c1{}, my0 + ((b + 2) % NBUF) * buf_fl);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the copies past the end (zeros) must not land in the reduction buffer
    __syncthreads();
    // In-workgroup reduction and store: as bg_wgrad.hip's wgrad_tile (C layout of a 32 x 32 tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h;
    // tile (t, u) holds output rows co = 4 row + t and columns ci = TCI col + u)
    float (*red)[64 * 16 * 2 * TCI] = reinterpret_cast<float (*)[64 * 16 * 2 * TCI]>(lds);
    float* pt = P + (size_t)slice * Cout * Cin + (size_t)(tco * 128) * Cin + tci * (32 * TCI) + TCI * i;
    const int ipw = 32 / ks;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        if (half) __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                avec v;
#pragma unroll
                for (int u = 0; u < TCI; u++) v[u] = acc[2 * half + tt][u][r];
                *reinterpret_cast<avec*>(&red[wave][((tt * 16 + r) * 64 + lane) * TCI]) = v;
            }
        __syncthreads();
        for (int k = 0; k < ipw; k++) {
            const int it = ksub * ipw + k, tt = it >> 4, r = it & 15, idx = ((tt * 16 + r) * 64 + lane) * TCI;
            // (sub-range c of this slice accumulated the negated sums where slice * ks + c is odd: the sign goes back here)
            avec v = *reinterpret_cast<const avec*>(&red[tl][idx]);
            if (((long)slice * ks) & 1) v = -v;
            for (int c = 1; c < ks; c++) {
                const avec w = *reinterpret_cast<const avec*>(&red[tl + tw * c][idx]);
                if (((long)slice * ks + c) & 1) v -= w; else v += w;
            }
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h, t = 2 * half + tt;
            *reinterpret_cast<avec*>(pt + (size_t)(4 * row + t) * Cin) = v;
        }
    }
}

template <int TERMS>
__global__ __launch_bounds__(256, 1) void mlp_wgrad_group_split_kernel(WgradGroup grp) {
    __shared__ __attribute__((aligned(16))) float lds[WS_LDS_FLOATS];
    const int b = blockIdx.x;
    int k = 0;
#pragma unroll
    for (int j = 1; j < WG_MAX_PROBLEMS; j++)
        if (j < grp.np && b >= grp.p[j].wg_begin) k = j;
    const WgradProblem& pr = grp.p[k];
    const int groups = pr.ntiles / pr.tw, local = b - pr.wg_begin, tile0 = (local % groups) * pr.tw, slice = local / groups;
    // the hidden-layer shapes of the two networks (utils/model.py:9-26); others are refused on the host and take the fp32 launch
    if (pr.Cout == 256 && pr.Cin == 256) wgrad_split_tile<256, 256, TERMS>(lds, pr.M, pr.G, pr.A, pr.P, pr.ntile_ci, tile0, pr.tw, slice, pr.slices);
    else if (pr.Cout == 128 && pr.Cin == 256) wgrad_split_tile<128, 256, TERMS>(lds, pr.M, pr.G, pr.A, pr.P, pr.ntile_ci, tile0, pr.tw, slice, pr.slices);
    else if (pr.Cout == 128 && pr.Cin == 128) wgrad_split_tile<128, 128, TERMS>(lds, pr.M, pr.G, pr.A, pr.P, pr.ntile_ci, tile0, pr.tw, slice, pr.slices);
    else wgrad_split_tile<256, 64, TERMS>(lds, pr.M, pr.G, pr.A, pr.P, pr.ntile_ci, tile0, pr.tw, slice, pr.slices);
}

static int split_launch(const bg_wgrad_problem* problems, int32_t count, int32_t terms, void* stream, bool finish) {
    if (terms != 9 && terms != 6) return bg_set_error(-4, "bg_mlp_weight_grad_group_split: terms must be 9 or 6");
    WgradGroup grp;
    int wg = 0, fin = 0;
    const int rc = bg_wgrad_group_fill(problems, count, grp, wg, fin, "bg_mlp_weight_grad_group_split");
    if (rc) return rc;
    for (int k = 0; k < count; k++) {
        const WgradProblem& p = grp.p[k];
        // the tw waves of a sub-range stage ALL columns of the layer: the workgroup's tiles must be the layer's tiles
        if (p.tw != p.ntiles) return bg_set_error(-4, "bg_mlp_weight_grad_group_split: tiles_per_workgroup must equal the layer's tile count (1, 2 or 4)");
        if (p.M % (2 * WS_BLOCK) != 0) return bg_set_error(-4, "bg_mlp_weight_grad_group_split: M must be a multiple of 32");
        if ((long)p.slices * (4 / p.tw) > p.M / (2 * WS_BLOCK)) return bg_set_error(-4, "bg_mlp_weight_grad_group_split: too many slices for M");
        if ((4 / p.tw) * WS_BLOCK * (p.Cout + p.Cin) * 2 > WS_LDS_FLOATS) return bg_set_error(-4, "bg_mlp_weight_grad_group_split: layer too wide for the LDS staging");
        const bool known = (p.Cout == 256 && (p.Cin == 256 || p.Cin == 64)) || (p.Cout == 128 && (p.Cin == 256 || p.Cin == 128));
        if (!known) return bg_set_error(-4, "bg_mlp_weight_grad_group_split: unsupported shape (256 x 256, 128 x 256, 128 x 128, 256 x 64)");
    }
    hipStream_t st = (hipStream_t)stream;
    if (terms == 9) hipLaunchKernelGGL(mlp_wgrad_group_split_kernel<9>, dim3(wg), dim3(256), 0, st, grp);
    else hipLaunchKernelGGL(mlp_wgrad_group_split_kernel<6>, dim3(wg), dim3(256), 0, st, grp);
    if (finish && bg_wgrad_group_finish_launch(grp, fin, st)) return bg_set_error(-2, "bg_mlp_weight_grad_group_split: launch failed");
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bg_mlp_weight_grad_group_split(const bg_wgrad_problem* problems, int32_t count, int32_t terms, void* stream) {
    return split_launch(problems, count, terms, stream, true);
}
// ... without its finish: the partial tiles stay in `scratch` for bg_update_tail (as bg_mlp_weight_grad_group_partial; same scratch layout)
extern "C" int bg_mlp_weight_grad_group_split_partial(const bg_wgrad_problem* problems, int32_t count, int32_t terms, void* stream) {
    return split_launch(problems, count, terms, stream, false);
}
