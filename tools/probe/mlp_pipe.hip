// PROBE, not part of libbooster_gym_amd.so.  Round-3 experiments on a PERSISTENT form of mlp_fwd_kernel (tools/mlp_pipe_probe.py; all variants
// bit-identical to mlp_fwd_kernel; times alone on the GPU at M = 98,304, current kernel 118.7 us at 256 x 256, 55.2 us at 256 -> 128):
//   1. two accumulator sets, the previous item's epilogue issued under the next item's MFMAs, 64 x 128 items (two tiles per wave): 144.5 / 74.4 us.
//      The compiler emits a chunk's MFMAs first and the epilogue as one clump behind them, and -- the real flaw -- stores and loads share vmcnt on
//      this part, so the vmcnt(0) in front of every chunk barrier drained the previous item's stores right there;
//   2. the same with the epilogue's VALUES computed under the MFMAs and its STORES issued behind the chunk's wait + barrier (a whole chunk to
//      drain): 130.9 / 69.4 us; 128 x 128 items (four tiles, 2 x 64 accumulators) spill 270 B per lane at two waves per SIMD;
//   3. (this file, "design Q") ONE accumulator set, three waves per SIMD like mlp_fwd_kernel, weights staged by global_load_lds, an item's
//      epilogue right behind its loop with the stores NOT waited for (the next item's first operands were fetched under the last chunk, its
//      chunk 0 starts at once and the stores drain under it): 114.5 us at 256 x 256 with 768 workgroups x 2 items (-3.6 %), but 59.9 / 34.9 /
//      65.1 us against 55.2 / 31.7 / 59.9 at 256 -> 128, 128 -> 128, 128 -> 256 (profiles/r03_mlp_pipe_probe_designQ.log).
// Conclusion: hiding the store tail buys a few per cent on the one shape with two items per workgroup and loses elsewhere; the lockstep of
// the co-resident workgroups is untouched by persistence.  Not adopted.
//
// Persistent, software-pipelined form of the fused fp32-MFMA layer kernels of bg_mlp.hip (reference utils/model.py:9-26 Linear + ELU stacks under
// utils/runner.py:132,147,163), gfx950 only.
//
// Why.  A wave of mlp_fwd_kernel spends 30 % of its life in the epilogue (bias + ELU, the transposition through LDS, the stores: VALU-issue
// bound, no MFMA in flight), and the co-resident workgroups of a CU start together and reach their epilogues together, so the matrix pipe idles
// for them (90 % busy while workgroups are resident, 72-75 % over the launch).  Here a workgroup stays resident and walks a
// list of (128-row slab, 128-column block) items with TWO accumulator sets: while the MFMAs of item n + 1 run, the epilogue of item n is issued
// between them, one output tile per group of k-chunks, so the matrix pipe of a SIMD always has MFMAs of its own wave to run.  The first loads of
// the next item are issued under the last k-chunk of the current one.  Same arithmetic and the same accumulation order as mlp_fwd_kernel: the
// results are bit-identical to it.
//
// Resources: 2 x 32 accumulator registers + the operand sets -> three workgroups per CU; LDS 32 KB weight staging (double buffer, filled by
// global_load_lds) + 18 KB wave-private transposition blocks (they can no longer alias the staging buffer: the next item is already using it).
#include <hip/hip_runtime.h>

#include "../../include/booster_gym_amd.h"  // (built by tools/mlp_pipe_probe.py from tools/probe/)

static int bg_set_error(int code, const char*) { return code; }
#define HIP_OK(expr)                                                                        \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return bg_set_error(-2, hipGetErrorString(_e));               \
    } while (0)
#include "../../booster_gym_amd/csrc/bg_mlp_tile.h"

namespace {

// Workgroup item: 64 rows x 128 columns.  The 4 waves form a 2 x 2 grid: wave (wr, wc) owns rows 32 wr .. + 31 and columns 64 wc .. + 63, i.e. TWO
// 32 x 32 tiles.  With two accumulator sets that is 64 accumulator registers per wave (128-row items with four tiles per wave needed 128 and
// spilled at two waves per SIMD), three workgroups fit a CU, and a layer has twice as many items (1536 / 3072 at M = 98,304), which divide evenly
// over the resident workgroups.
constexpr int PN = 128, PBM = 128, PNT = 4, PLS = 36;

struct PipeItem {           // what the epilogue of an item needs after its main loop is over
    int rbase;              // first row of this wave's 32 rows
    float* Y;               // output, already offset to the item's column block
    const float* bias;      // bias of the column block
};

// One k-chunk of the weights ([128 rows][32 k] floats) global -> LDS with no register stop (global_load_lds_dwordx4: the LDS side of a
// wave-instruction is 64 consecutive 16-byte slots, the global side is per lane).  Rows are 128 bytes, unpadded; slot (n, q) holds the 16-byte
// piece p = q ^ ((n >> 1) & 7) of row n (swizzled on the SOURCE side), which spreads the 16 lanes of a read pass -- rows 128 bytes apart, the
// same piece -- over all 16 bank groups.  16 wave-instructions per chunk, 4 per wave.
constexpr int PSW = 32;  // LDS row stride of the staged weights (floats)
template <int K>
__device__ __forceinline__ void pipe_stage_w(const float* __restrict__ W, int kc, float* sWbuf, int wave, int lane) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int q = u * 4 + wave, s = q * 64 + lane, n = s >> 3, piece = (s & 7) ^ ((n >> 1) & 7);
        const float* g = W + (size_t)n * K + kc * FW_KC + 4 * piece;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(sWbuf + q * 256), 16, 0, 0);
    }
}
// sw = the staging buffer; lane (i, h) reads piece 2 s + h of row 32 t + i at its swizzled place
__device__ __forceinline__ void pipe_mfma_chunk(f32x16 (&acc)[PNT], const f32x4 (&a4)[4], const float* sw, int i, int h) {
    const int sz = (i >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int t = 0; t < PNT; t++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sw + (t * 32 + i) * PSW + 4 * ((2 * s + h) ^ sz));  // sw: this wave's 64 weight rows
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].x, b4.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].y, b4.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].z, b4.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s].w, b4.w, acc[t], 0, 0, 0);
        }
    }
}
// epilogue of ONE 32-column tile of a finished item, in two halves.  COMPUTE: C layout -> (row, 4 columns) through the wave's LDS block, + bias,
// ELU, the four 16-byte row pieces left in registers.  STORE: the global stores.  They are separate because loads and stores share vmcnt on
// this part and return out of order against each other: a store issued before the chunk's `s_waitcnt vmcnt(0)` (which the weight / A loads of
// the next chunk need) is drained right there, at full HBM write latency, in the middle of the main loop.  Issued AFTER that wait, the stores
// have the whole next chunk to drain.
template <int EPI, int T>
__device__ __forceinline__ void pipe_epilogue_compute(const f32x16 (&acc)[PNT], const f32x4 b4, float* wl, int lane, int i, int h, f32x4 (&v)[4]) {
    const int r8 = lane >> 3, c8 = (lane & 7) * 4;
#pragma unroll
    for (int r = 0; r < 16; r++) wl[((r & 3) + 8 * (r >> 2) + 4 * h) * PLS + i] = acc[T][r];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = *reinterpret_cast<const f32x4*>(&wl[(r8 + 8 * k) * PLS + c8]) + b4;
        if (EPI == 1) { v[k].x = elu_f(v[k].x); v[k].y = elu_f(v[k].y); v[k].z = elu_f(v[k].z); v[k].w = elu_f(v[k].w); }
    }
}
template <int T>
__device__ __forceinline__ void pipe_epilogue_store(const f32x4 (&v)[4], const PipeItem& it, int M, int ldy, int lane) {
    const int r8 = lane >> 3, c8 = (lane & 7) * 4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int rr = it.rbase + r8 + 8 * k;
        if (rr < M) *reinterpret_cast<f32x4*>(it.Y + (size_t)rr * ldy + T * 32 + c8) = v[k];
    }
}
template <int EPI, int T>
__device__ __forceinline__ void pipe_epilogue_tile(const f32x16 (&acc)[PNT], const PipeItem& it, int M, int ldy, float* wl, int lane, int i, int h) {
    f32x4 v[4];
    pipe_epilogue_compute<EPI, T>(acc, *reinterpret_cast<const f32x4*>(it.bias + T * 32 + (lane & 7) * 4), wl, lane, i, h, v);
    pipe_epilogue_store<T>(v, it, M, ldy, lane);
}
struct PipeOperands {  // per-item pointers of the main loop
    const float* xrow;   // this lane's row of X, offset by 4 h
    const float* W;      // weight rows of the item's column block
};

template <int K, int EPI>
struct PipeKernel {
    static constexpr int CH = K / FW_KC;
    static_assert(CH % 2 == 0, "K must be a multiple of 64");

    // Main loop of one item into `acc`, with the epilogue of the previous item (accp / prev) issued between its MFMAs when PREV.  On entry the
    // first weight chunk of this item is in sW[0] (barrier passed) and its first A chunk in aA; on exit the same holds for `next` (when it has
    // one), so the pipeline never drains between items.
    template <bool PREV>
    static __device__ __forceinline__ void item(f32x16 (&acc)[PNT], const f32x16 (&accp)[PNT], const PipeOperands& cur, const PipeOperands& next, bool has_next,
                                                const PipeItem& prev, int M, int ldy, float (&sW)[2][PN * PSW], float* wl, f32x4 (&aA)[4], int wave, int lane,
                                                int i, int h) {
        f32x4 aB[4];
#pragma unroll
        for (int t = 0; t < PNT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
        // the previous item's four tiles are finished one per CH / 4 chunks (CH == 2: two per chunk): values computed between the chunk's MFMAs
        // and its wait, stores issued behind the barrier
        f32x4 bias[PNT], v[4], v2[4];
        if constexpr (PREV) {  // all bias pieces up front: no load is issued behind a store
#pragma unroll
            for (int t = 0; t < PNT; t++) bias[t] = *reinterpret_cast<const f32x4*>(prev.bias + t * 32 + (lane & 7) * 4);
        }
#pragma unroll
        for (int kc = 0; kc < CH; kc += 2) {
            pipe_stage_w<K>(cur.W, kc + 1, sW[1], wave, lane);
            load_a_chunk(aB, cur.xrow, kc + 1);
            pipe_mfma_chunk(acc, aA, sW[0], i, h);
            if constexpr (PREV) epi_compute(accp, bias, wl, lane, i, h, v, v2, kc);
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's share of sW[1] has landed (and aB); the stores of the chunk before are long done
            __syncthreads();
            if constexpr (PREV) epi_store(v, v2, prev, M, ldy, lane, kc);
            if (kc + 2 < CH) {
                pipe_stage_w<K>(cur.W, kc + 2, sW[0], wave, lane);
                load_a_chunk(aA, cur.xrow, kc + 2);
            } else if (has_next) {  // the next item's first chunks, under this item's last MFMAs
                pipe_stage_w<K>(next.W, 0, sW[0], wave, lane);
                load_a_chunk(aA, next.xrow, 0);
            }
            pipe_mfma_chunk(acc, aB, sW[1], i, h);
            if constexpr (PREV) epi_compute(accp, bias, wl, lane, i, h, v, v2, kc + 1);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            if constexpr (PREV) epi_store(v, v2, prev, M, ldy, lane, kc + 1);
        }
    }
    // which tile(s) of the previous item belong to chunk kc (kc is a constant after unrolling: the branches fold)
    static __device__ __forceinline__ void epi_compute(const f32x16 (&accp)[PNT], const f32x4 (&bias)[PNT], float* wl, int lane, int i, int h, f32x4 (&v)[4],
                                                       f32x4 (&v2)[4], int kc) {
        if constexpr (CH == 2) {
            if (kc == 0) { pipe_epilogue_compute<EPI, 0>(accp, bias[0], wl, lane, i, h, v); pipe_epilogue_compute<EPI, 1>(accp, bias[1], wl, lane, i, h, v2); }
            else { pipe_epilogue_compute<EPI, 2>(accp, bias[2], wl, lane, i, h, v); pipe_epilogue_compute<EPI, 3>(accp, bias[3], wl, lane, i, h, v2); }
        } else {
            constexpr int STEP = CH / 4;
            if (kc == 0 * STEP) pipe_epilogue_compute<EPI, 0>(accp, bias[0], wl, lane, i, h, v);
            else if (kc == 1 * STEP) pipe_epilogue_compute<EPI, 1>(accp, bias[1], wl, lane, i, h, v);
            else if (kc == 2 * STEP) pipe_epilogue_compute<EPI, 2>(accp, bias[2], wl, lane, i, h, v);
            else if (kc == 3 * STEP) pipe_epilogue_compute<EPI, 3>(accp, bias[3], wl, lane, i, h, v);
        }
    }
    static __device__ __forceinline__ void epi_store(const f32x4 (&v)[4], const f32x4 (&v2)[4], const PipeItem& prev, int M, int ldy, int lane, int kc) {
        if constexpr (CH == 2) {
            if (kc == 0) { pipe_epilogue_store<0>(v, prev, M, ldy, lane); pipe_epilogue_store<1>(v2, prev, M, ldy, lane); }
            else { pipe_epilogue_store<2>(v, prev, M, ldy, lane); pipe_epilogue_store<3>(v2, prev, M, ldy, lane); }
        } else {
            constexpr int STEP = CH / 4;
            if (kc == 0 * STEP) pipe_epilogue_store<0>(v, prev, M, ldy, lane);
            else if (kc == 1 * STEP) pipe_epilogue_store<1>(v, prev, M, ldy, lane);
            else if (kc == 2 * STEP) pipe_epilogue_store<2>(v, prev, M, ldy, lane);
            else if (kc == 3 * STEP) pipe_epilogue_store<3>(v, prev, M, ldy, lane);
        }
    }
};

// items [blockIdx.x * per, + per) of the launch: item = slab * ncb + column block, so the column blocks of a slab follow each other in ONE
// workgroup and the second read of the slab's X rows comes out of this CU's L2
template <int K, int EPI>
__global__ __launch_bounds__(256, 3) void mlp_pipe_kernel(int M, int ldy, const float* __restrict__ X, const float* __restrict__ Wfull,
                                                          const float* __restrict__ biasfull, float* __restrict__ Yfull, int nitems, int per) {
    using PK = PipeKernel<K, EPI>;
    __shared__ __attribute__((aligned(16))) float sW[2][PN * PSW];
    __shared__ __attribute__((aligned(16))) float sT[4 * 32 * PLS];
    const int it0 = blockIdx.x * per, it1 = min(nitems, it0 + per);
    if (it0 >= it1) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int ncb = ldy / PN;
    float* wl = sT + wave * (32 * PLS);
    auto operands = [&](int it) {
        const int bx = it / ncb, by = it % ncb;
        const int row = bx * PBM + wave * 32 + i;
        PipeOperands o;
        o.xrow = X + (size_t)(row < M ? row : M - 1) * K + 4 * h;
        o.W = Wfull + (size_t)by * PN * K;
        return o;
    };
    auto epi_item = [&](int it) {
        const int bx = it / ncb, by = it % ncb;
        PipeItem e;
        e.rbase = bx * PBM + wave * 32;
        e.Y = Yfull + by * PN;
        e.bias = biasfull + by * PN;
        return e;
    };
    f32x16 acc0[PNT], accn[PNT];  // accn: never read (the PREV = false form of item() takes no previous item)
    f32x4 aA[4];
    {   // prologue: first chunks of the first item
        const PipeOperands o = operands(it0);
        pipe_stage_w<K>(o.W, 0, sW[0], wave, lane);
        load_a_chunk(aA, o.xrow, 0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
    }
    PipeItem none; none.rbase = 0; none.Y = nullptr; none.bias = nullptr;
    // Design Q: ONE accumulator set.  An item's epilogue follows its main loop at once; its 16 stores are issued and NOT waited for: the next item's
    // first operands are already there (fetched under this item's last chunk), so its chunk 0 starts immediately and the stores drain under its 64
    // MFMAs -- the first vmcnt(0) they meet is the one at the end of that chunk.
    for (int it = it0; it < it1; it++) {
        const PipeOperands cur = operands(it);
        const bool has_next = it + 1 < it1;
        const PipeOperands nxt = has_next ? operands(it + 1) : cur;
        PK::template item<false>(acc0, accn, cur, nxt, has_next, none, M, ldy, sW, wl, aA, wave, lane, i, h);
        const PipeItem e = epi_item(it);
        f32x4 bias[PNT];
#pragma unroll
        for (int t = 0; t < PNT; t++) bias[t] = *reinterpret_cast<const f32x4*>(e.bias + t * 32 + (lane & 7) * 4);
        f32x4 v[4];
        pipe_epilogue_compute<EPI, 0>(acc0, bias[0], wl, lane, i, h, v); pipe_epilogue_store<0>(v, e, M, ldy, lane);
        pipe_epilogue_compute<EPI, 1>(acc0, bias[1], wl, lane, i, h, v); pipe_epilogue_store<1>(v, e, M, ldy, lane);
        pipe_epilogue_compute<EPI, 2>(acc0, bias[2], wl, lane, i, h, v); pipe_epilogue_store<2>(v, e, M, ldy, lane);
        pipe_epilogue_compute<EPI, 3>(acc0, bias[3], wl, lane, i, h, v); pipe_epilogue_store<3>(v, e, M, ldy, lane);
    }
}

}  // namespace

// Experimental entry (not in include/booster_gym_amd.h until it wins): bg_mlp_layer_forward on the persistent pipelined kernel.
// workgroups: resident workgroups to launch (<= 0: 512 = two per CU).
extern "C" int bg_mlp_layer_forward_pipe(int32_t M, int32_t K, int32_t N, const float* X, const float* W, const float* bias, float* Y, int32_t elu,
                                         int32_t workgroups, void* stream) {
    if (M <= 0 || !X || !W || !bias || !Y) return bg_set_error(-1, "bg_mlp_layer_forward_pipe: bad argument");
    if ((((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y | (uintptr_t)bias) & 15) != 0) return bg_set_error(-1, "bg_mlp_layer_forward_pipe: pointers must be 16-byte aligned");
    if (N % 128 != 0 || N > 1024) return bg_set_error(-4, "bg_mlp_layer_forward_pipe: unsupported N (multiples of 128 up to 1024)");
    const int nitems = ((M + PBM - 1) / PBM) * (N / 128);
    int grid = workgroups > 0 ? workgroups : 768;
    if (grid > nitems) grid = nitems;
    const int per = (nitems + grid - 1) / grid;
    grid = (nitems + per - 1) / per;
    hipStream_t st = (hipStream_t)stream;
#define BG_PIPE(KK)                                                                                                                        \
    if (K == KK) {                                                                                                                         \
        if (elu) hipLaunchKernelGGL((mlp_pipe_kernel<KK, 1>), dim3(grid), dim3(256), 0, st, M, N, X, W, bias, Y, nitems, per);             \
        else hipLaunchKernelGGL((mlp_pipe_kernel<KK, 0>), dim3(grid), dim3(256), 0, st, M, N, X, W, bias, Y, nitems, per);                 \
        HIP_OK(hipGetLastError());                                                                                                         \
        return 0;                                                                                                                          \
    }
    BG_PIPE(256)
    BG_PIPE(128)
    BG_PIPE(64)
#undef BG_PIPE
    return bg_set_error(-4, "bg_mlp_layer_forward_pipe: unsupported K (64, 128, 256)");
}
