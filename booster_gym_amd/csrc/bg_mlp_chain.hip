// The forward chain of one network of the PPO update (reference utils/model.py:9-26: three Linear + ELU layers in front of the output layer; called
// from utils/runner.py:132,147) as ONE kernel per 128-row slab, gfx950 only.  Activations go from layer to layer in REGISTERS.
//   * The products are computed transposed (D = W X^T: the weights are the A operand, read from LDS; the activations are the B operand): the
//     accumulator layout of v_mfma_f32_32x32x2_f32 then gives a lane ONE sample and 16 features per tile, feature 32 t + (r & 3) + 8 (r >> 2) + 4 h
//     in register r of tile t (h = lane >> 5) -- which is exactly the B operand of k-step 16 t + r of the next layer when that layer walks its k
//     in the same permuted order (the weights are staged in LDS, so their order is free).  No transposition, no LDS round trip, no re-read of the
//     activations; the HBM store of every layer's activations (the backward pass needs them) is fire-and-forget.
//   * One wave per SIMD (inputs K/2 registers + accumulators N/2), so nothing but the wave's own instruction stream hides latency:
//     - the weights of all three layers are ONE stream of 32-wide k-chunks through four LDS buffers, copied by global_load_lds_dwordx4 three chunks
//       ahead of their use; one barrier per chunk; the copies and the activation stores are counted by hand (behind()), so a wait never asks for
//       more than the chunk it needs;
//     - consecutive MFMAs go to different accumulators; the last chunk of a layer runs two tiles at a time, and bias + ELU + store of a finished
//       pair are issued under the MFMAs of the next pair;
//   * Up to four networks in ONE launch (bg_mlp_chain_forward_group): workgroups are dispatched in block order, so the slabs of the first network
//     (the critic, whose values the GAE and with it the actor's loss wait for) go first and the slabs of the next fill the machine as they retire --
//     the critic's 800 slabs are 3.125 rounds of the 256 CUs, and as two launches on two streams the partial round stayed partly empty.
//   Same sums in the same order as bg_mlp.hip's per-layer kernels: the results are bit-identical to three bg_mlp_layer_forward launches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/booster_gym_amd.h"

extern int bg_set_error(int code, const char* msg);

#ifdef BG_CHAIN_PROBE_STAMPS  // tools/mlp_chain_stamps.py: shader-clock stamp of every wave behind every chunk barrier (never defined in the product build)
__device__ long long bg_chain_stamp_buf[2048 * 4 * 24];
extern "C" int bg_probe_read_chain_stamps(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(bg_chain_stamp_buf), bytes); }
#define BG_CHAIN_STAMP(K) stamps[K] = clock64()
#else
#define BG_CHAIN_STAMP(K) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;    // k-chunk staged in LDS (one row of a chunk = 128 bytes = 8 units of 16 bytes)
constexpr int NMAX = 256;
constexpr int BUF = NMAX * KC;  // floats per LDS buffer
constexpr int NBUF = 4, AHEAD = 3;
constexpr int CHAIN_MAX = 4;
struct ChainGroup { int n; int begin[CHAIN_MAX + 1]; bg_mlp_chain net[CHAIN_MAX]; };

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
__device__ __forceinline__ void dma_rows(const float* __restrict__ W, int kc, float* sWbuf, int wave, unsigned lane_ofs) {
#pragma unroll
    for (int u = 0; u < N / 32; u++) {
        const int q = u * 4 + wave;  // wave-uniform: rows 8 q .. 8 q + 7
        // inline asm, not __builtin_amdgcn_global_load_lds: the compiler orders LDS reads behind a DMA it knows about with vmcnt(0) (it cannot tell
        // the buffers apart) and drains vmcnt at every workgroup fence; the bookkeeping of these copies is explicit here (wait_vm / behind()).
        // Scalar base + 32-bit lane offset: no 64-bit address registers per copy.
        const float* base = W + (size_t)q * 8 * K + kc * KC;
        const unsigned lds = (unsigned)(uintptr_t)(sWbuf + q * 256);
        // M0 is written here.  It cannot be named as a clobber (the backend reserves M0: "inline asm clobber list contains reserved registers" and
        // the entry is ignored); the backend never keeps a value of its own live in M0 across an inline asm -- it writes M0 immediately in front of each
        // instruction of its own that reads it -- and nothing else in this kernel uses M0.
        // Invariants the vmcnt bookkeeping (S::behind) rests on: every wave issues exactly N / 32 of these per chunk and every activation store
        // of chain_slab unconditionally -- no copy and no store may sit under a lane- or wave-dependent branch.
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_ofs), "s"(base), "s"(lds) : "memory");
    }
}
// byte offset of this lane's 16 bytes inside an 8-row group of a chunk of a [.][K] matrix: row (lane >> 3), unit (lane & 7) ^ ((n >> 1) & 7) with
// n = 8 q + (lane >> 3): (n >> 1) & 7 = (4 (q & 1) + (lane >> 4)) & 7, and the parity of q = 4 u + wave is the wave's
template <int K>
__device__ __forceinline__ unsigned dma_lane_offset(int lane, int wave) {
    return (unsigned)(((lane >> 3) * K + 4 * ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7))) * 4);
}

// One 128-row slab of one network.  Y1 / Y2 / Y3 hold whole slabs (the stores are unconditional: their number is part of the vmcnt bookkeeping).
template <int K0, int N1, int N2, int N3>
__device__ __forceinline__ void chain_slab(const bg_mlp_chain& a, int slab, float* sW, float* sB) {
    constexpr int C0 = K0 / KC, C1 = N1 / KC, C2 = N2 / KC, C = C0 + C1 + C2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const unsigned lo1 = dma_lane_offset<K0>(lane, wave), lo2 = dma_lane_offset<N1>(lane, wave), lo3 = dma_lane_offset<N2>(lane, wave);
    const float* __restrict__ W1 = a.W1; const float* __restrict__ W2 = a.W2; const float* __restrict__ W3 = a.W3;
    // chunk cc of the stream: DMA instructions per wave (0 past the end); activation stores per lane issued in iteration cc (a layer's last chunk)
    struct S {
        static constexpr int ndma(int cc) { return (cc < C0 ? N1 : cc < C0 + C1 ? N2 : cc < C ? N3 : 0) / 32; }
        static constexpr int stores_in(int cc) { return cc == C0 - 1 ? N1 / 8 : cc == C0 + C1 - 1 ? N2 / 8 : cc == C - 1 ? N3 / 8 : 0; }
        // vector-memory operations issued behind the DMA of chunk cc when iteration cc begins: the DMAs of chunks cc + 1 .. cc + AHEAD - 1 and the
        // stores of the iterations since its issue (it went out at the top of iteration cc - AHEAD, or in the prologue)
        static constexpr int behind(int cc) {
            int n = 0;
            for (int k = cc + 1; k < cc + AHEAD; k++) n += ndma(k);
            for (int it = (cc - AHEAD > 0 ? cc - AHEAD : 0); it < cc; it++) n += stores_in(it);
            return n;
        }
    };
    auto dma = [&](int cc) {
        float* dst = sW + (cc % NBUF) * BUF;
        if (cc < C0) dma_rows<K0, N1>(W1, cc, dst, wave, lo1);
        else if (cc < C0 + C1) dma_rows<N1, N2>(W2, cc - C0, dst, wave, lo2);
        else if (cc < C) dma_rows<N2, N3>(W3, cc - C0 - C1, dst, wave, lo3);
    };
#ifdef BG_CHAIN_PROBE_STAMPS
    long long stamps[24];
#endif
    BG_CHAIN_STAMP(0);
    dma(0);
    dma(1);
    dma(2);
    __builtin_amdgcn_sched_barrier(0);  // (the input rows below are loaded behind the copies: behind(0) then asks for a little more than necessary)
    for (int j = threadIdx.x; j < N1 + N2 + N3; j += 256) sB[j] = j < N1 ? a.b1[j] : j < N1 + N2 ? a.b2[j - N1] : a.b3[j - N1 - N2];
    const int row = slab * 128 + wave * 32 + i;
    const float* xrow = a.X + (size_t)(row < a.M ? row : a.M - 1) * K0 + 4 * h;
    float x0[K0 / 2];
#pragma unroll
    for (int j = 0; j < K0 / 8; j++) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xrow + 8 * j);
        x0[4 * j + 0] = v.x; x0[4 * j + 1] = v.y; x0[4 * j + 2] = v.z; x0[4 * j + 3] = v.w;
    }
    const int sx = (i >> 1) & 7;  // this lane's unit swizzle (rows 32 t + i: the tile offset does not change it)
    const int lofs = i * KC;
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
            const float* sw = sW + (c % NBUF) * BUF + lofs;
            // chunk c complete in LDS (this wave's part), then published by the barrier; everything issued behind its DMA may stay in flight
            __builtin_amdgcn_sched_barrier(0);
            wait_vm<S::behind(c)>();
            if (c == 0) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the bias values written to sB above
            asm volatile("s_barrier" ::: "memory");  // no fence: a workgroup fence would drain vmcnt (stores and younger copies included)
            BG_CHAIN_STAMP(c + 1);
            __builtin_amdgcn_sched_barrier(0);
            dma(c + AHEAD);
            __builtin_amdgcn_sched_barrier(0);
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
    layer(a1, [&](int s) { return x0[s]; }, std::integral_constant<int, K0>{}, std::integral_constant<int, N1>{}, std::integral_constant<int, 0>{}, 0, a.Y1);
    layer(a2, [&](int s) { return a1[s >> 4][s & 15]; }, std::integral_constant<int, N1>{}, std::integral_constant<int, N2>{}, std::integral_constant<int, C0>{}, N1, a.Y2);
    layer(a3, [&](int s) { return a2[s >> 4][s & 15]; }, std::integral_constant<int, N2>{}, std::integral_constant<int, N3>{}, std::integral_constant<int, C0 + C1>{}, N1 + N2, a.Y3);
    if (a.v_out) {
        // scalar output layer on the last activations, straight from the registers that hold them: the lane has 64 of its sample's 128 features
        // (behind the last wait of the chunk stream: the store below is nobody's business)
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
    stamps[C + 1] = clock64();
    if (lane == 0 && blockIdx.x < 2048)
        for (int k = 0; k < 24; k++) bg_chain_stamp_buf[((size_t)blockIdx.x * 4 + wave) * 24 + k] = k <= C + 1 ? stamps[k] : 0;
#endif
}

// TAG: 1 / 2 = one network with N2 = 128 / 256 (the layer shape gets its own kernel symbol: a profiler's per-kernel average is then the average of
// ONE shape, and the launch carries only that shape's code), 0 = a group, shapes looked up per workgroup.
template <int TAG>
__global__ __launch_bounds__(256) void mlp_chain_fwd_kernel(ChainGroup grp) {
    __shared__ __attribute__((aligned(16))) float sW[NBUF * BUF];
    __shared__ __attribute__((aligned(16))) float sB[3 * NMAX];
    int k = 0;
    if constexpr (TAG == 0) {
#pragma unroll
        for (int j = 1; j < CHAIN_MAX; j++)
            if (j < grp.n && (int)blockIdx.x >= grp.begin[j]) k = j;
    }
    const bg_mlp_chain& a = grp.net[k];
    // One slab per workgroup, or (bg_mlp_chain::workgroups > 0) that many workgroups walking the network's slabs: the update's two networks, launched side
    // by side on two streams, then SHARE the chip by CUs instead of by slabs (a workgroup fills a CU: 128 KB of LDS).  Equal-sized slabs of unequal cost
    // (critic 65 us, actor 41 us) dispatched as CUs fall free run in lockstep rounds and leave the last 32 slabs to run alone: 370 us for the pair against
    // 325 us of work per CU; 160 workgroups x 5 critic slabs beside 96 x 8 actor slabs is 323 / 328 us (utils/runner.py plans the split).
    const int nslabs = (a.M + 127) / 128, stride = grp.begin[k + 1] - grp.begin[k];
    for (int slab = blockIdx.x - grp.begin[k]; slab < nslabs; slab += stride) {
        if (TAG == 2 || (TAG == 0 && a.N2 == 256)) chain_slab<64, 256, 256, 128>(a, slab, sW, sB);
        else chain_slab<64, 256, 128, 128>(a, slab, sW, sB);
        // every wave has read the last chunk (and the biases) before anybody's copies of the next slab land in the buffers
        asm volatile("s_barrier" ::: "memory");
    }
}

static int chain_check(const bg_mlp_chain& q) {
    if (q.M <= 0 || !q.X || !q.W1 || !q.b1 || !q.W2 || !q.b2 || !q.W3 || !q.b3 || !q.Y1 || !q.Y2 || !q.Y3) return bg_set_error(-1, "bg_mlp_chain_forward: bad argument");
    if ((((uintptr_t)q.X | (uintptr_t)q.W1 | (uintptr_t)q.W2 | (uintptr_t)q.W3 | (uintptr_t)q.Y1 | (uintptr_t)q.Y2 | (uintptr_t)q.Y3 | (uintptr_t)q.b1 |
          (uintptr_t)q.b2 | (uintptr_t)q.b3) & 15) != 0)
        return bg_set_error(-1, "bg_mlp_chain_forward: pointers must be 16-byte aligned");
    if ((q.v_w || q.v_b || q.v_out) && (!q.v_w || !q.v_b || !q.v_out || ((uintptr_t)q.v_w & 15) != 0))
        return bg_set_error(-1, "bg_mlp_chain_forward: value head needs v_w (16-byte aligned), v_b and v_out");
    if (!(q.K0 == 64 && q.N1 == 256 && (q.N2 == 128 || q.N2 == 256) && q.N3 == 128))
        return bg_set_error(-4, "bg_mlp_chain_forward: unsupported widths (64-256-128-128 and 64-256-256-128)");
    return 0;
}
extern "C" int bg_mlp_chain_forward_group(const bg_mlp_chain* nets, int32_t count, void* stream) {
    if (!nets || count <= 0 || count > CHAIN_MAX) return bg_set_error(-1, "bg_mlp_chain_forward_group: 1 to 4 networks");
    ChainGroup grp;
    grp.n = count;
    int blocks = 0;
    for (int k = 0; k < count; k++) {
        const int rc = chain_check(nets[k]);
        if (rc) return rc;
        grp.begin[k] = blocks;
        grp.net[k] = nets[k];
        const int slabs = (nets[k].M + 127) / 128;
        if (nets[k].workgroups < 0) return bg_set_error(-1, "bg_mlp_chain_forward_group: workgroups < 0");
        blocks += nets[k].workgroups > 0 && nets[k].workgroups < slabs ? nets[k].workgroups : slabs;
    }
    grp.begin[count] = blocks;
    if (count > 1) hipLaunchKernelGGL(mlp_chain_fwd_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    else if (nets[0].N2 == 256) hipLaunchKernelGGL(mlp_chain_fwd_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    else hipLaunchKernelGGL(mlp_chain_fwd_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grp);
    if (hipGetLastError() != hipSuccess) return bg_set_error(-2, "bg_mlp_chain_forward_group: launch failed");
    return 0;
}
extern "C" int bg_mlp_chain_forward(int32_t M, int32_t K0, int32_t N1, int32_t N2, int32_t N3, const float* X, const float* W1, const float* b1,
                                    const float* W2, const float* b2, const float* W3, const float* b3, float* Y1, float* Y2, float* Y3, void* stream) {
    bg_mlp_chain q;
    q.M = M; q.K0 = K0; q.N1 = N1; q.N2 = N2; q.N3 = N3; q.workgroups = 0;
    q.X = X; q.W1 = W1; q.b1 = b1; q.W2 = W2; q.b2 = b2; q.W3 = W3; q.b3 = b3; q.Y1 = Y1; q.Y2 = Y2; q.Y3 = Y3;
    q.v_w = nullptr; q.v_b = nullptr; q.v_out = nullptr;
    return bg_mlp_chain_forward_group(&q, 1, stream);
}
