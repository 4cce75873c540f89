// Cycles per v_mfma_f32_32x32x16_bf16 with NF single-issue VALU fillers (v_perm_b32 / v_and / v_sub mix, independent of the MFMAs) pinned behind every
// MFMA, one wave per SIMD: ONE accumulator (each MFMA depends on the one before, as in a k-loop) against FOUR accumulators used in turn.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS, int NF>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters, unsigned a, unsigned b) {
    f32x16 acc[CHAINS];
    for (int t = 0; t < CHAINS; t++) for (int r = 0; r < 16; r++) acc[t][r] = threadIdx.x * 1e-3f + t;
    const u32x4 xa = {a, a + threadIdx.x, a, a}, xb = {b, b, b + threadIdx.x, b};
    float f0 = threadIdx.x * 0.5f, f1 = threadIdx.x * 0.25f + 1.f;
    unsigned p = threadIdx.x;
    const long long c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xa), __builtin_bit_cast(bf16x8, xb), acc[u % CHAINS], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; f++) {
                if (f % 3 == 0) p = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0) + p, 0x07060302u);
                else if (f % 3 == 1) f0 = f0 - __uint_as_float(__float_as_uint(f0) & 0xffff0000u);
                else f1 = f1 - __uint_as_float(p & 0xffff0000u);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long c1 = clock64();
    float s = f0 + f1 + p;
    for (int t = 0; t < CHAINS; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}
template <int CHAINS, int NF> void run(float* d, long long* dc) {
    static long long h[256];
    const int it = 2000;
    hipLaunchKernelGGL((k<CHAINS, NF>), dim3(256), dim3(256), 0, 0, d, dc, it, 0x3f803f80u, 0x3c003c00u);
    hipLaunchKernelGGL((k<CHAINS, NF>), dim3(256), dim3(256), 0, 0, d, dc, it, 0x3f803f80u, 0x3c003c00u);
    hipDeviceSynchronize();
    hipMemcpy(h, dc, sizeof(long long) * 256, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < 256; b++) s += (double)h[b];
    printf("{\"accumulators\": %d, \"valu_fillers_per_mfma\": %d, \"shader_cycles_per_mfma\": %.1f}\n", CHAINS, NF, s / 256 / ((double)it * 16));
    fflush(stdout);
}
int main() {
    float* d; hipMalloc(&d, (size_t)256 * 256 * 4); long long* dc; hipMalloc(&dc, sizeof(long long) * 256);
    run<1, 0>(d, dc); run<1, 1>(d, dc); run<1, 2>(d, dc); run<1, 3>(d, dc); run<1, 4>(d, dc); run<1, 6>(d, dc);
    run<4, 0>(d, dc); run<4, 1>(d, dc); run<4, 2>(d, dc); run<4, 3>(d, dc); run<4, 4>(d, dc); run<4, 6>(d, dc);
    return 0;
}
