// Cycles per v_mfma_f32_32x32x2_f32 (64 cycles in the pipe) with NF VALU fillers pinned behind every MFMA, one wave per SIMD, FOUR accumulators used in turn.
// KIND 0: plain VGPR fp32 ops (fma / add chain independent of the MFMAs); 1: the chain kernel's epilogue element (accumulator read, add, mul, v_exp, cmp,
// add, select, written back to the accumulator file) repeated NF times on registers the MFMAs do not touch; 2: v_exp_f32 only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND, int NF>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters, float a, float b) {
    f32x16 acc[4], other[2];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = threadIdx.x * 1e-3f + t;
    for (int t = 0; t < 2; t++) for (int r = 0; r < 16; r++) other[t][r] = threadIdx.x * 1e-2f - r;
    float f0 = threadIdx.x * 0.5f, f1 = threadIdx.x * 0.25f + 1.f;
    const long long c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; f++) {
                if (KIND == 0) { if (f & 1) f0 = fmaf(f0, 0.999f, f1); else f1 = f1 * 1.0001f + 0.5f; }
                else if (KIND == 1) { float x = other[(u >> 3) & 1][(u * NF + f) & 15] + f0; x = x > 0.f ? x : __expf(x) - 1.0f; other[(u >> 3) & 1][(u * NF + f) & 15] = x; }
                else { f0 = __expf(f0) ; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long c1 = clock64();
    float s = f0 + f1;
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    for (int t = 0; t < 2; t++) for (int r = 0; r < 16; r++) s += other[t][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}
template <int KIND, int NF> void run(float* d, long long* dc) {
    static long long h[256];
    const int it = 1000;
    hipLaunchKernelGGL((k<KIND, NF>), dim3(256), dim3(256), 0, 0, d, dc, it, 1.0f, 0.001f);
    hipLaunchKernelGGL((k<KIND, NF>), dim3(256), dim3(256), 0, 0, d, dc, it, 1.0f, 0.001f);
    hipDeviceSynchronize();
    hipMemcpy(h, dc, sizeof(long long) * 256, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < 256; b++) s += (double)h[b];
    printf("{\"kind\": %d, \"fillers_per_mfma\": %d, \"shader_cycles_per_mfma\": %.1f}\n", KIND, NF, s / 256 / ((double)it * 16));
    fflush(stdout);
}
int main() {
    float* d; hipMalloc(&d, (size_t)256 * 256 * 4); long long* dc; hipMalloc(&dc, sizeof(long long) * 256);
    run<0, 0>(d, dc); run<0, 4>(d, dc); run<0, 8>(d, dc); run<0, 12>(d, dc); run<0, 16>(d, dc); run<0, 24>(d, dc);
    run<1, 1>(d, dc); run<1, 2>(d, dc); run<1, 3>(d, dc);
    run<2, 1>(d, dc); run<2, 2>(d, dc); run<2, 4>(d, dc);
    return 0;
}
