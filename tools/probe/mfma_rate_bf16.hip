// Sustained bf16 MFMA rate (v_mfma_f32_32x32x16_bf16, operands in registers, 4 independent accumulators or ONE dependent chain), 1-3 waves per SIMD,
// short and long launches, with the shader clock each launch ran at (clock64 / wall_clock64).  Data sheet: 2.5 PFLOP/s dense at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_loop(float* out, long long* clk, int iters, unsigned a, unsigned b) {
    f32x16 acc[CHAINS];
    for (int t = 0; t < CHAINS; t++) for (int r = 0; r < 16; r++) acc[t][r] = threadIdx.x * 1e-3f + t;
    const u32x4 xa = {a, a + threadIdx.x, a, a}, xb = {b, b, b + threadIdx.x, b};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16 / CHAINS; u++)
#pragma unroll
            for (int t = 0; t < CHAINS; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xa), __builtin_bit_cast(bf16x8, xb), acc[t], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int t = 0; t < CHAINS; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int CHAINS> void run(float* d, long long* dc, int w, int it) {
    static long long h[2 * 1024];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(256 * w), dim3(256), 0, 0, d, dc, 10, 0x3f803f80u, 0x3c003c00u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(256 * w), dim3(256), 0, 0, d, dc, it, 0x3f803f80u, 0x3c003c00u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, dc, sizeof(long long) * 2 * 256 * w, hipMemcpyDeviceToHost);
    double g = 0; for (int b = 0; b < 256 * w; b++) g += (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;
    const double flop = (double)256 * w * 4 * (double)it * 16 * 32768.0;
    printf("{\"accumulators\": %d, \"waves_per_simd\": %d, \"kernel_ms\": %.3f, \"pflops\": %.3f, \"frac_of_2.5\": %.3f, \"shader_GHz\": %.3f, \"cycles_per_mfma_per_simd\": %.1f}\n",
           CHAINS, w, ms, flop / ms / 1e12, flop / ms / 1e12 / 2.5, g / (256 * w), ms * 1e-3 * (g / (256 * w)) * 1e9 / ((double)it * 16 * w));
    fflush(stdout);
}
int main() {
    float* d; hipMalloc(&d, (size_t)256 * 8 * 256 * 4);
    long long* dc; hipMalloc(&dc, sizeof(long long) * 2 * 1024);
    const int its[3] = {200, 4000, 200000};
    for (int it : its) for (int w = 1; w <= 3; w++) { run<4>(d, dc, w, it); run<1>(d, dc, w, it); }
    return 0;
}
