// Sustained fp32 MFMA rate on this part: back-to-back v_mfma_f32_32x32x2_f32 on 4 independent accumulators per wave, operands in registers, no
// memory traffic; 1, 2, 3 waves per SIMD; short (0.1 ms) and long (20 ms) launches.  Peak by the data sheet: 157.3 TFLOP/s at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) acc[t][r] = threadIdx.x * 1e-3f + t;
    float x = a + threadIdx.x * 1e-6f, y = b;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, (size_t)256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int wps[3] = {1, 2, 3};
    const int its[2] = {2000, 400000};
    for (int it : its)
        for (int w : wps) {
            hipLaunchKernelGGL(mfma_loop, dim3(256 * w), dim3(256), 0, 0, d, 10, 1.0f, 1e-6f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop, dim3(256 * w), dim3(256), 0, 0, d, it, 1.0f, 1e-6f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)256 * w * 4 * (double)it * 16 * 4096.0;  // workgroups x waves x iterations x 16 MFMAs x 4096 flop
            const double per_simd_cycles = (double)it * 16 * 64 * w;             // MFMA pipe cycles each SIMD must spend
            printf("{\"waves_per_simd\": %d, \"iterations\": %d, \"kernel_ms\": %.3f, \"tflops\": %.1f, \"frac_of_157.3\": %.3f, \"implied_clock_GHz_if_pipe_saturated\": %.3f}\n",
                   w, it, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, per_simd_cycles / (ms * 1e-3) / 1e9);
            fflush(stdout);
        }
    return 0;
}
