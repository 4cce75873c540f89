// Calibration of fp32 VALU issue on MI355X: cycles per v_fma_f32 (and per v_pk_fma_f32) per SIMD at 1, 2, 4, 8 resident waves per SIMD,
// for dependent chains (ILP 1) up to 8 independent chains per wave.  256-thread workgroups put one wave on each of a CU's four SIMDs, so a
// grid of 256 * k workgroups = k waves per SIMD on every CU; `blocks 128 x 64 threads` reproduces the env kernel's launch at 4096 envs
// (128 single-wave workgroups).  Output: one JSON object per line (kept under profiles/ as r02_issue_rate.jsonl).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/issue_rate.hip -o gpurun_out/issue_rate && gpurun_out/issue_rate > gpurun_out/issue_rate.jsonl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int ILP, int PK, int THREADS>
__global__ __launch_bounds__(THREADS) void fma_chain(float* out, int iters, float a, float b) {
    f32x2 x[ILP];
    for (int k = 0; k < ILP; k++) x[k] = f32x2{threadIdx.x * 1e-3f + k, threadIdx.x * 2e-3f - k};
    const f32x2 a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int k = 0; k < ILP; k++) {
                if (PK) x[k] = __builtin_elementwise_fma(x[k], a2, b2);
                else x[k].x = __builtin_fmaf(x[k].x, a, b);
            }
    }
    float s = 0;
    for (int k = 0; k < ILP; k++) s += x[k].x + x[k].y;
    out[(size_t)blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int ILP, int PK, int THREADS> void run(float* d, int blocks, const char* geom, double waves_per_simd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL((fma_chain<ILP, PK, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((fma_chain<ILP, PK, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 16 * ILP;                 // instructions per wave
    const double cyc_wave = ms * 1e6 / n * 2.4;                // cycles per instruction as seen by ONE wave (2.4 GHz nominal)
    printf("{\"geometry\": \"%s\", \"waves_per_simd\": %.2f, \"instruction\": \"%s\", \"independent_chains_per_wave\": %d, \"kernel_us\": %.1f, "
           "\"cycles_per_instruction_per_wave\": %.2f, \"cycles_per_instruction_per_simd\": %.2f}\n",
           geom, waves_per_simd, PK ? "v_pk_fma_f32" : "v_fma_f32", ILP, ms * 1e3, cyc_wave, cyc_wave / (waves_per_simd < 1 ? 1 : waves_per_simd));
    fflush(stdout);
}
template <int PK> void sweep(float* d) {
    // the env kernel's geometry at 4096 envs: 128 one-wave workgroups (half the CUs hold one wave, the rest idle)
    run<1, PK, 64>(d, 128, "128 x 64 threads", 0.125); run<2, PK, 64>(d, 128, "128 x 64 threads", 0.125);
    run<4, PK, 64>(d, 128, "128 x 64 threads", 0.125); run<8, PK, 64>(d, 128, "128 x 64 threads", 0.125);
    const int ks[4] = {1, 2, 4, 8};
    for (int k : ks) {
        char g[64]; snprintf(g, sizeof g, "%d x 256 threads", 256 * k);
        run<1, PK, 256>(d, 256 * k, g, k); run<2, PK, 256>(d, 256 * k, g, k); run<4, PK, 256>(d, 256 * k, g, k); run<8, PK, 256>(d, 256 * k, g, k);
    }
}
int main() {
    float* d; hipMalloc(&d, (size_t)2048 * 256 * 4);
    sweep<0>(d); sweep<1>(d);
    return 0;
}
