// What shader clock does a launch actually run at?  clock64() (s_memtime, shader clocks) against wall_clock64() (s_memrealtime, 100 MHz) around a
// dependent fp32 FMA chain, for the env kernel's geometry (128 one-wave workgroups), a full-chip VALU launch and a full-chip MFMA launch; each
// after idle and back to back.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void chain(float* out, long long* clk, int iters, float a, float b, int mfma) {
    const long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x * 1e-3f;
    f32x16 acc; for (int r = 0; r < 16; r++) acc[r] = x;
    if (mfma) for (int i = 0; i < iters; i++) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0); }
    else for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) x = __builtin_fmaf(x, a, b);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = x; for (int r = 0; r < 16; r++) s += acc[r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
static void run(const char* name, int blocks, int threads, int iters, int mfma, int reps, float* d, long long* dc) {
    static long long h[2 * 4096];
    for (int r = 0; r < reps; r++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, d, dc, iters, 1.0001f, 0.5f, mfma);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, dc, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
        double lo = 1e9, hi = 0, sum = 0;
        for (int b = 0; b < blocks; b++) { const double g = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1; lo = g < lo ? g : lo; hi = g > hi ? g : hi; sum += g; }
        printf("{\"launch\": \"%s\", \"rep\": %d, \"kernel_us\": %.1f, \"shader_GHz_mean\": %.3f, \"min\": %.3f, \"max\": %.3f}\n", name, r, ms * 1e3, sum / blocks, lo, hi);
        fflush(stdout);
    }
}
int main() {
    float* d; long long* dc; hipMalloc(&d, (size_t)4096 * 256 * 4); hipMalloc(&dc, sizeof(long long) * 2 * 4096);
    hipDeviceSynchronize(); usleep(500000);
    run("128 x 64 threads, fma chain, ~100 us, after idle", 128, 64, 2500, 0, 6, d, dc);
    usleep(500000);
    run("1024 x 256 threads, fma chain, ~100 us, after idle", 1024, 256, 2500, 0, 6, d, dc);
    usleep(500000);
    run("768 x 256 threads, mfma, ~100 us, after idle", 768, 256, 1200, 1, 6, d, dc);
    run("128 x 64 threads, fma chain, right after mfma launches", 128, 64, 2500, 0, 4, d, dc);
    run("768 x 256 threads, mfma, 10 ms", 768, 256, 120000, 1, 3, d, dc);
    run("128 x 64 threads, fma chain, right after a long mfma launch", 128, 64, 2500, 0, 4, d, dc);
    return 0;
}
