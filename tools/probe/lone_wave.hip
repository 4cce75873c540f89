// Why does a wave alone on its CU issue dependent fp32 FMAs more slowly than one of four waves on the four SIMDs of a CU?  Same chain, different
// launch geometries; cycles per FMA from clock64() inside the kernel (shader clocks), so the DVFS state does not enter.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ILP>
__global__ void chain(float* out, long long* clk, int iters, float a, float b) {
    float x[ILP];
    for (int k = 0; k < ILP; k++) x[k] = threadIdx.x * 1e-3f + k;
    const long long c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int k = 0; k < ILP; k++) x[k] = __builtin_fmaf(x[k], a, b);
    }
    const long long c1 = clock64();
    float s = 0; for (int k = 0; k < ILP; k++) s += x[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = c1 - c0;
}
template <int ILP> void run(int blocks, int threads, float* d, long long* dc) {
    static long long h[1 << 16];
    const int iters = 2000;
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(chain<ILP>, dim3(blocks), dim3(threads), 0, 0, d, dc, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    const int nw = blocks * (threads / 64);
    hipMemcpy(h, dc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    double lo = 1e18, hi = 0, sum = 0;
    for (int w = 0; w < nw; w++) { const double c = (double)h[w] / ((double)iters * 16 * ILP); lo = c < lo ? c : lo; hi = c > hi ? c : hi; sum += c; }
    printf("{\"workgroups\": %d, \"threads\": %d, \"independent_chains\": %d, \"shader_cycles_per_fma_mean\": %.2f, \"min\": %.2f, \"max\": %.2f}\n", blocks, threads, ILP, sum / nw, lo, hi);
    fflush(stdout);
}
int main() {
    float* d; long long* dc; hipMalloc(&d, (size_t)4096 * 256 * 4); hipMalloc(&dc, sizeof(long long) * (1 << 16));
    const int geo[][2] = {{1, 64}, {1, 256}, {8, 64}, {128, 64}, {256, 64}, {512, 64}, {1024, 64}, {64, 128}, {32, 256}, {128, 256}, {256, 256}, {16, 512}, {8, 1024}};
    for (auto& g : geo) { run<1>(g[0], g[1], d, dc); run<4>(g[0], g[1], d, dc); }
    return 0;
}
