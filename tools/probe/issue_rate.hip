// Calibration: cycles per VALU instruction for ONE wave per CU (the env kernel's launch geometry), dependent vs independent streams.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ILP>
__global__ __launch_bounds__(64) void fma_chain(float* out, int iters, float a, float b) {
    float x[ILP];
    for (int k = 0; k < ILP; k++) x[k] = threadIdx.x * 1e-3f + k;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int k = 0; k < ILP; k++) x[k] = __builtin_fmaf(x[k], a, b);
    }
    float s = 0;
    for (int k = 0; k < ILP; k++) s += x[k];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int ILP> void run(float* d, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 4000;
    hipLaunchKernelGGL(fma_chain<ILP>, dim3(blocks), dim3(64), 0, 0, d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(fma_chain<ILP>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 16 * ILP;
    printf("blocks %5d ILP %2d: %8.1f us, %6.2f ns per instruction = %5.2f cycles @2.4GHz\n", blocks, ILP, ms * 1e3, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
int main() {
    float* d; hipMalloc(&d, 4096 * 64 * 4);
    for (int blocks : {128, 1024, 4096}) { run<1>(d, blocks); run<2>(d, blocks); run<4>(d, blocks); run<8>(d, blocks); }
    return 0;
}
