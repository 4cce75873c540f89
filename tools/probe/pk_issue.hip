// What does ONE wave per SIMD pay for packed fp32 (v_pk_*) against scalar fp32 VALU, and what do two waves pay?  Decides the shape of the
// packed ABA kernel (round 5): one env per lane with the two legs in the halves of 64-bit register pairs needs ~450 registers, i.e. one wave
// per SIMD.  Shader cycles from clock64() inside the kernel (DVFS does not enter); 8 independent chains per wave, operands in distinct VGPRs.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/pk_issue.hip -o gpurun_out/pk_issue && gpurun_out/pk_issue > gpurun_out/pk_issue.jsonl
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
enum { FMA = 0, PK_FMA = 1, PK_MUL = 2, PK_ADD = 3, MIX_2PK_1S = 4, PK_FMA_SGPR = 5, MIX_1PK_1S = 6, PK_FMA_DEP2 = 7, MOV_ACC = 8, FMAC_E32 = 9, MUL_E32 = 10, MIX_PK_E32 = 11, PK_FMA_LIT = 12 };
template <int MODE, int ILP>
__global__ __launch_bounds__(256) void chain(float* out, long long* clk, int iters, float a, float b) {
    f2 x[ILP];
    float y[ILP];
    for (int k = 0; k < ILP; k++) { x[k] = f2{threadIdx.x * 1e-3f + k, threadIdx.x * 2e-3f - k}; y[k] = threadIdx.x * 3e-3f + k; }
    f2 a2 = {a + threadIdx.x * 1e-9f, a}, b2 = {b, b + threadIdx.x * 1e-9f};  // VGPR pairs
    float a1 = a + threadIdx.x * 1e-9f, b1 = b + threadIdx.x * 1e-9f;
    const f2 as = {a, a};
    const long long c0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int k = 0; k < ILP; k++) {
                if (MODE == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[k]) : "v"(a1), "v"(b1));
                if (MODE == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a2), "v"(b2));
                if (MODE == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a2));
                if (MODE == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(b2));
                if (MODE == PK_FMA_SGPR) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "s"(as), "v"(b2));
                if (MODE == MIX_2PK_1S) {
                    if (k % 3 == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[k]) : "v"(a1), "v"(b1));
                    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a2), "v"(b2));
                }
                if (MODE == MIX_1PK_1S) {
                    if (k % 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y[k]) : "v"(a1), "v"(b1));
                    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a2), "v"(b2));
                }
                if (MODE == PK_FMA_DEP2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[k]) : "v"(a2), "v"(b2));  // accumulate form
                if (MODE == FMAC_E32) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(y[k]) : "v"(a1), "v"(b1));
                if (MODE == MUL_E32) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(y[k]) : "v"(a1));
                if (MODE == MIX_PK_E32) {
                    if (k % 2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(y[k]) : "v"(a1), "v"(b1));
                    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a2), "v"(b2));
                }
                if (MODE == PK_FMA_LIT) asm volatile("v_pk_fma_f32 %0, %0, %1, 1.0 op_sel_hi:[1,1,0]" : "+v"(x[k]) : "v"(a2));
                if (MODE == MOV_ACC) asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0" : "+v"(y[k]) : : "a0");
            }
    }
    const long long c1 = clock64();
    float s = 0;
    for (int k = 0; k < ILP; k++) s += x[k].x + x[k].y + y[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = c1 - c0;
}
static const char* NAMES[] = {"v_fma_f32 (3 VGPR)", "v_pk_fma_f32 (3 VGPR pairs)", "v_pk_mul_f32", "v_pk_add_f32", "2 v_pk_fma : 1 v_fma", "v_pk_fma_f32 (SGPR pair source)",
                              "1 v_pk_fma : 1 v_fma", "v_pk_fma_f32 accumulate form", "accvgpr write + read (2 instructions)", "v_fmac_f32_e32 (VOP2, 4-byte encoding)", "v_mul_f32_e32 (VOP2)", "1 v_pk_fma : 1 v_fmac_e32", "v_pk_fma_f32 with an inline constant"};
template <int MODE, int ILP> void run(int waves_per_simd, float* d, long long* dc) {
    static long long h[1 << 16];
    const int iters = 1000, blocks = 256 * waves_per_simd;
    // ~60 ms of the same kernel first (the clock under a new full-chip kernel is a transient for ~40 ms, tools/aba_series.py), then 5 timed launches:
    // wall time per instruction from HIP events next to the in-kernel counter
    for (int r = 0; r < 150; r++) hipLaunchKernelGGL((chain<MODE, ILP>), dim3(blocks), dim3(256), 0, 0, d, dc, iters, 1.0001f, 0.5f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((chain<MODE, ILP>), dim3(blocks), dim3(256), 0, 0, d, dc, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int nw = blocks * 4;
    hipMemcpy(h, dc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int w = 0; w < nw; w++) sum += (double)h[w] / ((double)iters * 16 * ILP);
    const double per_wave = sum / nw;
    const double ns_per_simd = ms / 5 * 1e6 / ((double)iters * 16 * ILP * waves_per_simd);  // launch overhead (~5 us of ~400+) included
    printf("{\"mode\": \"%s\", \"independent_chains\": %d, \"waves_per_simd\": %d, \"counter_ticks_per_instruction_per_wave\": %.2f, \"ticks_per_simd\": %.2f, "
           "\"wall_ns_per_instruction_per_simd\": %.3f, \"kernel_us\": %.1f}\n", NAMES[MODE], ILP, waves_per_simd, per_wave, per_wave / waves_per_simd, ns_per_simd, ms / 5 * 1e3);
    fflush(stdout);
}
template <int MODE> void sweep(float* d, long long* dc) {
    for (int w = 1; w <= 2; w++) { run<MODE, 4>(w, d, dc); run<MODE, 8>(w, d, dc); }
}
int main() {
    float* d; long long* dc; hipMalloc(&d, (size_t)4096 * 256 * 4); hipMalloc(&dc, sizeof(long long) * (1 << 16));
    sweep<FMA>(d, dc); sweep<FMAC_E32>(d, dc); sweep<MUL_E32>(d, dc); sweep<PK_FMA>(d, dc); sweep<PK_MUL>(d, dc); sweep<MIX_PK_E32>(d, dc);
    run<FMAC_E32, 16>(1, d, dc); run<PK_FMA, 16>(1, d, dc); run<FMAC_E32, 16>(2, d, dc); run<PK_FMA, 16>(2, d, dc);
    run<FMAC_E32, 8>(4, d, dc); run<PK_FMA, 8>(4, d, dc); run<MUL_E32, 8>(4, d, dc);
    return 0;
}
