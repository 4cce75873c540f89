// Descriptors of the grouped weight-gradient launch, shared by bg_wgrad.hip (fp32 MFMA) and bg_wgrad_split.hip (split bf16 MFMA).
#pragma once
#include <hip/hip_runtime.h>

constexpr int WG_MAX_PROBLEMS = 8;
struct WgradProblem {
    const float* G; const float* A; float* P; float* dW;
    int M, Cout, Cin, Cin_real, tci, ntile_ci, ntiles, tw, slices, wg_begin, fin_begin, n4;
};
struct WgradGroup { int np; WgradProblem p[WG_MAX_PROBLEMS]; };

