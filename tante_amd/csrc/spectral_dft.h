// Internal interface of the truncated-DFT spectral layer (spectral_dft.hip), called by tante_spectral_layer (operators.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int tante_spectral_dft_supported(int64_t n, int Cin, int Cout, int H, int W, int m1, int m2);
int64_t tante_spectral_dft_workspace_bytes(int64_t n, int Cin, int Cout, int H, int m1, int m2);
// out_bf16: `out` is a bf16 (n, Cout, H, W) image (tante_spectral_dft_bf16out_supported: the shape runs the split-bf16 last kernel)
int tante_spectral_dft_bf16out_supported(int64_t n, int Cin, int Cout, int H, int W, int m1, int m2);
int tante_spectral_dft_forward(const float* x, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1, int wm2, int m1,
                               int m2, const float* w0, const float* b0, int Cout, int act, float* out, void* work, int compute, hipStream_t s,
                               int out_bf16 = 0, long x_istride = 0, int out_nhwc = 0);
// x_istride: elements between consecutive images of x (0 = dense, Cin H W); out_nhwc: `out` as channels-last rows ((n h w), Cout) fp32.
// Both only on the split-bf16 kernels (bf16 compute mode); -> 1 when tante_spectral_dft_forward serves the combination
int tante_spectral_dft_x_supported(int64_t n, int Cin, int Cout, int H, int W, int m1, int m2, int strided, int out_bf16, int out_nhwc);
