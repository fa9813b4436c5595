// SpectralLayer.forward (models/enc_dec_fno.py:184-222) as a TRUNCATED DFT on the fp32 matrix pipe.
//
//   out = act( irfft2( M . rfft2(x) [low modes only], s = (H, W), 'ortho' ) + conv1x1(x) + b0 )
//
// The layer keeps m1 x m2 modes in two row bands (rows [0, m1) and [H - m1, H), columns [0, m2)) of an H x (W/2 + 1) spectrum --
// 20 x 20 of 512 x 257 in configs/fno.yaml.  The round-1 path ran full hipFFT transforms both ways and moved the whole spectrum of
// every channel through HBM (R2C, a contraction kernel that writes zeros outside the bands, C2R, then a 1x1-conv pass: 425 us for the
// 8 -> 32 channel layer at 512 x 512, 0.04 of the HBM roofline on the layer's algorithmic bytes).  Only 2 m1 x m2 coefficients per
// channel are ever non-zero, so both transforms are skinny matrix products with cos / sin tables:
//
//   A  rows:   Ar[c,h,k]  = sum_w x[c,h,w] F[w,k]              F = [cos(2 pi j w / W) | -sin(2 pi j w / W)], k = j | m2 + j       (MFMA)
//   B  cols:   X[c,i,j]   = sum_h e^{-2 pi i r(i) h / H} (Ar[c,h,j] + i Ar[c,h,m2+j])        r(i) = the 2 m1 kept rows
//   C  mix:    Y[o,i,j]   = (1 / HW) sum_c X[c,i,j] Wt[c,o,wi(i),j]                          (both 'ortho' factors)
//   D  cols:   Z[h,k,o]   = Re | Im of sum_i e^{+2 pi i r(i) h / H} Y[o,i,j]
//   E  rows:   out[o,h,w] = act( sum_k Z[h,k,o] G[k,w] + sum_c w0[o,c] x[c,h,w] + b0[o] )    G = [a_j cos | -a_j sin], a_0 = 1, a_j = 2  (MFMA)
//
// (C2R ignores the imaginary part of the j = 0 column: so does Re(.) with a_0 = 1; m2 <= W/2 keeps the Nyquist column out.)
// Everything is fp32: v_mfma_f32_16x16x4_f32 / 32x32x2_f32 are exact fp32 fma chains, the tables come from sincospif on an exactly
// reduced integer phase, the sums have 512 terms -- the result agrees with the FFT path to ~1e-6 relative (tests: 1e-5 against the
// oracle, and against the hipFFT path of this library).  x is read twice (A and E), `out` written once, and the only intermediates
// are the compact Ar (n Cin H x 2 m2) and Z (n H x 2 m2 x Cout).  The backward pass and shapes outside the rules below stay on hipFFT.
#include "common.cuh"
#include "spectral_dft.h"
#include <algorithm>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ void sincos_frac(long num, int den, float& s, float& c) {   // angle = 2 pi num / den, num reduced exactly
  const int m = (int)(num % den);
  sincospif(2.0f * (float)m / (float)den, &s, &c);
}
// kept row i2 in [0, 2 m1) -> spectrum row
__device__ __forceinline__ int dft_row(int i2, int m1, int H) { return i2 < m1 ? i2 : H - 2 * m1 + i2; }

// ---- A: row DFT.  A wave = 32 image rows (two 16-row tiles), K = W in steps of 16 columns, N = 16 NT >= 2 m2 table columns.
// The A operand of v_mfma_f32_16x16x4_f32 is one float per lane, lane (i = l & 15, k = l >> 4).  Each lane loads 16 BYTES of its row
// (columns c0 + 4 kq .. + 3, kq = l >> 4) and feeds element s to k-step s: the k index inside a step is the lane's quarter, i.e. column
// c0 + 4 kq + s -- the table operand is read with the same permutation.  tw: [W][TS] floats in LDS, TS = 16 NT + 4 (the four k-lanes of
// a B read sit 4 rows apart: 4 TS = 16 mod 32 banks keeps the two halves of a 32-lane LDS access on different banks).
template <int NT>
__global__ __launch_bounds__(256) void dft_rows_kernel(const float* __restrict__ x, long R, int W, int m2, float* __restrict__ Ar) {
  constexpr int TS = 16 * NT + 4;
  extern __shared__ __attribute__((aligned(16))) float tw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  for (int e = tid; e < W * 16 * NT; e += 256) {
    const int w = e / (16 * NT), k = e % (16 * NT);
    float v = 0.0f;
    if (k < 2 * m2) {
      float s, c;
      sincos_frac((long)(k < m2 ? k : k - m2) * w, W, s, c);
      v = k < m2 ? c : -s;
    }
    tw[w * TS + k] = v;
  }
  __syncthreads();
  const long ntile = (R + 31) / 32;
  for (long t = (long)blockIdx.x * 4 + wave; t < ntile; t += (long)gridDim.x * 4) {
    const long row0 = t * 32;
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xr[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      long r = row0 + 16 * mt + l15;
      if (r >= R) r = R - 1;                      // clamped rows are computed and never stored
      xr[mt] = x + r * W + 4 * kq;
    }
#pragma unroll 4
    for (int c0 = 0; c0 < W; c0 += 16) {
      const f32x4 xa0 = *(const f32x4*)(xr[0] + c0), xa1 = *(const f32x4*)(xr[1] + c0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float b[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = tw[(c0 + 4 * kq + s) * TS + 16 * nt + l15];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0[s], b[nt], acc[0][nt], 0, 0, 0);
          acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1[s], b[nt], acc[1][nt], 0, 0, 0);
        }
      }
    }
    // D: column l15 = table column 16 nt + l15, rows 4 kq + r
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int k = 16 * nt + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long row = row0 + 16 * mt + 4 * kq + r;
          if (k < 2 * m2 && row < R) Ar[row * (2 * m2) + k] = acc[mt][nt][r];
        }
      }
  }
}

// ---- B: column DFT over the kept rows.  block = (kept row i2, a group of images); the 2 H twiddles of the row go to LDS once.
__global__ __launch_bounds__(256) void dft_cols_kernel(const float* __restrict__ Ar, long NC, int H, int m1, int m2, float2* __restrict__ X) {
  extern __shared__ float2 cs[];   // [H] (cos, sin) of 2 pi r h / H
  const int i2 = blockIdx.x, r = dft_row(i2, m1, H);
  for (int h = threadIdx.x; h < H; h += 256) {
    float s, c;
    sincos_frac((long)r * h, H, s, c);
    cs[h] = make_float2(c, s);
  }
  __syncthreads();
  const int per = 256 / m2;                       // images per block pass
  const int img_l = threadIdx.x / m2, j = threadIdx.x % m2;
  if (img_l >= per) return;
  for (long nc = (long)blockIdx.y * per + img_l; nc < NC; nc += (long)gridDim.y * per) {
    const float* a = Ar + nc * H * (2 * m2);
    float xr = 0.f, xi = 0.f;
    for (int h = 0; h < H; ++h) {
      const float are = a[(long)h * 2 * m2 + j], aim = a[(long)h * 2 * m2 + m2 + j];
      const float2 t = cs[h];
      xr += t.x * are + t.y * aim;                // (cos - i sin)(are + i aim)
      xi += t.x * aim - t.y * are;
    }
    X[(nc * 2 * m1 + i2) * m2 + j] = make_float2(xr, xi);
  }
}

// ---- C: channel mixing on the kept modes.  Y[n, o, i2, j] = scale sum_c X[n, c, i2, j] Wt[c, o, wi, j]
__global__ void spectral_mix_kernel(const float2* __restrict__ X, const float* __restrict__ w_re, const float* __restrict__ w_im, long n, int Cin,
                                    int Cout, int m1, int m2, int wm1, int wm2, float scale, float2* __restrict__ Y) {
  const long total = n * Cout * 2 * m1 * m2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % m2);
    long q = idx / m2;
    const int i2 = (int)(q % (2 * m1)); q /= 2 * m1;
    const int o = (int)(q % Cout);
    const long b = q / Cout;
    const int wi = i2 < m1 ? i2 : i2 - m1;
    float ar = 0.f, ai = 0.f;
    for (int c = 0; c < Cin; ++c) {
      const float2 xv = X[((b * Cin + c) * 2 * m1 + i2) * m2 + j];
      const long wo = (((long)c * Cout + o) * wm1 + wi) * wm2 + j;
      const float wr = w_re[wo], wim = w_im[wo];
      ar += xv.x * wr - xv.y * wim;
      ai += xv.x * wim + xv.y * wr;
    }
    Y[idx] = make_float2(ar * scale, ai * scale);
  }
}

// ---- D: inverse column DFT.  block = (image n, 16 rows h); a thread owns (o, j), keeps its 2 m1 coefficients in registers and walks
// the block's rows.  Z[n][h][k][o] (o innermost: the A operand rows of kernel E are contiguous), k = j (real part) | m2 + j (imaginary).
template <int M1X2>
__global__ __launch_bounds__(256) void idft_cols_kernel(const float2* __restrict__ Y, int H, int Cout, int m2, float* __restrict__ Z) {
  constexpr int HB = 16;
  __shared__ float2 cs[HB][M1X2];
  const int m1 = M1X2 / 2;
  const long n = blockIdx.y;
  const int h0 = blockIdx.x * HB;
  for (int e = threadIdx.x; e < HB * M1X2; e += 256) {
    const int hh = e / M1X2, i2 = e % M1X2;
    float s, c;
    sincos_frac((long)dft_row(i2, m1, H) * (h0 + hh), H, s, c);
    cs[hh][i2] = make_float2(c, s);
  }
  __syncthreads();
  for (int p = threadIdx.x; p < Cout * m2; p += 256) {
    const int o = p % Cout, j = p / Cout;
    float2 y[M1X2];
#pragma unroll
    for (int i2 = 0; i2 < M1X2; ++i2) y[i2] = Y[((n * Cout + o) * M1X2 + i2) * m2 + j];
    for (int hh = 0; hh < HB && h0 + hh < H; ++hh) {
      float zr = 0.f, zi = 0.f;
#pragma unroll
      for (int i2 = 0; i2 < M1X2; ++i2) {
        const float2 t = cs[hh][i2];
        zr += t.x * y[i2].x - t.y * y[i2].y;      // (cos + i sin)(yr + i yi)
        zi += t.y * y[i2].x + t.x * y[i2].y;
      }
      float* z = Z + ((n * H + h0 + hh) * 2 * m2) * Cout;
      z[(long)j * Cout + o] = zr;
      z[(long)(m2 + j) * Cout + o] = zi;
    }
  }
}

// ---- E: inverse row DFT + 1x1 conv + bias + activation.  Workgroup = W / 32 waves, wave = 32 output columns; a workgroup walks image
// rows (n, h).  v_mfma_f32_32x32x2_f32: A (32 x 2) lane (i = l & 31, k = l >> 5), B (2 x 32) lane (k = l >> 5, j = l & 31), D 16 registers:
// column l & 31, row (g & 3) + 8 (g >> 2) + 4 (l >> 5).  M = 32 output channels, N = 32 columns, K = 2 m2 table rows, then Cin channels:
//   A = Z[n][h][k][o] (128 contiguous bytes per k)       B = G[k][w] (LDS table of the workgroup, read 32 floats per half wave)
//   A = w0[o][c]                                           B = x[n][c][h][w] (128 contiguous bytes per channel)
// and the accumulator leaves as 16 stores of two 128-byte runs.
__device__ __forceinline__ float act_fast(float v, int act) {
  switch (act) {
    case TANTE_ACT_GELU_ERF: return gelu_erf_fast(v);      // |error| <= 1.5e-7 against erff: far inside the fp32 bar, a third of its cost
    case TANTE_ACT_NONE: return v;
    default: return apply_act(v, act);
  }
}
__global__ __launch_bounds__(1024) void idft_rows_conv_kernel(const float* __restrict__ Z, const float* __restrict__ x, const float* __restrict__ w0,
                                                              const float* __restrict__ b0, long n, int Cin, int Cout, int H, int W, int m2, int act,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float G[];   // [2 m2][W]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kk = lane >> 5;
  for (int e = tid; e < 2 * m2 * W; e += blockDim.x) {
    const int k = e / W, w = e % W, j = k < m2 ? k : k - m2;
    float s, c;
    sincos_frac((long)j * w, W, s, c);
    const float a = j == 0 ? 1.0f : 2.0f;
    G[e] = k < m2 ? a * c : -a * s;
  }
  __syncthreads();
  const int w0c = 32 * wave;
  const long rows = n * H;
  const int otiles = (Cout + 31) / 32;
  for (long rho = blockIdx.x; rho < rows; rho += gridDim.x) {
    const long b = rho / H;
    const int h = (int)(rho - b * H);
    const float* zrow = Z + rho * 2 * m2 * Cout;
    for (int ot = 0; ot < otiles; ++ot) {
      const int o = 32 * ot + l31;                 // this lane's A-operand row (output channel)
      const bool olive = o < Cout;
      f32x16_t acc;
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
#pragma unroll 4
      for (int s = 0; s < m2; ++s) {
        const int k = 2 * s + kk;
        const float a = olive ? zrow[(long)k * Cout + o] : 0.0f;
        const float bv = G[k * W + w0c + l31];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
      }
      const float* xp = x + ((b * Cin) * H + h) * (long)W + w0c + l31;
#pragma unroll 4
      for (int s = 0; s < (Cin + 1) / 2; ++s) {
        const int c = 2 * s + kk;
        const bool cl = c < Cin;
        const float a = (olive && cl) ? w0[(long)o * Cin + c] : 0.0f;
        const float bv = cl ? xp[(long)c * H * W] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int oo = 32 * ot + (g & 3) + 8 * (g >> 2) + 4 * kk;
        if (oo < Cout) {
          const float v = acc[g] + (b0 ? b0[oo] : 0.0f);
          out[((b * Cout + oo) * H + h) * (long)W + w0c + l31] = act_fast(v, act);
        }
      }
    }
  }
}

}  // namespace

int tante_spectral_dft_supported(int64_t n, int Cin, int Cout, int H, int W, int m1, int m2) {
  if (n <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if (W % 32 || W < 32 || W > 512 || H % 16) return 0;                  // whole 32-column waves, at most 16 of them; 16-row tiles
  if (m1 < 1 || m2 < 1 || 2 * m1 > H || m2 > W / 2 || m2 > 32) return 0;   // disjoint row bands, no Nyquist column, <= 4 table tiles
  switch (2 * m1) { case 4: case 8: case 10: case 16: case 20: case 32: case 40: case 64: break; default: return 0; }   // idft_cols_kernel instantiations
  const int NT = (2 * m2 + 15) / 16;
  if ((size_t)W * (16 * NT + 4) * 4 > 150 * 1024) return 0;             // kernel A's table
  if ((size_t)2 * m2 * W * 4 > 150 * 1024) return 0;                    // kernel E's table
  return 1;
}

int64_t tante_spectral_dft_workspace_bytes(int64_t n, int Cin, int Cout, int H, int m1, int m2) {
  const int64_t ar = n * Cin * H * 2 * m2 * 4, xx = n * Cin * 2 * m1 * m2 * 8, yy = n * Cout * 2 * m1 * m2 * 8, zz = n * H * 2 * m2 * Cout * 4;
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  return up(ar) + up(xx) + up(yy) + up(zz);
}

int tante_spectral_dft_forward(const float* x, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1, int wm2, int m1,
                               int m2, const float* w0, const float* b0, int Cout, int act, float* out, void* work, hipStream_t s) {
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  char* p = (char*)work;
  float* Ar = (float*)p; p += up(n * Cin * H * 2 * m2 * 4);
  float2* X = (float2*)p; p += up(n * Cin * 2 * m1 * m2 * 8);
  float2* Y = (float2*)p; p += up(n * Cout * 2 * m1 * m2 * 8);
  float* Z = (float*)p;
  const long R = (long)n * Cin * H;
  const int NT = (2 * m2 + 15) / 16;
  const size_t ldsA = (size_t)W * (16 * NT + 4) * 4;
  const long tiles = (R + 31) / 32;
  const unsigned gridA = (unsigned)std::min<long>(256, (tiles + 3) / 4);
  static TantePerDevice attrA[4], attrE;
#define TANTE_DFT_A(NTV)                                                                                                          \
  case NTV:                                                                                                                       \
    attrA[NTV - 1].once([&] { (void)hipFuncSetAttribute((const void*)dft_rows_kernel<NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }); \
    hipLaunchKernelGGL(dft_rows_kernel<NTV>, dim3(gridA), dim3(256), ldsA, s, x, R, W, m2, Ar);                                  \
    break;
  switch (NT) { TANTE_DFT_A(1) TANTE_DFT_A(2) TANTE_DFT_A(3) TANTE_DFT_A(4) default: return -2; }
#undef TANTE_DFT_A
  const long NC = (long)n * Cin;
  const int per = 256 / m2;
  hipLaunchKernelGGL(dft_cols_kernel, dim3(2 * m1, (unsigned)std::min<long>(64, (NC + per - 1) / per)), dim3(256), (size_t)H * 8, s, Ar, NC, H, m1, m2, X);
  const long totalC = (long)n * Cout * 2 * m1 * m2;
  hipLaunchKernelGGL(spectral_mix_kernel, dim3((unsigned)std::min<long>(4096, (totalC + 255) / 256)), dim3(256), 0, s, X, w_re, w_im, (long)n, Cin, Cout,
                     m1, m2, wm1, wm2, 1.0f / ((float)H * (float)W), Y);
  const dim3 gridD((unsigned)((H + 15) / 16), (unsigned)n);
  switch (2 * m1) {
#define TANTE_DFT_D(V) case V: hipLaunchKernelGGL(idft_cols_kernel<V>, gridD, dim3(256), 0, s, Y, H, Cout, m2, Z); break;
    TANTE_DFT_D(4) TANTE_DFT_D(8) TANTE_DFT_D(10) TANTE_DFT_D(16) TANTE_DFT_D(20) TANTE_DFT_D(32) TANTE_DFT_D(40) TANTE_DFT_D(64)
#undef TANTE_DFT_D
    default: return -2;
  }
  const size_t ldsE = (size_t)2 * m2 * W * 4;
  attrE.once([&] { (void)hipFuncSetAttribute((const void*)idft_rows_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
  const long rows = (long)n * H;
  // each workgroup builds the table once (2 m2 W sincos), so it should walk many rows; W / 32 waves per workgroup
  const int wpb = W / 32;
  const long wgs = std::min<long>(rows, 256L * std::max(1, 16 / wpb));
  hipLaunchKernelGGL(idft_rows_conv_kernel, dim3((unsigned)wgs), dim3(64 * wpb), ldsE, s, Z, x, w0, b0, (long)n, Cin, Cout, H, W, m2, act, out);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
