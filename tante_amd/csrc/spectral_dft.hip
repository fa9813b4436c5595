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
#include "common.hip.h"
#include "spectral_dft.h"
#include <algorithm>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// (every product this file reduces -- mode index x position, kept row x row -- is below 2^31: a 32-bit unsigned remainder, a mask for the
// power-of-two sizes.  As a 64-bit signed remainder it was ~80 instructions per table entry, most of kernel A's 40 us at W = 512.)
__device__ __forceinline__ unsigned mod_u32(unsigned v, unsigned den) { return (den & (den - 1)) == 0 ? (v & (den - 1)) : v % den; }
__device__ __forceinline__ void sincos_frac(long num, int den, float& s, float& c) {   // angle = 2 pi num / den, num reduced exactly
  const int m = (int)mod_u32((unsigned)num, (unsigned)den);
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
  float2* base = (float2*)(tw + W * TS);     // (cos, sin) of 2 pi m / W, m = 0 .. W - 1: W evaluations, the table is a gather of them
  for (int m = tid; m < W; m += 256) {
    float s, c;
    sincos_frac(m, W, s, c);
    base[m] = make_float2(c, s);
  }
  __syncthreads();
  for (int e = tid; e < W * 16 * NT; e += 256) {
    const int w = e / (16 * NT), k = e % (16 * NT);
    float v = 0.0f;
    if (k < 2 * m2) {
      const float2 t = base[mod_u32((unsigned)((k < m2 ? k : k - m2) * w), (unsigned)W)];
      v = k < m2 ? t.x : -t.y;
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

// ---- B: column DFT over the kept rows.  A WAVE per (image, kept row): lane l sums rows h = l, l + 64, ... for all m2 columns (a row of Ar is
// 2 m2 contiguous floats), then the 2 m2 partial sums are reduced across the wave.  (A thread per output with a 512-step serial loop
// over h, the first version, was 35 us of pure latency.)
template <int M2>
__global__ __launch_bounds__(256) void dft_cols_kernel(const float* __restrict__ Ar, long NC, int H, int m1, float2* __restrict__ X) {
  const int lane = threadIdx.x & 63;
  const long unit = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= NC * 2 * m1) return;
  const long nc = unit / (2 * m1);
  const int i2 = (int)(unit - nc * 2 * m1), r = dft_row(i2, m1, H);
  const float* a = Ar + nc * H * (2 * M2);
  float xr[M2], xi[M2];
#pragma unroll
  for (int j = 0; j < M2; ++j) xr[j] = xi[j] = 0.0f;
  for (int h = lane; h < H; h += 64) {
    float sn, cs;
    sincos_frac((long)r * h, H, sn, cs);
    const float* row = a + (long)h * 2 * M2;
#pragma unroll
    for (int j = 0; j < M2; ++j) {
      const float are = row[j], aim = row[M2 + j];
      xr[j] += cs * are + sn * aim;               // (cos - i sin)(are + i aim)
      xi[j] += cs * aim - sn * are;
    }
  }
#pragma unroll
  for (int j = 0; j < M2; ++j) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      xr[j] += __shfl_xor(xr[j], o);
      xi[j] += __shfl_xor(xi[j], o);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < M2; ++j) X[(nc * 2 * m1 + i2) * M2 + j] = make_float2(xr[j], xi[j]);
  }
}

// ---- C: channel mixing on the kept modes.  Y[n, i2, j, o] = scale sum_c X[n, c, i2, j] Wt[c, o, wi, j]
__global__ void spectral_mix_kernel(const float2* __restrict__ X, const float* __restrict__ w_re, const float* __restrict__ w_im, long n, int Cin,
                                    int Cout, int m1, int m2, int wm1, int wm2, float scale, float2* __restrict__ Y, int CS) {
  // CS (a power of two <= 16) neighbouring lanes share one output and split its sum over the input channels: with few kept modes the
  // launch has only n Cout 2 m1 m2 outputs (6 400 at 128 x 128 with 5 x 5 modes), and one thread walking all 128 channels of its
  // output -- three dependent-latency loads per step -- made a 25-workgroup launch of 55 us
  const long total = n * Cout * 2 * m1 * m2;
  const long nthr = (long)gridDim.x * blockDim.x / CS;
  const int seg = threadIdx.x & (CS - 1);
  for (long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) / CS; idx < (total + nthr - 1) / nthr * nthr; idx += nthr) {
    const bool live = idx < total;
    const long id = live ? idx : 0;
    const int j = (int)(id % m2);
    long q = id / m2;
    const int i2 = (int)(q % (2 * m1)); q /= 2 * m1;
    const int o = (int)(q % Cout);
    const long b = q / Cout;
    const int wi = i2 < m1 ? i2 : i2 - m1;
    float ar = 0.f, ai = 0.f;
#pragma unroll 8
    for (int c = seg; c < Cin; c += CS) {
      const float2 xv = X[((b * Cin + c) * 2 * m1 + i2) * m2 + j];
      const long wo = (((long)c * Cout + o) * wm1 + wi) * wm2 + j;
      const float wr = w_re[wo], wim = w_im[wo];
      ar += xv.x * wr - xv.y * wim;
      ai += xv.x * wim + xv.y * wr;
    }
    for (int d = 1; d < CS; d <<= 1) {
      ar += __shfl_xor(ar, d);
      ai += __shfl_xor(ai, d);
    }
    if (live && seg == 0)
      Y[((b * 2 * m1 + i2) * m2 + j) * Cout + o] = make_float2(ar * scale, ai * scale);      // [n][i2][j][o]: kernel D's lanes run over o
  }
}

// ---- D: inverse column DFT.  block = (image n, DFT_D_HB rows h); a thread owns (o, j), keeps its 2 m1 coefficients in registers and walks
// the block's rows.  Z[n][h][k][o] (o innermost: the A operand rows of kernel E are contiguous), k = j (real part) | m2 + j (imaginary).
template <int M1X2>
__global__ __launch_bounds__(256) void idft_cols_kernel(const float2* __restrict__ Y, int H, int Cout, int m2, float* __restrict__ Z) {
#ifndef DFT_D_HB
#define DFT_D_HB 2      // rows per workgroup: a thread's pass is a serial chain over its rows (8 rows: 31 us per launch at cfg5 on 128 workgroups;
#endif                  // 4 / 2 / 1 rows: cfg5 3 080 -> 3 124 / 3 150 / 3 137 frames/s, profiles/r05_ab_fno_idft_cols_rows.log)
  constexpr int HB = DFT_D_HB;
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
    for (int i2 = 0; i2 < M1X2; ++i2) y[i2] = Y[((n * M1X2 + i2) * m2 + j) * Cout + o];
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
template <int M2>
__global__ __launch_bounds__(1024) void idft_rows_conv_kernel(const float* __restrict__ Z, const float* __restrict__ x, const float* __restrict__ w0,
                                                              const float* __restrict__ b0, long n, int Cin, int Cout, int H, int W, int m2, int act,
                                                              float* __restrict__ out) {
  // M2 = m2 when it is a shipped size (a row's coefficients are then loaded ten at a time ahead of their MFMAs), 0 = any.
  // Measured alternatives (round 3, cfg5, us per launch at 512 x 512 / 128 x 128): this form 139 / 69; 8-wave column blocks with two
  // output tiles' accumulators live and 192 registers 168 / 64; the same with the next row's coefficients prefetched into a second
  // register set: > 256 registers, spills.  The floor is the fp32 matrix pipe itself: 24 MFMAs x 64 cycles per 32 x 32 output tile
  // = 50 us for the 8 -> 32 channel layer at 512 x 512 (157 TFLOP/s), beside 56 us of HBM time for its 335 MB.
  extern __shared__ __attribute__((aligned(16))) float G[];   // [2 m2][W], then w0 transposed [Cin][CP] (zero-padded to 32 channels)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kk = lane >> 5;
  const int CP = (Cout + 31) / 32 * 32;
  float* wt = G + 2 * m2 * W;
  for (int e = tid; e < 2 * m2 * W; e += blockDim.x) {
    const int k = e / W, w = e % W, j = k < m2 ? k : k - m2;
    float s, c;
    sincos_frac((long)j * w, W, s, c);
    const float a = j == 0 ? 1.0f : 2.0f;
    G[e] = k < m2 ? a * c : -a * s;
  }
  // (global side contiguous: with o fastest every lane fetched its own 64-byte segment -- for the 64 -> 128 channel layers that
  // prologue, repeated by every one-row workgroup, was most of the launch)
  for (int e = tid; e < Cout * Cin; e += blockDim.x) {
    const int o = e / Cin, c = e - o * Cin;
    wt[c * CP + o] = w0[e];
  }
  for (int e = tid; e < Cin * (CP - Cout); e += blockDim.x) {
    const int c = e / (CP - Cout), o = Cout + e - c * (CP - Cout);
    wt[c * CP + o] = 0.0f;
  }
  __syncthreads();
  const int w0c = 32 * wave;
  const long rows = n * H;
  const long HWl = (long)H * W;
  for (long rho = blockIdx.x; rho < rows; rho += gridDim.x) {
    const long b = rho / H;
    const int h = (int)(rho - b * H);
    const float* zrow = Z + rho * 2 * m2 * Cout;
    const float* xp = x + ((b * Cin) * H + h) * (long)W + w0c + l31;
    for (int og = 0; og < Cout; og += 32) {
      const int o = og + l31;                      // this lane's A-operand row (output channel)
      const bool olive = o < Cout;
      f32x16_t acc;
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
      if constexpr (M2 > 0) {
        constexpr int CH = M2 > 5 ? 5 : M2;        // in chunks of 5: 16 waves per workgroup leave 128 registers per lane (10 spills)
#pragma unroll
        for (int s0 = 0; s0 < M2; s0 += CH) {
          float av[CH];
#pragma unroll
          for (int sidx = 0; sidx < CH; ++sidx) av[sidx] = (olive && s0 + sidx < M2) ? zrow[(long)(2 * (s0 + sidx) + kk) * Cout + o] : 0.0f;
#pragma unroll
          for (int sidx = 0; sidx < CH; ++sidx)
            if (s0 + sidx < M2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sidx], G[(2 * (s0 + sidx) + kk) * W + w0c + l31], acc, 0, 0, 0);
        }
      } else {
#pragma unroll 4
        for (int sidx = 0; sidx < m2; ++sidx) {
          const int k = 2 * sidx + kk;
          const float a = olive ? zrow[(long)k * Cout + o] : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, G[k * W + w0c + l31], acc, 0, 0, 0);
        }
      }
      for (int c0 = 0; c0 < Cin; c0 += 8) {        // 1x1 conv: 8 channels (4 loads of 128 contiguous bytes per half wave) at a time
        float xv[4];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
          const int c = c0 + 2 * sidx + kk;
          xv[sidx] = c < Cin ? xp[(long)c * HWl] : 0.0f;
        }
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
          const int c = c0 + 2 * sidx + kk;
          if (c0 + 2 * sidx < Cin) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c < Cin ? wt[c * CP + o] : 0.0f, xv[sidx], acc, 0, 0, 0);
        }
      }
      float* orow = out + (b * Cout * H + h) * (long)W + w0c + l31;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int oo = og + (g & 3) + 8 * (g >> 2) + 4 * kk;
        if (oo < Cout) orow[(long)oo * HWl] = act_fast(acc[g] + (b0 ? b0[oo] : 0.0f), act);
      }
    }
  }
}

// ---- E in the bf16 compute mode: the same products on the bf16 matrix pipe with SPLIT operands -------------------------------------------
// Kernel E above does 24 v_mfma_f32_32x32x2_f32 of 64 cycles per 32 x 32 output tile: a matrix floor of ~49 us for the 8 -> 32 channel
// layer of a 4-frame window at 512 x 512 (n = 8), beside 42 us of HBM time for its 335 MB -- and runs 229 us (one wave per 32 columns,
// operands fetched right in front of their MFMAs).  The bf16 pipe is 16 x faster per product; with every operand split into two bf16
// parts, v = hi + lo (16 mantissa bits kept), and the three products
//     a . b  ~=  a_hi b_hi + a_lo b_hi + a_hi b_lo                 (a_lo b_lo, 2^-18 relative, dropped; fp32 accumulation)
// the result differs from the fp32 chain by ~1e-5 relative per term -- a thousandth of the bf16 mode's own bar (1e-2; its convolutions
// round activations to 8 bits) -- for 9 v_mfma_f32_16x16x32_bf16 of 16 cycles per 16 x 16 tile: 2.7 x fewer matrix cycles (18 us at
// n = 8), so that the data movement is what is left to organise.  Only the bf16 compute mode uses it (tante_spectral_layer_c); fp32
// keeps the exact kernel above.
// A workgroup = 128 output columns; its G table sits in LDS as ready B-operand fragments (hi and lo).  A wave owns one image row at a
// time: it gathers and splits the row's Z coefficients once (A fragments), then walks the 8 column tiles, splitting x on the way.
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - bf16_lo(hi), b - bf16_hi(hi));
}
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned h, l;
    split_pair(v[2 * q], v[2 * q + 1], h, l);
    hi[q] = h;
    lo[q] = l;
  }
}
__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// OUT16: the (n, Cout, H, W) image is written as bf16 (round to nearest even, what the consumer's own conversion would do): for a layer whose
// output only feeds the bf16 patch gather of the convolution behind it (enc_FNO's first stage: 268 MB of fp32 written here and read
// back by tante_im2col at cfg5) both passes move half the bytes and the result is the same bit for bit.
template <int KZS, int KCS, bool OUT16 = false>   // k-steps of 32 per term: ceil(2 m2 / 32) spectral, ceil(Cin / 32) conv
__global__ __launch_bounds__(256) void idft_rows_conv_x3_kernel(const float* __restrict__ Z, const float* __restrict__ x, const float* __restrict__ w0,
                                                                const float* __restrict__ b0, long n, int Cin, int Cout, int H, int W, int m2,
                                                                int act, float* __restrict__ out, long x_istride) {
  // LDS: the G table as operand fragments [hi | lo][KZS][8 column tiles][64 lanes], then per wave a staging area for one image row's
  // operands in ROW form: zs[2 m2][Cout + 4] (the row's coefficients) and xs[Cin][132] (its 128 pixels of every input channel).
  // Every global access of the row loop is a 16-byte-per-lane access of whole contiguous runs; the operand (MFMA) layouts are taken
  // from LDS.  (The first version gathered both operands straight from global memory, 64-byte pieces at a time: 160 such
  // instructions per row and wave kept the texture-address unit busy for longer than the products and the HBM traffic together.)
  extern __shared__ __attribute__((aligned(16))) u32x4 tbl[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kk = lane >> 4;
  const int ncb = W / 128, cb = blockIdx.x % ncb, rg = blockIdx.x / ncb, nrg = gridDim.x / ncb, col0 = cb * 128;
  const int ZP = Cout + 4, XP = 132;
  float* stage = (float*)(tbl + 2 * KZS * 512) + wave * (2 * m2 * ZP + Cin * XP);
  float* zs = stage;
  float* xs = stage + 2 * m2 * ZP;
  {
    // (cos, sin)(2 pi m / W) once per workgroup (W evaluations, in the staging area, which is not in use yet); the table is a gather of them
    float2* base = (float2*)((float*)(tbl + 2 * KZS * 512));
    for (int m = tid; m < W; m += 256) {
      float sn, cs;
      sincos_frac(m, W, sn, cs);
      base[m] = make_float2(cs, sn);
    }
    __syncthreads();
    for (int e = tid; e < KZS * 512; e += 256) {
      const int s = e >> 9, wt = (e >> 6) & 7, ln = e & 63, w = col0 + 16 * wt + (ln & 15);
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = 32 * s + 8 * (ln >> 4) + q;
        v[q] = 0.0f;
        if (k < 2 * m2) {
          const int j = k < m2 ? k : k - m2;
          const float2 t = base[mod_u32((unsigned)(j * w), (unsigned)W)];
          const float a = j == 0 ? 1.0f : 2.0f;
          v[q] = k < m2 ? a * t.x : -a * t.y;
        }
      }
      u32x4 hi, lo;
      split8(v, hi, lo);
      tbl[e] = hi;
      tbl[KZS * 512 + e] = lo;
    }
    __syncthreads();
  }
  // the 1x1 conv's weights as operand fragments (the same for every row): lane (o = l15 of tile ot, kk) holds channels 32 sc + 8 kk ..
  u32x4 Wc[2][KCS][2];
#pragma unroll
  for (int sc = 0; sc < KCS; ++sc)
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      const int o = 16 * ot + l15;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = 32 * sc + 8 * kk + q;
        v[q] = (o < Cout && c < Cin) ? w0[(long)o * Cin + c] : 0.0f;
      }
      split8(v, Wc[0][sc][ot], Wc[1][sc][ot]);
    }
  float bias[2];
#pragma unroll
  for (int ot = 0; ot < 2; ++ot) bias[ot] = (b0 && 16 * ot + l15 < Cout) ? b0[16 * ot + l15] : 0.0f;
  __syncthreads();
  const long rows = n * H, HWl = (long)H * W;
  const bool two = Cout > 16;
  const int nz4 = 2 * m2 * Cout / 4;              // float4 pieces of a coefficient row (Cout % 4 == 0)
  for (long rho = (long)rg * 4 + wave; rho < rows; rho += (long)nrg * 4) {
    const long b = rho / H;
    const int h = (int)(rho - b * H);
    // global (16 bytes per lane, whole contiguous runs) -> registers -> the wave's staging area; LDS operations of one wave execute in order
    {
      const float* zrow = Z + rho * 2 * m2 * Cout;
      for (int i0 = 0; 64 * i0 < nz4; i0 += 4) {
        f32x4 zr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = lane + 64 * (i0 + u);
          zr[u] = idx < nz4 ? *(const f32x4*)(zrow + 4 * idx) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = lane + 64 * (i0 + u);
          if (idx < nz4) { const int k = (4 * idx) / Cout, o = 4 * idx - k * Cout; *(f32x4*)(zs + k * ZP + o) = zr[u]; }
        }
      }
      const float* xp = x + b * x_istride + (long)h * W + col0 + 4 * (lane & 31);
      for (int c0 = 0; c0 < Cin; c0 += 8) {         // two channels (2 x 512 bytes) per wave instruction
        f32x4 xr4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + 2 * u + (lane >> 5);
          xr4[u] = c < Cin ? *(const f32x4*)(xp + (long)c * HWl) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + 2 * u + (lane >> 5);
          if (c < Cin) *(f32x4*)(xs + c * XP + 4 * (lane & 31)) = xr4[u];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    u32x4 Zf[2][KZS][2];
#pragma unroll
    for (int s = 0; s < KZS; ++s)
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        const int o = 16 * ot + l15;
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = 32 * s + 8 * kk + q;
          v[q] = (o < Cout && k < 2 * m2) ? zs[k * ZP + o] : 0.0f;
        }
        split8(v, Zf[0][s][ot], Zf[1][s][ot]);
      }
    float* orow = out + (b * Cout * H + h) * (long)W + col0 + 4 * kk;
#pragma unroll 1
    for (int wt = 0; wt < 8; ++wt) {
      u32x4 Xf[2][KCS];
#pragma unroll
      for (int sc = 0; sc < KCS; ++sc) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int c = 32 * sc + 8 * kk + q;
          v[q] = c < Cin ? xs[c * XP + 16 * wt + l15] : 0.0f;
        }
        split8(v, Xf[0][sc], Xf[1][sc]);
      }
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};     // D[w][o]: lane (o = l15, kk) holds columns 4 kk .. 4 kk + 3 of the tile
#pragma unroll
      for (int s = 0; s < KZS; ++s) {
        const u32x4 gh = tbl[s * 512 + wt * 64 + lane], gl = tbl[(KZS + s) * 512 + wt * 64 + lane];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
          if (ot == 1 && !two) break;
          acc[ot] = mfma16(gh, Zf[1][s][ot], acc[ot]);
          acc[ot] = mfma16(gl, Zf[0][s][ot], acc[ot]);
          acc[ot] = mfma16(gh, Zf[0][s][ot], acc[ot]);
        }
      }
#pragma unroll
      for (int sc = 0; sc < KCS; ++sc)
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
          if (ot == 1 && !two) break;
          acc[ot] = mfma16(Xf[0][sc], Wc[1][sc][ot], acc[ot]);
          acc[ot] = mfma16(Xf[1][sc], Wc[0][sc][ot], acc[ot]);
          acc[ot] = mfma16(Xf[0][sc], Wc[0][sc][ot], acc[ot]);
        }
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) {
        const int o = 16 * ot + l15;
        if (o < Cout) {
          f32x4 v = acc[ot] + splat4(bias[ot]);
          // the bf16 mode's GELU (common.hip.h: degree-7 fit, |error| <= 8.3e-5, packed fp32 math, no transcendental): with erf + exp + rcp
          // per output this epilogue, not the products or the memory system, set the kernel's time
          if (act == TANTE_ACT_GELU_ERF) v = gelu_poly4<false>(v);
          else if (act != TANTE_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], act);
          }
          if constexpr (OUT16) {
            u32x2 pk;
            pk[0] = pack_bf16x2(v[0], v[1]);
            pk[1] = pack_bf16x2(v[2], v[3]);
            st_wt8((unsigned short*)out + (orow - out) + (long)o * HWl + 16 * wt, pk);
          } else
            st_wt16(orow + (long)o * HWl + 16 * wt, v);      // write-through: 268 MB of output never read back by this launch
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  }
}

// ---- A in the bf16 compute mode: the row DFT with split operands on the bf16 matrix pipe (the products of kernel E3) ------------------------
// Kernel A above feeds v_mfma_f32_16x16x4_f32 (32 cycles per 16 x 16 x 4) from one float per lane and one LDS word per table operand: a
// 512-wide image costs 768 such MFMAs per 32-row wave tile, one wave per SIMD (110 KB of tables), and ran 35-38 us per launch at cfg5
// whatever the grid or the load depth (profiles/r05_ab_fno_dft_unroll.log) -- the wave's LDS-read -> MFMA chain.  Here x and the table are
// split into bf16 hi + lo parts and x . T ~= x_hi T_hi + x_lo T_hi + x_hi T_lo runs as v_mfma_f32_16x16x32_bf16: 18 MFMAs of 16 cycles per
// 32 columns instead of 48 of 32, the table as ready operand fragments (one ds_read_b128 per fragment), and every global access is 32
// contiguous bytes per lane = whole 128-byte runs per image row.  ~1e-5 relative to the fp32 chain; bf16 compute mode only.
// (rows_img, img_gap: x may be n images of rows_img rows each with img_gap extra elements between consecutive images -- a frame of
// every batch item inside a longer rollout buffer; 0 gap = dense)
template <int NT>
__global__ __launch_bounds__(256) void dft_rows_x3_kernel(const float* __restrict__ x, long R, int W, int m2, float* __restrict__ Ar, long rows_img,
                                                          long img_gap) {
  extern __shared__ __attribute__((aligned(16))) u32x4 tfr[];      // [hi | lo][k-step s][column tile nt][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kk = lane >> 4;
  const int KS = W / 32, NF = KS * NT * 64;
  {
    float2* base = (float2*)(tfr + 2 * NF);      // (cos, sin)(2 pi m / W)
    for (int m = tid; m < W; m += 256) {
      float sn, cs;
      sincos_frac(m, W, sn, cs);
      base[m] = make_float2(cs, sn);
    }
    __syncthreads();
    for (int e = tid; e < NF; e += 256) {
      const int ln = e & 63, nt = (e >> 6) % NT, sk = (e >> 6) / NT;
      const int ncol = 16 * nt + (ln & 15), j = ncol < m2 ? ncol : ncol - m2;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int w = 32 * sk + 8 * (ln >> 4) + q;
        v[q] = 0.0f;
        if (ncol < 2 * m2) {
          const float2 t = base[mod_u32((unsigned)(j * w), (unsigned)W)];
          v[q] = ncol < m2 ? t.x : -t.y;
        }
      }
      u32x4 hi, lo;
      split8(v, hi, lo);
      tfr[e] = hi;
      tfr[NF + e] = lo;
    }
    __syncthreads();
  }
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
      xr[mt] = x + r * W + 8 * kk + (img_gap ? (r / rows_img) * img_gap : 0);
    }
    auto kstep = [&](int sk, const f32x4 (&c)[2][2]) {
      u32x4 xh[2], xl[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float v[8] = {c[mt][0][0], c[mt][0][1], c[mt][0][2], c[mt][0][3], c[mt][1][0], c[mt][1][1], c[mt][1][2], c[mt][1][3]};
        split8(v, xh[mt], xl[mt]);
      }
      // (the three products of one accumulator are issued 2 NT MFMAs apart: back to back each would wait for the one before it, and with
      // one or two waves per SIMD nothing else fills the matrix pipe)
      u32x4 th[NT], tl[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) { th[nt] = tfr[(sk * NT + nt) * 64 + lane]; tl[nt] = tfr[NF + (sk * NT + nt) * 64 + lane]; }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = mfma16(xl[mt], th[nt], acc[mt][nt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = mfma16(xh[mt], tl[nt], acc[mt][nt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = mfma16(xh[mt], th[nt], acc[mt][nt]);
    };
    auto fetch = [&](int sk, f32x4 (&c)[2][2]) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) { c[mt][0] = *(const f32x4*)(xr[mt] + 32 * sk); c[mt][1] = *(const f32x4*)(xr[mt] + 32 * sk + 4); }
    };
    if ((KS & 3) == 0) {      // a ring of four k-steps in flight (16 loads of 16 bytes per lane): the rows come cold from HBM
      f32x4 ring[4][2][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) fetch(u, ring[u]);
      for (int s0 = 0; s0 < KS; s0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          f32x4 c[2][2];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) { c[mt][0] = ring[u][mt][0]; c[mt][1] = ring[u][mt][1]; }
          if (s0 + 4 + u < KS) fetch(s0 + 4 + u, ring[u]);
          kstep(s0 + u, c);
        }
      }
    } else {
      f32x4 cur[2][2], nxt[2][2];
      fetch(0, cur);
      for (int sk = 0; sk < KS; ++sk) {
        if (sk + 1 < KS) fetch(sk + 1, nxt);
        kstep(sk, cur);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { cur[mt][0] = nxt[mt][0]; cur[mt][1] = nxt[mt][1]; }
      }
    }
    // D: column l15 = table column 16 nt + l15, rows 4 kk + r
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int k = 16 * nt + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long row = row0 + 16 * mt + 4 * kk + r;
          if (k < 2 * m2 && row < R) Ar[row * (2 * m2) + k] = acc[mt][nt][r];
        }
      }
  }
}

// ---- E3 for the WIDE layers (64 -> 128 and 128 -> 64 channels at W = 128: enc_FNO's second and dec_FNO's first spectral layer at cfg5) -----
// The kernel above keeps a wave's image row in a staging area of its own (Cin x 132 floats) and both output tiles' conv weights in
// registers: Cin <= 64, Cout <= 32.  The wide layers ran on the fp32 kernel E (v_mfma_f32_32x32x2_f32, 64 cycles each): 81 / 41 us per
// call at cfg5 against a matrix floor of ~20 us, every one-row workgroup rebuilding the table and transposing w0.  Here the WORKGROUP owns
// one image row at a time (W = 128 = its 8 column tiles): the four waves stage the row's coefficients zs[2 m2][Cout + 4] and its pixels
// xs[Cin][132] together (16 bytes per lane, whole contiguous runs), then wave w computes output channels 16 OTW w .. 16 OTW (w + 1) - 1
// for every column tile with the same split-operand products (a . b ~= a_hi b_hi + a_lo b_hi + a_hi b_lo on v_mfma_f32_16x16x32_bf16);
// its conv weights (OTW tiles x KCS k-steps, hi and lo) stay in registers across rows.  Every wave splits the row's pixels itself (the
// split is VALU work of the order of its MFMAs; sharing it would cost a second LDS round trip).
// NHWC: the output as channels-last rows ((n h w), Cout) -- what the transposed-conv GEMM behind dec_FNO's first spectral layer reads
// (enc_dec_fno.py:276-323; a 12 us layout copy per call before).  The same fragments with the MFMA operands exchanged give the tile as
// D[o][w]: a lane holds four consecutive CHANNELS of one pixel, 16 bytes of its row.
template <int KCS, int OTW, bool NHWC = false>   // Cin = 32 KCS, Cout = 64 OTW; 2 m2 <= 32; W = 128
__global__ __launch_bounds__(256) void idft_rows_conv_x3w_kernel(const float* __restrict__ Z, const float* __restrict__ x, const float* __restrict__ w0,
                                                                 const float* __restrict__ b0, long n, int H, int m2, int act, float* __restrict__ out) {
  constexpr int Cin = 32 * KCS, Cout = 64 * OTW, W = 128, ZP = Cout + 4, XP = 132;
  extern __shared__ __attribute__((aligned(16))) u32x4 tblw[];      // G fragments [hi | lo][8 column tiles][64 lanes], then zs, then xs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kk = lane >> 4;
  float* zs = (float*)(tblw + 2 * 512);
  float* xs = zs + 32 * ZP;
  {
    float2* base = (float2*)xs;      // (cos, sin)(2 pi m / W), in the pixel area, which is not in use yet
    for (int m = tid; m < W; m += 256) {
      float sn, cs;
      sincos_frac(m, W, sn, cs);
      base[m] = make_float2(cs, sn);
    }
    __syncthreads();
    for (int e = tid; e < 512; e += 256) {
      const int wt = (e >> 6) & 7, ln = e & 63, w = 16 * wt + (ln & 15);
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = 8 * (ln >> 4) + q;
        v[q] = 0.0f;
        if (k < 2 * m2) {
          const int j = k < m2 ? k : k - m2;
          const float2 t = base[mod_u32((unsigned)(j * w), (unsigned)W)];
          const float a = j == 0 ? 1.0f : 2.0f;
          v[q] = k < m2 ? a * t.x : -a * t.y;
        }
      }
      u32x4 hi, lo;
      split8(v, hi, lo);
      tblw[e] = hi;
      tblw[512 + e] = lo;
    }
    __syncthreads();
  }
  const int o_base = 16 * OTW * wave;      // this wave's first output channel
  u32x4 Wc[2][KCS][OTW];
#pragma unroll
  for (int sc = 0; sc < KCS; ++sc)
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot) {
      const int o = o_base + 16 * ot + l15;
      const float* wp = w0 + (long)o * Cin + 32 * sc + 8 * kk;
      const f32x4 a = *(const f32x4*)wp, b = *(const f32x4*)(wp + 4);
      const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      split8(v, Wc[0][sc][ot], Wc[1][sc][ot]);
    }
  float bias[OTW];
  f32x4 bias4[OTW];      // NHWC: the lane's four channels 4 kk .. 4 kk + 3 of tile ot
#pragma unroll
  for (int ot = 0; ot < OTW; ++ot) {
    bias[ot] = b0 ? b0[o_base + 16 * ot + l15] : 0.0f;
    bias4[ot] = b0 ? *(const f32x4*)(b0 + o_base + 16 * ot + 4 * kk) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const long rows = n * H, HWl = (long)H * W;
  const int nz4 = 2 * m2 * Cout / 4;
  for (long rho = blockIdx.x; rho < rows; rho += gridDim.x) {
    const long b = rho / H;
    const int h = (int)(rho - b * H);
    __syncthreads();      // the previous row's operands have been read by every wave
    {
      const float* zrow = Z + rho * 2 * m2 * Cout;
      for (int idx = tid; idx < nz4; idx += 256) {
        const f32x4 zr = *(const f32x4*)(zrow + 4 * idx);
        const int k = (4 * idx) / Cout, o = 4 * idx - k * Cout;
        *(f32x4*)(zs + k * ZP + o) = zr;
      }
      const float* xp = x + ((b * Cin) * H + h) * (long)W;
      constexpr int NX = Cin * 32 / 256;      // 16-byte pieces per thread
      f32x4 xr[NX];
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int piece = tid + 256 * u, c = piece >> 5, c4 = piece & 31;
        xr[u] = *(const f32x4*)(xp + (long)c * HWl + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int piece = tid + 256 * u, c = piece >> 5, c4 = piece & 31;
        *(f32x4*)(xs + c * XP + 4 * c4) = xr[u];
      }
    }
    __syncthreads();
    u32x4 Zf[2][OTW];
#pragma unroll
    for (int ot = 0; ot < OTW; ++ot) {
      const int o = o_base + 16 * ot + l15;
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = 8 * kk + q;
        v[q] = k < 2 * m2 ? zs[k * ZP + o] : 0.0f;
      }
      split8(v, Zf[0][ot], Zf[1][ot]);
    }
    float* orow = out + ((b * Cout + o_base) * H + h) * (long)W + 4 * kk;
#pragma unroll 1
    for (int wt = 0; wt < 8; ++wt) {
      u32x4 Xf[2][KCS];
#pragma unroll
      for (int sc = 0; sc < KCS; ++sc) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = xs[(32 * sc + 8 * kk + q) * XP + 16 * wt + l15];
        split8(v, Xf[0][sc], Xf[1][sc]);
      }
      const u32x4 gh = tblw[wt * 64 + lane], gl = tblw[512 + wt * 64 + lane];
      f32x4 acc[OTW];
#pragma unroll
      for (int ot = 0; ot < OTW; ++ot) {
        acc[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NHWC) {      // the same three products, operands exchanged: D[o][w]
          acc[ot] = mfma16(Zf[1][ot], gh, acc[ot]);
          acc[ot] = mfma16(Zf[0][ot], gl, acc[ot]);
          acc[ot] = mfma16(Zf[0][ot], gh, acc[ot]);
        } else {
          acc[ot] = mfma16(gh, Zf[1][ot], acc[ot]);
          acc[ot] = mfma16(gl, Zf[0][ot], acc[ot]);
          acc[ot] = mfma16(gh, Zf[0][ot], acc[ot]);
        }
      }
#pragma unroll
      for (int sc = 0; sc < KCS; ++sc)
#pragma unroll
        for (int ot = 0; ot < OTW; ++ot) {
          if constexpr (NHWC) {
            acc[ot] = mfma16(Wc[1][sc][ot], Xf[0][sc], acc[ot]);
            acc[ot] = mfma16(Wc[0][sc][ot], Xf[1][sc], acc[ot]);
            acc[ot] = mfma16(Wc[0][sc][ot], Xf[0][sc], acc[ot]);
          } else {
            acc[ot] = mfma16(Xf[0][sc], Wc[1][sc][ot], acc[ot]);
            acc[ot] = mfma16(Xf[1][sc], Wc[0][sc][ot], acc[ot]);
            acc[ot] = mfma16(Xf[0][sc], Wc[0][sc][ot], acc[ot]);
          }
        }
#pragma unroll
      for (int ot = 0; ot < OTW; ++ot) {
        f32x4 v = acc[ot] + (NHWC ? bias4[ot] : splat4(bias[ot]));
        if (act == TANTE_ACT_GELU_ERF) v = gelu_poly4<false>(v);
        else if (act != TANTE_ACT_NONE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], act);
        }
        if constexpr (NHWC) st_wt16(out + ((b * H + h) * (long)W + 16 * wt + l15) * Cout + o_base + 16 * ot + 4 * kk, v);
        else st_wt16(orow + (long)(16 * ot + l15) * HWl + 16 * wt, v);
      }
    }
  }
}

}  // namespace

int tante_spectral_dft_supported(int64_t n, int Cin, int Cout, int H, int W, int m1, int m2) {
  if (n <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if (W % 32 || W < 32 || W > 512 || H % 16) return 0;                  // whole 32-column waves, at most 16 of them; 16-row tiles
  if (m1 < 1 || m2 < 1 || 2 * m1 > H || m2 > W / 2 || m2 > 32) return 0;   // disjoint row bands, no Nyquist column, <= 4 table tiles
  switch (m2) { case 1: case 2: case 3: case 4: case 5: case 6: case 8: case 10: case 12: case 16: case 20: case 24: case 32: break; default: return 0; }   // dft_cols_kernel
  switch (2 * m1) { case 4: case 8: case 10: case 16: case 20: case 32: case 40: case 64: break; default: return 0; }   // idft_cols_kernel instantiations
  const int NT = (2 * m2 + 15) / 16;
  if ((size_t)W * (16 * NT + 4) * 4 + (size_t)W * 8 > 150 * 1024) return 0;   // kernel A's tables
  if ((size_t)2 * m2 * W * 4 + (size_t)Cin * ((Cout + 31) / 32 * 32) * 4 > 150 * 1024) return 0;   // kernel E's tables
  return 1;
}

int64_t tante_spectral_dft_workspace_bytes(int64_t n, int Cin, int Cout, int H, int m1, int m2) {
  const int64_t ar = n * Cin * H * 2 * m2 * 4, xx = n * Cin * 2 * m1 * m2 * 8, yy = n * Cout * 2 * m1 * m2 * 8, zz = n * H * 2 * m2 * Cout * 4;
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  return up(ar) + up(xx) + up(yy) + up(zz);
}

// the split-bf16 inverse row transform's conditions (kernel E3)
static bool dft_x3_ok(int Cin, int Cout, int W, int m2, size_t* lds_out) {
  const size_t ldsE3 = (size_t)2 * ((2 * m2 + 31) / 32) * 512 * 16 + (size_t)4 * (2 * m2 * (Cout + 4) + Cin * 132) * 4;     // table + four waves' staging areas
  if (lds_out) *lds_out = ldsE3;
  return W % 128 == 0 && Cout <= 32 && Cout % 4 == 0 && Cin <= 64 && ldsE3 <= 150 * 1024 && (size_t)W * 8 <= ldsE3 - (size_t)2 * ((2 * m2 + 31) / 32) * 512 * 16;
}
int tante_spectral_dft_bf16out_supported(int64_t n, int Cin, int Cout, int H, int W, int m1, int m2) {
  return tante_spectral_dft_supported(n, Cin, Cout, H, W, m1, m2) && dft_x3_ok(Cin, Cout, W, m2, nullptr);
}

static bool dft_x3w_ok(int Cin, int Cout, int W, int m2) {
  return W == 128 && 2 * m2 <= 32 && ((Cin == 64 && Cout == 128) || (Cin == 128 && Cout == 64));
}
int tante_spectral_dft_x_supported(int64_t n, int Cin, int Cout, int H, int W, int m1, int m2, int strided, int out_bf16, int out_nhwc) {
  if (!tante_spectral_dft_supported(n, Cin, Cout, H, W, m1, m2) || !tante_opt("TANTE_SPECTRAL_X3", 1)) return 0;
  const int NT = (2 * m2 + 15) / 16;
  if ((size_t)2 * (W / 32) * NT * 1024 + (size_t)W * 8 > 150 * 1024) return 0;      // the split-bf16 row transform (the one that takes a stride)
  const bool x3 = dft_x3_ok(Cin, Cout, W, m2, nullptr), x3w = !x3 && dft_x3w_ok(Cin, Cout, W, m2);
  if (out_bf16 && (!x3 || out_nhwc)) return 0;
  if (out_nhwc && !x3w) return 0;
  if (strided && !x3) return 0;      // (the wide layers' last kernel reads dense images)
  return (x3 || x3w) ? 1 : 0;
}

int tante_spectral_dft_forward(const float* x, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1, int wm2, int m1,
                               int m2, const float* w0, const float* b0, int Cout, int act, float* out, void* work, int compute, hipStream_t s,
                               int out_bf16, long x_istride, int out_nhwc) {
  const bool strided = x_istride != 0 && x_istride != (long)Cin * H * W;
  if (x_istride == 0) x_istride = (long)Cin * H * W;
  if ((strided || out_nhwc) && (compute != TANTE_BF16 || !tante_spectral_dft_x_supported(n, Cin, Cout, H, W, m1, m2, strided, out_bf16, out_nhwc)))
    return -2;
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  char* p = (char*)work;
  float* Ar = (float*)p; p += up(n * Cin * H * 2 * m2 * 4);
  float2* X = (float2*)p; p += up(n * Cin * 2 * m1 * m2 * 8);
  float2* Y = (float2*)p; p += up(n * Cout * 2 * m1 * m2 * 8);
  float* Z = (float*)p;
  const long R = (long)n * Cin * H;
  const int NT = (2 * m2 + 15) / 16;
  const size_t ldsA = (size_t)W * (16 * NT + 4) * 4 + (size_t)W * 8;
  const long tiles = (R + 31) / 32;
  const unsigned gridA = (unsigned)std::min<long>(256, (tiles + 3) / 4);
  static TantePerDevice attrA[4], attrE[9];
#define TANTE_DFT_A(NTV)                                                                                                          \
  case NTV:                                                                                                                       \
    attrA[NTV - 1].once([&] { (void)hipFuncSetAttribute((const void*)dft_rows_kernel<NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }); \
    hipLaunchKernelGGL(dft_rows_kernel<NTV>, dim3(gridA), dim3(256), ldsA, s, x, R, W, m2, Ar);                                  \
    break;
  const size_t ldsA3 = (size_t)2 * (W / 32) * NT * 1024 + (size_t)W * 8;
  if (compute == TANTE_BF16 && tante_opt("TANTE_SPECTRAL_X3", 1) && ldsA3 <= 150 * 1024 && ((uintptr_t)x % 16) == 0) {      // split-bf16 row transform
    static TantePerDevice attrA3[4];
#define TANTE_DFT_A3(NTV)                                                                                                         \
  case NTV:                                                                                                                       \
    attrA3[NTV - 1].once([&] { (void)hipFuncSetAttribute((const void*)dft_rows_x3_kernel<NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }); \
    hipLaunchKernelGGL(dft_rows_x3_kernel<NTV>, dim3(gridA), dim3(256), ldsA3, s, x, R, W, m2, Ar, (long)Cin * H,                  \
                       strided ? x_istride - (long)Cin * H * W : 0L);                                                              \
    break;
    switch (NT) { TANTE_DFT_A3(1) TANTE_DFT_A3(2) TANTE_DFT_A3(3) TANTE_DFT_A3(4) default: return -2; }
#undef TANTE_DFT_A3
  } else {
    if (strided) return -2;      // (the fp32 row transform reads dense images)
    switch (NT) { TANTE_DFT_A(1) TANTE_DFT_A(2) TANTE_DFT_A(3) TANTE_DFT_A(4) default: return -2; }
  }
#undef TANTE_DFT_A
  const long NC = (long)n * Cin;
  const unsigned gridB = (unsigned)((NC * 2 * m1 + 3) / 4);
  switch (m2) {
#define TANTE_DFT_B(V) case V: hipLaunchKernelGGL(dft_cols_kernel<V>, dim3(gridB), dim3(256), 0, s, Ar, NC, H, m1, X); break;
    TANTE_DFT_B(1) TANTE_DFT_B(2) TANTE_DFT_B(3) TANTE_DFT_B(4) TANTE_DFT_B(5) TANTE_DFT_B(6) TANTE_DFT_B(8) TANTE_DFT_B(10) TANTE_DFT_B(12) TANTE_DFT_B(16)
    TANTE_DFT_B(20) TANTE_DFT_B(24) TANTE_DFT_B(32)
#undef TANTE_DFT_B
    default: return -2;
  }
  const long totalC = (long)n * Cout * 2 * m1 * m2;
  int cs = 1;
  while (cs < 16 && 2 * cs <= Cin && totalC * cs < 65536) cs *= 2;
  hipLaunchKernelGGL(spectral_mix_kernel, dim3((unsigned)std::min<long>(4096, (totalC * cs + 255) / 256)), dim3(256), 0, s, X, w_re, w_im, (long)n, Cin, Cout,
                     m1, m2, wm1, wm2, 1.0f / ((float)H * (float)W), Y, cs);
  const dim3 gridD((unsigned)((H + DFT_D_HB - 1) / DFT_D_HB), (unsigned)n);
  switch (2 * m1) {
#define TANTE_DFT_D(V) case V: hipLaunchKernelGGL(idft_cols_kernel<V>, gridD, dim3(256), 0, s, Y, H, Cout, m2, Z); break;
    TANTE_DFT_D(4) TANTE_DFT_D(8) TANTE_DFT_D(10) TANTE_DFT_D(16) TANTE_DFT_D(20) TANTE_DFT_D(32) TANTE_DFT_D(40) TANTE_DFT_D(64)
#undef TANTE_DFT_D
    default: return -2;
  }
  // bf16 compute mode: split-operand products on the bf16 matrix pipe (idft_rows_conv_x3_kernel); TANTE_SPECTRAL_X3 = 0 keeps the fp32 kernel
  size_t ldsE3 = 0;
  const bool x3 = dft_x3_ok(Cin, Cout, W, m2, &ldsE3) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0;
  if (out_bf16 && !x3) return -2;      // (the caller asked tante_spectral_dft_bf16out_supported first)
  // the wide layers (idft_rows_conv_x3w_kernel): 64 -> 128 and 128 -> 64 channels at W = 128
  if (!x3 && !out_bf16 && compute == TANTE_BF16 && tante_opt("TANTE_SPECTRAL_X3", 1) && dft_x3w_ok(Cin, Cout, W, m2) && ((uintptr_t)x % 16) == 0 &&
      ((uintptr_t)out % 16) == 0 && ((uintptr_t)w0 % 16) == 0 && (!b0 || ((uintptr_t)b0 % 16) == 0 || !out_nhwc)) {
    const size_t ldsW = (size_t)2 * 512 * 16 + (size_t)(32 * (Cout + 4) + Cin * 132) * 4;
    const long rowsW = (long)n * H;
    static TantePerDevice attrW[4];
#define TANTE_DFT_W(IDX, A_, B_, NH, GR)                                                                                                   \
  {                                                                                                                                        \
    attrW[IDX].once([&] { (void)hipFuncSetAttribute((const void*)idft_rows_conv_x3w_kernel<A_, B_, NH>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }); \
    hipLaunchKernelGGL((idft_rows_conv_x3w_kernel<A_, B_, NH>), dim3((unsigned)std::min<long>(rowsW, GR)), dim3(256), ldsW, s, Z, x, w0, b0, (long)n, H, m2, act, out); \
  }
    if (Cin == 64) {
      if (out_nhwc) TANTE_DFT_W(2, 2, 2, true, 512) else TANTE_DFT_W(0, 2, 2, false, 512)
    } else {
      if (out_nhwc) TANTE_DFT_W(3, 4, 1, true, 256) else TANTE_DFT_W(1, 4, 1, false, 256)
    }
#undef TANTE_DFT_W
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  if (out_nhwc) return -2;
  if (out_bf16 || (compute == TANTE_BF16 && x3 && tante_opt("TANTE_SPECTRAL_X3", 1))) {
    const int kzs = (2 * m2 + 31) / 32, kcs = (Cin + 31) / 32, ncb = W / 128;
    const long rows4 = ((long)n * H + 3) / 4;
    const size_t lds = ldsE3;
    const long per_cu = std::max<long>(1, std::min<long>(4, (150 * 1024) / (long)lds));
    const long nrg = std::max<long>(1, std::min<long>(rows4, 256 * per_cu / ncb));
    static TantePerDevice attrE3[8];
#define TANTE_DFT_E3(A_, B_)                                                                                                                   \
  {                                                                                                                                            \
    if (out_bf16) {                                                                                                                            \
      attrE3[4 + (A_ - 1) * 2 + (B_ - 1)].once([&] { (void)hipFuncSetAttribute((const void*)idft_rows_conv_x3_kernel<A_, B_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }); \
      hipLaunchKernelGGL((idft_rows_conv_x3_kernel<A_, B_, true>), dim3((unsigned)(nrg * ncb)), dim3(256), lds, s, Z, x, w0, b0, (long)n, Cin, Cout, H, W, m2, act, out, x_istride); \
    } else {                                                                                                                                   \
      attrE3[(A_ - 1) * 2 + (B_ - 1)].once([&] { (void)hipFuncSetAttribute((const void*)idft_rows_conv_x3_kernel<A_, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }); \
      hipLaunchKernelGGL((idft_rows_conv_x3_kernel<A_, B_>), dim3((unsigned)(nrg * ncb)), dim3(256), lds, s, Z, x, w0, b0, (long)n, Cin, Cout, H, W, m2, act, out, x_istride); \
    }                                                                                                                                          \
  }
    if (kzs == 1 && kcs == 1) TANTE_DFT_E3(1, 1)
    else if (kzs == 1) TANTE_DFT_E3(1, 2)
    else if (kcs == 1) TANTE_DFT_E3(2, 1)
    else TANTE_DFT_E3(2, 2)
#undef TANTE_DFT_E3
    return hipGetLastError() == hipSuccess ? 0 : -3;
  }
  if (strided) return -2;                      // (the fp32 kernels read dense images)
  const int wpb = W / 32;                      // waves per workgroup: one per 32 output columns
  const size_t ldsE = (size_t)2 * m2 * W * 4 + (size_t)Cin * ((Cout + 31) / 32 * 32) * 4;
  const long rows = (long)n * H;
  // each workgroup builds the tables once (2 m2 W sincos), so it should walk many rows
  const long wgs = std::min<long>(rows, 256L * std::max(1, 16 / wpb));
#define TANTE_DFT_E(V, IDX)                                                                                                        \
  {                                                                                                                                \
    attrE[IDX].once([&] { (void)hipFuncSetAttribute((const void*)idft_rows_conv_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); }); \
    hipLaunchKernelGGL(idft_rows_conv_kernel<V>, dim3((unsigned)wgs), dim3(64 * wpb), ldsE, s, Z, x, w0, b0, (long)n, Cin, Cout, H, W, m2, act, out); \
  }
  if (m2 == 5) TANTE_DFT_E(5, 1)      // (M2 = 20 fully unrolled needs > 128 registers: the unroll-by-4 loop of the generic form serves it)
  else if (m2 == 8) TANTE_DFT_E(8, 2)
  else TANTE_DFT_E(0, 3)
#undef TANTE_DFT_E
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
