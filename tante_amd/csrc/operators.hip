// Operator kernels around the GEMM / attention core that the general encoder-decoder stages (padded or overlapping patch
// convolutions, bilinear-resized transposed convolutions), the spectral operator path (enc_dec_fno.py) and CViT (cvit.py) need.
// All of them are HBM-bound gather / pointwise / small-contraction kernels: coalesced on the innermost axis, fp32 arithmetic.
#include "common.hip.h"
#include "spectral_dft.h"
#include "fused_common.hip.h"
#include <hipfft/hipfft.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>

namespace {

__device__ __forceinline__ float ldx(const void* p, int dtype, long i) {
  return dtype == TANTE_BF16 ? __uint_as_float(((unsigned)((const unsigned short*)p)[i]) << 16) : ((const float*)p)[i];
}
__device__ __forceinline__ void stx(void* p, int dtype, long i, float v) {
  if (dtype == TANTE_BF16) ((__bf16*)p)[i] = (__bf16)v;
  else ((float*)p)[i] = v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// ---- im2col: rows = output positions (img, oh, ow), columns = (c, kh, kw) [korder 0] or (kh, kw, c) [korder 1] ---------------
__global__ void im2col_kernel(const void* __restrict__ x, int x_dtype, int nchw, long n_img, int C, int H, int W, int kh, int kw,
                              int sh, int sw, int ph, int pw, int Ho, int Wo, int korder, void* __restrict__ cols, int cols_dtype) {
  const int K = C * kh * kw;
  const long total = n_img * Ho * Wo * K;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / K;
    const int k = (int)(idx - m * K);
    int c, i, j;
    if (korder == 0) { c = k / (kh * kw); const int r = k - c * kh * kw; i = r / kw; j = r - i * kw; }
    else { const int r = k / C; c = k - r * C; i = r / kw; j = r - i * kw; }
    const long img = m / ((long)Ho * Wo);
    const int o = (int)(m - img * Ho * Wo), oh = o / Wo, ow = o - oh * Wo;
    const int y = oh * sh - ph + i, xx = ow * sw - pw + j;
    float v = 0.0f;
    if (y >= 0 && y < H && xx >= 0 && xx < W) {
      const long src = nchw ? ((img * C + c) * H + y) * W + xx : ((img * H + y) * W + xx) * C + c;
      v = ldx(x, x_dtype, src);
    }
    stx(cols, cols_dtype, idx, v);
  }
}

// Channels-FIRST fp32 source, (c, kh, kw) columns, stride = kernel (non-overlapping P x P patches, any padding < P): the padded 4 x 4 and
// 2 x 2 stages of the spectral encoder (enc_dec_fno.py:224-273 through RealConv2d).  The scalar kernel above walks the OUTPUT in order:
// its reads hop 4 bytes at a time between planes 1 MB apart (2.1 TB/s over 400 MB: 189 us for the 512 x 512 x 32-channel stage, the
// largest kernel of cfg5 once the spectral layers were fixed).  Here a workgroup owns TW patches of one output row: it reads the
// C x P input segments of TW P (+ padding shift) contiguous floats -- coalesced, zero outside the image -- into LDS and writes TW
// whole patch rows of K = C P P contiguous elements.  Row stride TW P + 1 floats: the (c, kh) segments a wave reads back at one kw sit
// on different banks.
template <int P, bool BF16OUT, bool XB16 = false>      // XB16: the source image is bf16 (a producer that rounded for this gather already)
__global__ __launch_bounds__(256) void im2col_nchw_tile_kernel(const void* __restrict__ xv, int C, int H, int W, int pad, int Ho, int Wo, int CC,
                                                               void* __restrict__ cols) {
  const float* const x = (const float*)xv;
  const unsigned short* const xh = (const unsigned short*)xv;
  constexpr int SEG = 64, TW = SEG / P, S = SEG + 1;   // a tile = TW patches = 64 input columns; LDS rows padded to 65 floats
  extern __shared__ float tile[];                        // [CC * P][S]
  const int tiles_w = (Wo + TW - 1) / TW;
  const int tw = blockIdx.x % tiles_w, oh = (blockIdx.x / tiles_w) % Ho;
  const long img = blockIdx.x / ((long)tiles_w * Ho);
  const int wo0 = tw * TW, x0 = wo0 * P - pad, y0 = oh * P - pad;
  const int K = C * P * P, nt = min(TW, Wo - wo0);
  const long row0 = (img * Ho + oh) * (long)Wo + wo0;
  const int j = threadIdx.x & 63, r0 = threadIdx.x >> 6;          // a wave reads one 256-byte segment per instruction
  const int xx = x0 + j;
  const bool xin = xx >= 0 && xx < W;
  for (int c0 = 0; c0 < C; c0 += CC) {
    const int cc = min(CC, C - c0), rows = cc * P;                 // (c, kh) segments of this pass
    constexpr int NLD = 8;                                         // independent loads in flight per thread (16: +-0, 32: -9 % on cfg5)
    for (int rb = r0; rb < rows; rb += 4 * NLD) {
      float v[NLD];
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const int r = rb + 4 * u, c = r / P, kh = r - c * P, y = y0 + kh;
        const bool ok = r < rows && xin && y >= 0 && y < H;
        const long src = ((img * C + c0 + c) * (long)H + y) * W + xx;
        if constexpr (XB16) v[u] = ok ? __uint_as_float((unsigned)xh[src] << 16) : 0.0f;
        else v[u] = ok ? x[src] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < NLD; ++u)
        if (rb + 4 * u < rows) tile[(rb + 4 * u) * S + j] = v[u];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nt * rows; e += 256) {           // one (patch, c, kh) = P consecutive columns per thread
      const int t = e / rows, ckh = e - t * rows;
      const float* src = tile + ckh * S + t * P;
      const long dst = (row0 + t) * K + (long)c0 * P * P + (long)ckh * P;
      if constexpr (BF16OUT) {
        if constexpr (P == 4) {
          u32x2 u;
          u[0] = pack_bf16x2(src[0], src[1]); u[1] = pack_bf16x2(src[2], src[3]);
          *(u32x2*)((unsigned short*)cols + dst) = u;
        } else {
          *(unsigned*)((unsigned short*)cols + dst) = pack_bf16x2(src[0], src[1]);
        }
      } else {
#pragma unroll
        for (int q = 0; q < P; ++q) ((float*)cols)[dst + q] = src[q];
      }
    }
    __syncthreads();
  }
}

// channels-last source, (kh, kw, c) columns, one dtype on both sides and whole 16-byte pieces per channel run: a gather of 16-byte
// pieces (the scalar kernel above moves 2 bytes per thread behind six integer divisions: 67 us for a 2 x 25 MB permutation)
template <int EB>   // element bytes
__global__ void im2col_vec_kernel(const char* __restrict__ x, long n_img, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                                  int Ho, int Wo, char* __restrict__ cols) {
  constexpr int EPV = 16 / EB;
  const int cpv = C / EPV, kv = kh * kw * cpv;        // 16-byte pieces per pixel / per patch row
  const long total = n_img * Ho * Wo * kv;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / kv;
    const int k = (int)(idx - m * kv);
    const int r = k / cpv, c = k - r * cpv, i = r / kw, j = r - i * kw;
    const long img = m / ((long)Ho * Wo);
    const int o = (int)(m - img * Ho * Wo), oh = o / Wo, ow = o - oh * Wo;
    const int y = oh * sh - ph + i, xx = ow * sw - pw + j;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (y >= 0 && y < H && xx >= 0 && xx < W) v = *(const u32x4*)(x + ((((img * H + y) * W + xx) * cpv + c) << 4));
    *(u32x4*)(cols + (idx << 4)) = v;
  }
}

// ---- adaptive average pooling (channels-last) + activation: cell o averages rows floor(o*H/Ht) .. ceil((o+1)*H/Ht) - 1 ---------
__global__ void avgpool_kernel(const void* __restrict__ x, int x_dtype, long n_img, int H, int W, int C, int Ht, int Wt, int act,
                               void* __restrict__ y, int y_dtype) {
  const long total = n_img * Ht * Wt * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long r = idx / C;
    const int ow = (int)(r % Wt); r /= Wt;
    const int oh = (int)(r % Ht);
    const long img = r / Ht;
    const int y0 = (oh * H) / Ht, y1 = ((oh + 1) * H + Ht - 1) / Ht, x0 = (ow * W) / Wt, x1 = ((ow + 1) * W + Wt - 1) / Wt;
    float s = 0.0f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) s += ldx(x, x_dtype, ((img * H + yy) * W + xx) * C + c);
    stx(y, y_dtype, idx, apply_act(s / (float)((y1 - y0) * (x1 - x0)), act));
  }
}

// backward of the pooling above (no activation), gather form: input pixel (yy, xx) lies in the cells oh = floor(yy Ht / H) .. ceil((yy + 1) Ht / H) - 1
// (one cell when H divides evenly, two where the adaptive windows overlap) and takes dy / area from each.
__global__ void avgpool_bwd_kernel(const void* __restrict__ dy, int dy_dtype, long n_img, int H, int W, int C, int Ht, int Wt,
                                   void* __restrict__ dx, int dx_dtype) {
  const long total = n_img * H * W * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long r = idx / C;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H);
    const long img = r / H;
    const int oh0 = (int)(((long)yy * Ht) / H), oh1 = (int)((((long)yy + 1) * Ht + H - 1) / H);
    const int ow0 = (int)(((long)xx * Wt) / W), ow1 = (int)((((long)xx + 1) * Wt + W - 1) / W);
    float s = 0.0f;
    for (int oh = oh0; oh < oh1; ++oh) {
      const int y0 = (oh * H) / Ht, y1 = ((oh + 1) * H + Ht - 1) / Ht;
      for (int ow = ow0; ow < ow1; ++ow) {
        const int x0 = (ow * W) / Wt, x1 = ((ow + 1) * W + Wt - 1) / Wt;
        s += ldx(dy, dy_dtype, ((img * Ht + oh) * Wt + ow) * C + c) / (float)((y1 - y0) * (x1 - x0));
      }
    }
    stx(dx, dx_dtype, idx, s);
  }
}

// ---- col2im for a transposed convolution with overlapping taps (stride < kernel): gather form, no atomics ----------------------
// cols[(img, ih, iw)][(kh, kw, co)] = sum_ci x[img, ih, iw, ci] W[ci, co, kh, kw] (one GEMM); output pixel (y, x) sums the taps with
// ih * s - p + kh = y, iw * s - p + kw = x, plus the bias; out is channels-last (img, Hf, Wf, Cout), Hf = (Hi - 1) s - 2 p + P.
__global__ void col2im_kernel(const void* __restrict__ cols, int cols_dtype, long n_img, int Hi, int Wi, int P, int st, int pd, int Cout,
                              const float* __restrict__ bias, int Hf, int Wf, void* __restrict__ out, int out_dtype) {
  const long total = n_img * Hf * Wf * Cout;
  const long ldc = (long)P * P * Cout;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int co = (int)(idx % Cout);
    long r = idx / Cout;
    const int x = (int)(r % Wf); r /= Wf;
    const int y = (int)(r % Hf);
    const long img = r / Hf;
    float acc = bias ? bias[co] : 0.0f;
    for (int kh = 0; kh < P; ++kh) {
      const int ty = y + pd - kh;
      if (ty < 0 || ty % st) continue;
      const int ih = ty / st;
      if (ih >= Hi) continue;
      for (int kw = 0; kw < P; ++kw) {
        const int tx = x + pd - kw;
        if (tx < 0 || tx % st) continue;
        const int iw = tx / st;
        if (iw >= Wi) continue;
        acc += ldx(cols, cols_dtype, ((img * Hi + ih) * Wi + iw) * ldc + ((long)kh * P + kw) * Cout + co);
      }
    }
    stx(out, out_dtype, idx, acc);
  }
}

// ---- bilinear resize (align_corners = False, F.interpolate semantics) of a cropped window + activation ------------------------
// in element (img, c, y, x) at img*isn + c*isc + (y + cy)*ish + (x + cx)*isw, window Hi x Wi;  out likewise with its own strides.
__global__ void resize_kernel(const void* __restrict__ in, int in_dtype, long n_img, int C, int Hi, int Wi, int cy, int cx, long isn,
                              long isc, long ish, long isw, int Ho, int Wo, long osn, long osc, long osh, long osw, int act,
                              void* __restrict__ out, int out_dtype, int c_fast) {
  const long total = n_img * C * Ho * Wo;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c, oy, ox;
    long img;
    if (c_fast) { c = (int)(idx % C); long r = idx / C; ox = (int)(r % Wo); r /= Wo; oy = (int)(r % Ho); img = r / Ho; }
    else { ox = (int)(idx % Wo); long r = idx / Wo; oy = (int)(r % Ho); r /= Ho; c = (int)(r % C); img = r / C; }
    float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
    fy = fy < 0.0f ? 0.0f : fy;
    fx = fx < 0.0f ? 0.0f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const long base = img * isn + c * isc;
    const float v00 = ldx(in, in_dtype, base + (y0 + cy) * ish + (x0 + cx) * isw), v01 = ldx(in, in_dtype, base + (y0 + cy) * ish + (x1 + cx) * isw);
    const float v10 = ldx(in, in_dtype, base + (y1 + cy) * ish + (x0 + cx) * isw), v11 = ldx(in, in_dtype, base + (y1 + cy) * ish + (x1 + cx) * isw);
    const float v = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
    stx(out, out_dtype, img * osn + c * osc + oy * osh + ox * osw, apply_act(v, act));
  }
}

// ---- LayerNorm with affine: one wave per row ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_affine_kernel(const void* __restrict__ x, int x_dtype, long M, int C, float eps,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        void* __restrict__ y, int y_dtype) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += ldx(x, x_dtype, row * C + c);
  const float mean = wave_sum(s) / C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = ldx(x, x_dtype, row * C + c) - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
  for (int c = lane; c < C; c += 64)
    stx(y, y_dtype, row * C + c, (ldx(x, x_dtype, row * C + c) - mean) * rstd * (gamma ? gamma[c] : 1.0f) + (beta ? beta[c] : 0.0f));
}

// fp32 rows of C = 256 NV channels: the row stays in registers (one 16-byte load per lane and 4-channel group)
template <int NV>
__global__ __launch_bounds__(256) void ln_affine_vec_kernel(const float* __restrict__ x, long M, float eps, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, void* __restrict__ y, int y_dtype) {
  constexpr int C = 256 * NV;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = *(const f32x4*)(x + row * C + (i * 64 + lane) * 4);
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float mean = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    f32x4 h = (v[i] - mean) * rstd;
    if (gamma) h = h * *(const f32x4*)(gamma + c);
    if (beta) h = h + *(const f32x4*)(beta + c);
    if (y_dtype == TANTE_BF16) {
      u32x2 u;
      u[0] = pack_bf16x2(h[0], h[1]);
      u[1] = pack_bf16x2(h[2], h[3]);
      *(u32x2*)((unsigned short*)y + row * C + c) = u;
    } else {
      *(f32x4*)((float*)y + row * C + c) = h;
    }
  }
}

// ---- spectral layer: low-mode complex contraction (writes the WHOLE spectrum: zeros outside the two bands) --------------------
// Y[b, o, i, j] = scale * sum_c X[b, c, i, j] * Wt[c, o, wi, j]; top band i < m1 (wi = i), bottom band i >= H - m1
// (wi = i - (H - m1)); the bottom band wins where they overlap (it is written second at enc_dec_fno.py:207-210).
__global__ void spectral_modes_kernel(const float2* __restrict__ X, const float* __restrict__ w_re, const float* __restrict__ w_im,
                                      long n, int Cin, int Cout, int H, int Wf, int m1, int m2, int wm1, int wm2, float scale,
                                      float2* __restrict__ Y) {
  const long total = n * Cout * H * Wf;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % Wf);
    long r = idx / Wf;
    const int i = (int)(r % H); r /= H;
    const int o = (int)(r % Cout);
    const long b = r / Cout;
    float2 acc = make_float2(0.f, 0.f);
    int wi = -1;
    if (j < m2) {
      if (i >= H - m1) wi = i - (H - m1);
      else if (i < m1) wi = i;
    }
    if (wi >= 0) {
      for (int c = 0; c < Cin; ++c) {
        const float2 xv = X[((b * Cin + c) * H + i) * Wf + j];
        const long wo = (((long)c * Cout + o) * wm1 + wi) * wm2 + j;
        const float wr = w_re[wo], wim = w_im[wo];
        acc.x += xv.x * wr - xv.y * wim;
        acc.y += xv.x * wim + xv.y * wr;
      }
      acc.x *= scale; acc.y *= scale;
    }
    Y[idx] = acc;
  }
}

// out[b, o, p] = act(y[b, o, p] + b0[o] + sum_c w0[o, c] x[b, c, p])   (channels-first 1x1 conv + the spectral branch)
// block: 64 pixels x all outputs; x tile (Cin x 64) and w0 staged in LDS; thread (p = tid & 63, og = tid >> 6) owns outputs og, og+4, ...
__global__ __launch_bounds__(256) void conv1x1_add_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                          const float* __restrict__ b0, const float* y, long HW, int Cin,
                                                          int Cout, int act, float* out) {   // y may alias out
  extern __shared__ float sm[];   // [Cin][64] x tile, then w0 [Cout][Cin]
  float* xs = sm;
  float* ws = sm + Cin * 64;
  const long b = blockIdx.y;
  const long p0 = (long)blockIdx.x * 64;
  const int tid = threadIdx.x, p = tid & 63, og = tid >> 6;
  for (int i = tid; i < Cin * 64; i += 256) {
    const int c = i >> 6, pp = i & 63;
    xs[i] = (p0 + pp < HW) ? x[(b * Cin + c) * HW + p0 + pp] : 0.0f;
  }
  for (int i = tid; i < Cout * Cin; i += 256) ws[i] = w0[i];
  __syncthreads();
  if (p0 + p >= HW) return;
  for (int o = og; o < Cout; o += 4) {
    float acc = b0 ? b0[o] : 0.0f;
    const float* wr = ws + o * Cin;
    for (int c = 0; c < Cin; ++c) acc += wr[c] * xs[c * 64 + p];
    const long oi = (b * Cout + o) * HW + p0 + p;
    out[oi] = apply_act(acc + (y ? y[oi] : 0.0f), act);
  }
}

// ---- CViT: softmax attention with separate query and key/value sequences (fp32, exact; online softmax over key chunks) ---------
// q row (b, i) at q + (b*Lq + i)*ldq + h*D, k / v rows (b, j) at k|v + (b*Lk + j)*ldkv + h*D, o row at o + (b*Lq + i)*ldo + h*D.
// block = 256 queries of one (batch, head); keys/values staged in LDS 64 at a time (broadcast reads in the inner loops).
template <int D>
__global__ __launch_bounds__(256) void cross_attn_kernel(const void* __restrict__ q, const void* __restrict__ k, const void* __restrict__ v,
                                                         void* __restrict__ o, int dtype, int n_head, int Lq, int Lk, long ldq, long ldkv,
                                                         long ldo, float scale, long qbr) {
  __shared__ float ks[64][D];
  __shared__ float vs[64][D];
  const int bh = blockIdx.x, b = bh / n_head, h = bh - b * n_head;
  const int i = blockIdx.y * 256 + threadIdx.x;
  const bool live = i < Lq;
  float qr[D], acc[D];
  const long qoff = ((long)b * qbr + (live ? i : 0)) * ldq + (long)h * D;
#pragma unroll
  for (int d = 0; d < D; ++d) { qr[d] = ldx(q, dtype, qoff + d) * scale; acc[d] = 0.0f; }
  float m = -INFINITY, l = 0.0f;
  for (int j0 = 0; j0 < Lk; j0 += 64) {
    const int nj = Lk - j0 < 64 ? Lk - j0 : 64;
    __syncthreads();
    for (int e = threadIdx.x; e < nj * D; e += 256) {
      const int j = e / D, d = e - j * D;
      const long off = ((long)b * Lk + j0 + j) * ldkv + (long)h * D + d;
      ks[j][d] = ldx(k, dtype, off);
      vs[j][d] = ldx(v, dtype, off);
    }
    __syncthreads();
    for (int j = 0; j < nj; ++j) {
      float s = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) s += qr[d] * ks[j][d];
      const float mn = fmaxf(m, s);
      const float corr = __expf(m - mn), pj = __expf(s - mn);   // first key: exp(-inf) = 0 rescales the empty sums
      l = l * corr + pj;
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] = acc[d] * corr + pj * vs[j][d];
      m = mn;
    }
  }
  if (live) {
    const float inv = 1.0f / l;
    const long ooff = ((long)b * Lq + i) * ldo + (long)h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) stx(o, dtype, ooff + d, acc[d] * inv);
  }
}

// ---- CViT attention on the matrix cores (bf16, head dim 64, up to 512 keys) ----------------------------------------------------
// One workgroup = one (batch, head) and XQ_PER_WG queries.  K (rows = keys) and V^T (rows = head dims, keys k-permuted inside every
// 32-block) are staged ONCE per workgroup in LDS as bf16 in the swizzled 16-byte-chunk image the MFMA A operand reads with
// ds_read_b128; every wave then walks its queries 32 at a time (two B operands per fragment read, halving the LDS traffic):
//   S^T tile (16 keys x 16 queries) = K_tile . Q^T            -- "rows = keys", so a lane holds 4 keys of ONE query column
//   online softmax over 128-key chunks: per-lane max / partial sums, two __shfl_xor for the cross-lane max, exp2 with the
//   1/sqrt(D) log2(e) factor applied to the fp32 scores
//   O^T (16 dims x 16 queries) += V^T_tile . P^T              -- the probability tiles, packed to bf16, ARE the B operand
// (key order inside a 32-block = accumulator order, matched by the V^T staging permutation).  Nothing but q, k, v, o touches HBM.
__device__ __forceinline__ u32x4 xpack8(const f32x4& a, const f32x4& b) {
  u32x4 f;
  f[0] = pack_bf16x2(a[0], a[1]); f[1] = pack_bf16x2(a[2], a[3]); f[2] = pack_bf16x2(b[0], b[1]); f[3] = pack_bf16x2(b[2], b[3]);
  return f;
}
__device__ __forceinline__ f32x4 xmfma(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
constexpr int XD = 64;            // head dim
constexpr int XQG = 2;            // 16-query groups a wave processes together (2: half the LDS reads per MFMA, but 50 more VGPRs)
constexpr int XWAVES = 4;         // waves per workgroup (measured: XQG 1 with 6 waves = 3 waves/SIMD is 25 % slower than XQG 2 with 4)
constexpr int XGPW_MAX = 8;       // iterations per wave (32 queries each): 8 where the queries fill the chip anyway (K / V staged once per 1024 queries);
                                  // fewer -- more, shorter workgroups -- for the encoder's few hundred queries per (sample, head)

template <int OFF>
__device__ __forceinline__ u32x2 x_tr_read_off(unsigned base) {   // base VGPR + 16-bit immediate offset
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(base), "n"(OFF) : "memory");
  return r;
}
__device__ __forceinline__ u32x2 x_tr_read(unsigned addr) {   // ds_read_b64_tr_b16: 4 rows x 16 columns per 16-lane group, column-major out
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
// V rows are 128 bytes (64 bf16): 16-byte chunk c of row r sits at chunk c ^ (((r >> 1) & 3) << 1), which makes the two 16-lane groups of
// a half-wave (8 consecutive rows, one aligned chunk pair each) hit 16 distinct 16-byte slots of the 256-byte bank row
__device__ __forceinline__ int xv_swz(int row) { return ((row >> 1) & 3) << 1; }

__global__ __launch_bounds__(XWAVES * 64, 2) void xattn_mfma_kernel(const unsigned short* __restrict__ q, const unsigned short* __restrict__ k,
                                                            const unsigned short* __restrict__ v, unsigned short* __restrict__ o, int n_head,
                                                            int Lq, int Lk, int Sp, long ldq, long ldkv, long ldo, float scale_log2e, long qbr, int XGPW) {
  const int XQ_PER_WG = XWAVES * XGPW * 16 * XQG;
  extern __shared__ __attribute__((aligned(16))) char xsm[];   // K image [Sp][8 chunks] (row reads), then V image [Sp][8 chunks] (transposed reads)
  char* ks = xsm;
  char* vs = xsm + (size_t)Sp * XD * 2;
  const int bh = blockIdx.x, b = bh / n_head, h = bh - b * n_head;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  // ---- stage K and V row-major by LDS-DMA (no register round trip, no transposing stores): a wave instruction moves 8 rows of 128 bytes;
  // the swizzles are applied to the SOURCE chunk each lane fetches.  Rows >= Lk re-read row Lk - 1 (their scores are masked below). ----
  {
    const int rl = lane >> 3, cs = lane & 7;
    for (int r0 = wave * 8; r0 < Sp; r0 += XWAVES * 8) {
      const int row = r0 + rl, src = row < Lk ? row : Lk - 1;
      const unsigned short* gk = k + ((long)b * Lk + src) * ldkv + (long)h * XD + ((cs ^ ((row >> 1) & 7)) * 8);
      const unsigned short* gv = v + ((long)b * Lk + src) * ldkv + (long)h * XD + ((cs ^ xv_swz(row)) * 8);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gk, (__attribute__((address_space(3))) void*)(ks + r0 * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gv, (__attribute__((address_space(3))) void*)(vs + r0 * 128), 16, 0, 0);
    }
  }
  // per-lane LDS byte addresses; everything that varies with the (compile-time) tile indices is an instruction immediate
  unsigned kb[2];   // K row l15 of a 16-key tile, 32-dim block bb (row reads, ds_read_b128)
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) kb[bb] = lds_addr(ks) + l15 * 128 + (swz_chunk(l15, bb * 4 + kk, 8) << 4);
  // transposed V reads: lane 4 qq + pp of a 16-lane group supplies row (4 kk + qq), columns 4 pp .. 4 pp + 3 of the 16-dim tile dt; the
  // swizzle only looks at row bits 1-2, which the 32-key block offset and the +16 of the second read leave alone
  const int qq = l15 >> 2, pp = l15 & 3;
  unsigned vb[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
    vb[dt] = lds_addr(vs) + (4 * kk + qq) * 128 + (((dt * 2 + (pp >> 1)) ^ xv_swz(4 * kk + qq)) << 4) + 8 * (pp & 1);
  const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // the next iteration's query fragments are fetched while this one computes
  auto load_q = [&](int it, u32x4 (&dst)[XQG][2]) {
    const int q0n = blockIdx.y * XQ_PER_WG + (wave * XGPW + it) * 16 * XQG;
#pragma unroll
    for (int g = 0; g < XQG; ++g) {
      const int qi = q0n + g * 16 + l15;
      const bool lv = it < XGPW && qi < Lq;
      const long qoff = ((long)b * qbr + (lv ? qi : 0)) * ldq + (long)h * XD;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) dst[g][bb] = lv ? *(const u32x4*)(q + qoff + bb * 32 + kk * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  };
  u32x4 qnext[XQG][2];
  load_q(0, qnext);
  for (int it = 0; it < XGPW; ++it) {
    const int q0 = blockIdx.y * XQ_PER_WG + (wave * XGPW + it) * 16 * XQG;
    if (q0 >= Lq) break;
    u32x4 qf[XQG][2];
    bool live[XQG];
#pragma unroll
    for (int g = 0; g < XQG; ++g) {
      live[g] = q0 + g * 16 + l15 < Lq;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) qf[g][bb] = qnext[g][bb];
    }
    load_q(it + 1, qnext);
    float m[XQG];                           // running max of the RAW scores (the positive scale commutes with max)
    f32x4 oacc[XQG][4], lacc[XQG];          // lacc: row sums of P by an all-ones MFMA (every row of the tile holds the same sum)
#pragma unroll
    for (int g = 0; g < XQG; ++g) {
      m[g] = -INFINITY;
      lacc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) oacc[g][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int c0 = 0; c0 < Sp; c0 += 128) {
      const unsigned kc[2] = {kb[0] + c0 * 128, kb[1] + c0 * 128};
      f32x4 s[XQG][8];
      // S^T tiles: 16 K fragments (tile t, dim block bb), each feeding both query groups, through the counted LDS read ring
      mfma_stream<16, 4>(
          [&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<(i >> 1) * 2048>{kc[i & 1]}; },
          [&](auto ic, const u32x4& kf) {
            constexpr int i = decltype(ic)::value, t = i >> 1, bb = i & 1;
#pragma unroll
            for (int g = 0; g < XQG; ++g) {
              if constexpr (bb == 0) s[g][t] = xmfma(kf, qf[g][0], f32x4{0.f, 0.f, 0.f, 0.f});
              else s[g][t] = xmfma(kf, qf[g][1], s[g][t]);
            }
          });
      const bool tail = c0 + 128 > Lk;   // uniform: this chunk holds padded keys
#pragma unroll
      for (int g = 0; g < XQG; ++g) {
        if (tail) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (c0 + t * 16 + kk * 4 + r >= Lk) s[g][t][r] = -INFINITY;
        }
        float cm = -INFINITY;
#pragma unroll
        for (int t = 0; t < 8; ++t) cm = fmaxf(cm, fmaxf(fmaxf(s[g][t][0], s[g][t][1]), fmaxf(s[g][t][2], s[g][t][3])));
        cm = fmaxf(cm, __shfl_xor(cm, 16));
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m[g], cm);
        const float corr = __builtin_amdgcn_exp2f((m[g] - mn) * scale_log2e);
        m[g] = mn;
        const float nm = -mn * scale_log2e;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[g][t][r] = __builtin_amdgcn_exp2f(fmaf(s[g][t][r], scale_log2e, nm));   // one fma + one exp per score
        lacc[g] *= corr;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oacc[g][dt] *= corr;
      }
      // O^T += V^T P^T per 32-key block.  P's k order inside the block is the accumulator order (keys 16 h + 4 kk + r), so an A fragment
      // is two transposed reads of 4 keys each; the reads of block blk + 1 are in flight under the MFMAs of block blk.
      unsigned vc[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vc[dt] = vb[dt] + c0 * 128;
      u32x2 vlo[2][4], vhi[2][4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { vlo[0][dt] = x_tr_read_off<0>(vc[dt]); vhi[0][dt] = x_tr_read_off<2048>(vc[dt]); }
      static_for<4>([&](auto bc) {
        constexpr int blk = decltype(bc)::value, cur = blk & 1;
        u32x4 pf[XQG];
#pragma unroll
        for (int g = 0; g < XQG; ++g) pf[g] = xpack8(s[g][2 * blk], s[g][2 * blk + 1]);
        if constexpr (blk < 3) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            vlo[cur ^ 1][dt] = x_tr_read_off<(blk + 1) * 4096>(vc[dt]);
            vhi[cur ^ 1][dt] = x_tr_read_off<(blk + 1) * 4096 + 2048>(vc[dt]);
          }
          asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vlo[cur][0]), "+v"(vlo[cur][1]), "+v"(vlo[cur][2]), "+v"(vlo[cur][3]), "+v"(vhi[cur][0]),
                       "+v"(vhi[cur][1]), "+v"(vhi[cur][2]), "+v"(vhi[cur][3]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[cur][0]), "+v"(vlo[cur][1]), "+v"(vlo[cur][2]), "+v"(vlo[cur][3]), "+v"(vhi[cur][0]),
                       "+v"(vhi[cur][1]), "+v"(vhi[cur][2]), "+v"(vhi[cur][3]));
        }
#pragma unroll
        for (int g = 0; g < XQG; ++g) lacc[g] = xmfma(ones, pf[g], lacc[g]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const u32x4 vf = u32x4{vlo[cur][dt][0], vlo[cur][dt][1], vhi[cur][dt][0], vhi[cur][dt][1]};
#pragma unroll
          for (int g = 0; g < XQG; ++g) oacc[g][dt] = xmfma(vf, pf[g], oacc[g][dt]);
        }
      });
    }
#pragma unroll
    for (int g = 0; g < XQG; ++g) {
      const float inv = 1.0f / lacc[g][0];
      if (live[g]) {
        const long ooff = ((long)b * Lq + q0 + g * 16 + l15) * ldo + (long)h * XD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {   // lane holds dims 16 dt + 4 kk + r of its query
          u32x2 w;
          w[0] = pack_bf16x2(oacc[g][dt][0] * inv, oacc[g][dt][1] * inv);
          w[1] = pack_bf16x2(oacc[g][dt][2] * inv, oacc[g][dt][3] * inv);
          *(u32x2*)(o + ooff + dt * 16 + kk * 4) = w;
        }
      }
    }
  }
}

// ---- CViT grid embedding: out[n, :] = sum_g w_ng latents[g, :],  w_ng = exp(-eps |x_n - g|^2) / sum_g' exp(...)  (cvit.py:435-438)
// One workgroup per query point.  The weights are evaluated for EVERY grid point (the grid is a trainable parameter, no lattice is
// assumed), 256 at a time; points whose weight is exactly 0 in fp32 (underflow, as in the reference) are dropped by an ordered
// compaction -- wave ballot + prefix popcount, waves in fixed order -- so the summation order is the grid order, deterministic.
// With eps = 1e5 on a 128 x 128 grid ~50 of 16384 points survive, which turns the dense N x G x latent_dim contraction
// (1.1 PFLOP for cfg4) into ~0.2 GFLOP without changing a bit of the result's definition.
__global__ __launch_bounds__(256) void grid_embed_kernel(const float* __restrict__ coords, const float* __restrict__ grid,
                                                         const float* __restrict__ latents, long N, int G, int LD, float eps,
                                                         float* __restrict__ out) {
  constexpr int CAP = 1024;
  __shared__ int l_idx[CAP];
  __shared__ float l_w[CAP];
  __shared__ int wave_cnt[4];
  __shared__ float red[4];
  const long n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float cx = coords[2 * n], cy = coords[2 * n + 1];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};   // latent dims tid, tid + 256, ... (LD <= 1024)
  float wsum = 0.0f;
  int cnt = 0;
  auto flush = [&]() {
    for (int e = 0; e < cnt; ++e) {
      const float w = l_w[e];
      const float* lr = latents + (long)l_idx[e] * LD;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        if (tid + 256 * kk < LD) acc[kk] += w * lr[tid + 256 * kk];
    }
  };
  for (int g0 = 0; g0 < G; g0 += 256) {
    const int g = g0 + tid;
    float w = 0.0f;
    if (g < G) {
      const float dx = cx - grid[2 * g], dy = cy - grid[2 * g + 1];
      w = expf(-eps * (dx * dx + dy * dy));
    }
    wsum += w;
    const unsigned long long bal = __ballot(w != 0.0f);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int base = cnt;
    for (int ww = 0; ww < wave; ++ww) base += wave_cnt[ww];
    const int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    if (cnt + tot > CAP) {   // uniform: the list would overflow -> fold what is there into the accumulators first
      flush();
      base -= cnt;
      cnt = 0;
      __syncthreads();
    }
    if (w != 0.0f) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      l_idx[pos] = g;
      l_w[pos] = w;
    }
    cnt += tot;
    __syncthreads();
  }
  flush();
  wsum = wave_sum(wsum);
  if (lane == 0) red[wave] = wsum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    if (tid + 256 * kk < LD) out[n * LD + tid + 256 * kk] = acc[kk] * inv;
}

// FourierEmbs (cvit.py:308-331): out[n] = [cos(c . K[:, j]), sin(c . K[:, j])], K (2, E/2)
__global__ void fourier_embed_kernel(const float* __restrict__ coords, const float* __restrict__ kern, long N, int half, float* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * half) return;
  const long n = idx / half;
  const int j = (int)(idx - n * half);
  const float dp = coords[2 * n] * kern[j] + coords[2 * n + 1] * kern[half + j];
  out[n * 2 * half + j] = cosf(dp);
  out[n * 2 * half + half + j] = sinf(dp);
}

// ---- CViT training: backward of softmax(q k^T / sqrt(D)) v with separate query / key sequences (fp32 arithmetic) ----------------
// Kernel A (one lane per query): softmax statistics m, 1/l, delta = dO . O, and dq = scale sum_j p_j (dp_j - delta) k_j.
// Kernel B (one lane per key, one workgroup per (batch, head, 256 keys, XB_QC queries)): dk_j = scale sum_i ds_ij q_i,
// dv_j = sum_i p_ij dO_i with the query rows broadcast from LDS; partial sums are added to fp32 buffers with atomics.
template <int D>
__global__ __launch_bounds__(256) void xattn_bwd_dq_kernel(const void* __restrict__ q, const void* __restrict__ k, const void* __restrict__ v,
                                                           const void* __restrict__ o, const void* __restrict__ dO, void* __restrict__ dq,
                                                           float* __restrict__ stats, int dtype, int n_head, int Lq, int Lk, long ldq,
                                                           long ldkv, long ldo, float scale) {
  __shared__ float ks[64][D];
  __shared__ float vs[64][D];
  const int bh = blockIdx.x, b = bh / n_head, h = bh - b * n_head;
  const int i = blockIdx.y * 256 + threadIdx.x;
  const bool live = i < Lq;
  const long qoff = ((long)b * Lq + (live ? i : 0)) * ldq + (long)h * D, ooff = ((long)b * Lq + (live ? i : 0)) * ldo + (long)h * D;
  float qr[D], gr[D], acc[D];
  float delta = 0.0f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    qr[d] = ldx(q, dtype, qoff + d) * scale;
    gr[d] = ldx(dO, dtype, ooff + d);
    delta += gr[d] * ldx(o, dtype, ooff + d);
    acc[d] = 0.0f;
  }
  float m = -INFINITY, l = 0.0f;
  for (int pass = 0; pass < 2; ++pass) {
    const float inv_l = pass ? 1.0f / l : 0.0f;
    for (int j0 = 0; j0 < Lk; j0 += 64) {
      const int nj = Lk - j0 < 64 ? Lk - j0 : 64;
      __syncthreads();
      for (int e = threadIdx.x; e < nj * D; e += 256) {
        const int j = e / D, d = e - j * D;
        const long off = ((long)b * Lk + j0 + j) * ldkv + (long)h * D + d;
        ks[j][d] = ldx(k, dtype, off);
        if (pass) vs[j][d] = ldx(v, dtype, off);
      }
      __syncthreads();
      for (int j = 0; j < nj; ++j) {
        float sc = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) sc += qr[d] * ks[j][d];
        if (!pass) {
          const float mn = fmaxf(m, sc);
          l = l * __expf(m - mn) + __expf(sc - mn);
          m = mn;
        } else {
          float dp = 0.0f;
#pragma unroll
          for (int d = 0; d < D; ++d) dp += gr[d] * vs[j][d];
          const float ds = __expf(sc - m) * inv_l * (dp - delta) * scale;
#pragma unroll
          for (int d = 0; d < D; ++d) acc[d] += ds * ks[j][d];
        }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int d = 0; d < D; ++d) stx(dq, dtype, qoff + d, acc[d]);
    float* st = stats + ((long)bh * Lq + i) * 3;
    st[0] = m; st[1] = 1.0f / l; st[2] = delta;
  }
}

constexpr int XB_QC = 512;   // queries per workgroup of the dk / dv kernel
template <int D>
__global__ __launch_bounds__(256) void xattn_bwd_dkv_kernel(const void* __restrict__ q, const void* __restrict__ k, const void* __restrict__ v,
                                                            const void* __restrict__ dO, const float* __restrict__ stats,
                                                            float* __restrict__ dk, float* __restrict__ dv, int dtype, int n_head, int Lq,
                                                            int Lk, long ldq, long ldkv, long ldo, long ldg, float scale) {
  __shared__ float qs[32][D];
  __shared__ float gs[32][D];
  __shared__ float sst[32][3];
  const int bh = blockIdx.x, b = bh / n_head, h = bh - b * n_head;
  const int j = blockIdx.y * 256 + threadIdx.x;
  const bool live = j < Lk;
  const long koff = ((long)b * Lk + (live ? j : 0)) * ldkv + (long)h * D;
  float kr[D], vr[D], ak[D], av[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { kr[d] = ldx(k, dtype, koff + d); vr[d] = ldx(v, dtype, koff + d); ak[d] = 0.0f; av[d] = 0.0f; }
  const int i_begin = blockIdx.z * XB_QC, i_end = min(Lq, i_begin + XB_QC);
  for (int i0 = i_begin; i0 < i_end; i0 += 32) {
    const int ni = i_end - i0 < 32 ? i_end - i0 : 32;
    __syncthreads();
    for (int e = threadIdx.x; e < ni * D; e += 256) {
      const int ii = e / D, d = e - ii * D;
      qs[ii][d] = ldx(q, dtype, ((long)b * Lq + i0 + ii) * ldq + (long)h * D + d);
      gs[ii][d] = ldx(dO, dtype, ((long)b * Lq + i0 + ii) * ldo + (long)h * D + d);
    }
    if (threadIdx.x < ni * 3) sst[threadIdx.x / 3][threadIdx.x % 3] = stats[((long)bh * Lq + i0 + threadIdx.x / 3) * 3 + threadIdx.x % 3];
    __syncthreads();
    for (int ii = 0; ii < ni; ++ii) {
      float sc = 0.0f, dp = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) { sc += qs[ii][d] * kr[d]; dp += gs[ii][d] * vr[d]; }
      const float p = __expf(sc * scale - sst[ii][0]) * sst[ii][1];
      const float ds = p * (dp - sst[ii][2]) * scale;
#pragma unroll
      for (int d = 0; d < D; ++d) { ak[d] += ds * qs[ii][d]; av[d] += p * gs[ii][d]; }
    }
  }
  if (live) {
    const long goff = ((long)b * Lk + j) * ldg + (long)h * D;
#pragma unroll
    for (int d = 0; d < D; ++d) { atomicAdd(dk + goff + d, ak[d]); atomicAdd(dv + goff + d, av[d]); }
  }
}

// ---- LayerNorm with affine, backward: dx and the (accumulated) gamma / beta gradients; a wave walks LNB_RPW rows -----------------
constexpr int LNB_RPW = 32;
__global__ __launch_bounds__(256) void ln_affine_bwd_kernel(const void* __restrict__ dy, int dy_dtype, const void* __restrict__ x, int x_dtype,
                                                            const float* __restrict__ gamma, long M, int C, float eps, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r0 = ((long)blockIdx.x * 4 + wave) * LNB_RPW;
  float ag[16], ab[16];   // this lane's columns lane, lane + 64, ... (C <= 1024)
#pragma unroll
  for (int kq = 0; kq < 16; ++kq) { ag[kq] = 0.0f; ab[kq] = 0.0f; }
  for (long row = r0; row < r0 + LNB_RPW && row < M; ++row) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ldx(x, x_dtype, row * C + c);
    const float mean = wave_sum(s) / C;
    float qv = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = ldx(x, x_dtype, row * C + c) - mean; qv += d * d; }
    const float rstd = rsqrtf(wave_sum(qv) / C + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kq = 0; kq < 16; ++kq) {
      const int c = lane + 64 * kq;
      if (c < C) {
        const float xh = (ldx(x, x_dtype, row * C + c) - mean) * rstd, g = ldx(dy, dy_dtype, row * C + c);
        ag[kq] += g * xh;
        ab[kq] += g;
        const float gg = g * (gamma ? gamma[c] : 1.0f);
        s1 += gg;
        s2 += gg * xh;
      }
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
#pragma unroll
    for (int kq = 0; kq < 16; ++kq) {
      const int c = lane + 64 * kq;
      if (c < C) {
        const float xh = (ldx(x, x_dtype, row * C + c) - mean) * rstd, gg = ldx(dy, dy_dtype, row * C + c) * (gamma ? gamma[c] : 1.0f);
        dx[row * C + c] = rstd * (gg - s1 - xh * s2);
      }
    }
  }
#pragma unroll
  for (int kq = 0; kq < 16; ++kq) {
    const int c = lane + 64 * kq;
    if (c < C) {
      if (dgamma) atomicAdd(dgamma + c, ag[kq]);
      if (dbeta) atomicAdd(dbeta + c, ab[kq]);
    }
  }
}

// ---- CViT grid embedding, backward: one workgroup per GRID POINT gathers from every query (no atomics, deterministic) -----------
// w_ng = exp(-eps |x_n - g|^2) / wsum_n;  dlatents[g] = sum_n w_ng dout_n;  dgrid[g] = sum_n w_ng (dout_n . (latents_g - out_n)) 2 eps (x_n - g)
__global__ __launch_bounds__(256) void grid_embed_wsum_kernel(const float* __restrict__ coords, const float* __restrict__ grid, long N, int G,
                                                              float eps, float* __restrict__ wsum) {
  __shared__ float red[4];
  const long n = blockIdx.x;
  const float cx = coords[2 * n], cy = coords[2 * n + 1];
  float s = 0.0f;
  for (int g = threadIdx.x; g < G; g += 256) {
    const float dx = cx - grid[2 * g], dy = cy - grid[2 * g + 1];
    s += expf(-eps * (dx * dx + dy * dy));
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) wsum[n] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void grid_embed_bwd_kernel(const float* __restrict__ coords, const float* __restrict__ grid,
                                                             const float* __restrict__ latents, const float* __restrict__ out,
                                                             const float* __restrict__ dout, const float* __restrict__ wsum, long N, int G,
                                                             int LD, float eps, float* __restrict__ dlatents, float* __restrict__ dgrid) {
  constexpr int CAP = 256;
  __shared__ int l_idx[CAP];
  __shared__ float l_w[CAP];
  __shared__ int wave_cnt[4];
  __shared__ float red[4];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float gx = grid[2 * g], gy = grid[2 * g + 1];
  float lat[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) lat[kq] = (tid + 256 * kq < LD) ? latents[(long)g * LD + tid + 256 * kq] : 0.0f;
  float dgx = 0.0f, dgy = 0.0f;
  int cnt = 0;
  auto flush = [&]() {
    for (int e = 0; e < cnt; ++e) {
      const float w = l_w[e];
      const long n = l_idx[e];
      float part = 0.0f;
#pragma unroll
      for (int kq = 0; kq < 4; ++kq)
        if (tid + 256 * kq < LD) {
          const float dv = dout[n * LD + tid + 256 * kq];
          acc[kq] += w * dv;
          part += dv * (lat[kq] - out[n * LD + tid + 256 * kq]);
        }
      part = wave_sum(part);
      __syncthreads();
      if (lane == 0) red[wave] = part;
      __syncthreads();
      const float dot = red[0] + red[1] + red[2] + red[3];
      const float f = w * dot * 2.0f * eps;
      dgx += f * (coords[2 * n] - gx);
      dgy += f * (coords[2 * n + 1] - gy);
    }
  };
  for (long n0 = 0; n0 < N; n0 += 256) {
    const long n = n0 + tid;
    float w = 0.0f;
    if (n < N) {
      const float dx = coords[2 * n] - gx, dy = coords[2 * n + 1] - gy;
      w = expf(-eps * (dx * dx + dy * dy)) / wsum[n];
    }
    const unsigned long long bal = __ballot(w != 0.0f);
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int base = cnt;
    for (int ww = 0; ww < wave; ++ww) base += wave_cnt[ww];
    const int tot = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    if (cnt + tot > CAP) {
      flush();
      base -= cnt;
      cnt = 0;
      __syncthreads();
    }
    if (w != 0.0f) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      l_idx[pos] = (int)n;
      l_w[pos] = w;
    }
    cnt += tot;
    __syncthreads();
  }
  flush();
#pragma unroll
  for (int kq = 0; kq < 4; ++kq)
    if (tid + 256 * kq < LD) dlatents[(long)g * LD + tid + 256 * kq] = acc[kq];
  if (tid == 0) { dgrid[2 * g] = dgx; dgrid[2 * g + 1] = dgy; }
}

// ---- spectral layer, backward ----------------------------------------------------------------------------------------------------
// With 'ortho' transforms the layer's spectral branch is y = A(M(B(x))): B = rfft2, M the per-mode contraction, A = irfft2 (a real-linear
// map that counts every interior half-spectrum column twice: D_j = 1 for j = 0 and j = W/2 (W even), 2 otherwise).  The adjoints are
// A* = D . rfft2 and B* = irfft2 . (1/D); M is diagonal in (i, j), so the D factors cancel in the input gradient
//     dx = irfft2(M^H rfft2(dy))
// and survive in the weight gradient            dW[c, o, wi, j] = D_j sum_b sum_{rows i -> wi} conj(X[b, c, i, j]) R[b, o, i, j],  R = rfft2(dy).
// Unnormalised hipFFT transforms are used throughout, so both carry the factor 1 / (H W).
__global__ void spectral_adjoint_kernel(const float2* __restrict__ R, const float* __restrict__ w_re, const float* __restrict__ w_im, long n,
                                        int Cin, int Cout, int H, int Wf, int m1, int m2, int wm1, int wm2, float scale, float2* __restrict__ dX) {
  const long total = n * Cin * H * Wf;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % Wf);
    long r = idx / Wf;
    const int i = (int)(r % H); r /= H;
    const int c = (int)(r % Cin);
    const long b = r / Cin;
    float2 acc = make_float2(0.f, 0.f);
    int wi = -1;
    if (j < m2) {
      if (i >= H - m1) wi = i - (H - m1);
      else if (i < m1) wi = i;
    }
    if (wi >= 0) {
      for (int o = 0; o < Cout; ++o) {
        const float2 rv = R[((b * Cout + o) * H + i) * Wf + j];
        const long wo = (((long)c * Cout + o) * wm1 + wi) * wm2 + j;
        const float wr = w_re[wo], wim = w_im[wo];
        acc.x += rv.x * wr + rv.y * wim;     // R * conj(W)
        acc.y += rv.y * wr - rv.x * wim;
      }
      acc.x *= scale; acc.y *= scale;
    }
    dX[idx] = acc;
  }
}

__global__ void spectral_wgrad_kernel(const float2* __restrict__ X, const float2* __restrict__ R, long n, int Cin, int Cout, int H, int W, int Wf,
                                      int m1, int m2, int wm1, int wm2, float scale, float* __restrict__ dw_re, float* __restrict__ dw_im) {
  const long total = (long)Cin * Cout * wm1 * wm2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % wm2);
    long r = idx / wm2;
    const int wi = (int)(r % wm1); r /= wm1;
    const int o = (int)(r % Cout);
    const int c = (int)(r / Cout);
    float2 acc = make_float2(0.f, 0.f);
    if (wi < m1 && j < m2) {
      const int itop = wi, ibot = H - m1 + wi;
      const bool use_top = itop < H - m1;            // a top row inside the bottom band was overwritten in the forward
      for (long b = 0; b < n; ++b) {
        for (int which = 0; which < 2; ++which) {
          if (which == 0 && !use_top) continue;
          const int i = which ? ibot : itop;
          const float2 xv = X[((b * Cin + c) * H + i) * Wf + j], rv = R[((b * Cout + o) * H + i) * Wf + j];
          acc.x += xv.x * rv.x + xv.y * rv.y;        // conj(X) * R
          acc.y += xv.x * rv.y - xv.y * rv.x;
        }
      }
      const float d = (j == 0 || ((W & 1) == 0 && j == W / 2)) ? 1.0f : 2.0f;
      acc.x *= scale * d; acc.y *= scale * d;
    }
    dw_re[idx] = acc.x;
    dw_im[idx] = acc.y;
  }
}

// ---- bilinear resize (+ crop) backward: every output pixel hands its gradient to its (up to) four source pixels --------------------
__global__ void resize_bwd_kernel(const void* __restrict__ dout, int d_dtype, long n_img, int C, int Hi, int Wi, int cy, int cx, long isn, long isc,
                                  long ish, long isw, int Ho, int Wo, long osn, long osc, long osh, long osw, float* __restrict__ din, int c_fast) {
  const long total = n_img * C * Ho * Wo;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int c, oy, ox;
    long img;
    if (c_fast) { c = (int)(idx % C); long r = idx / C; ox = (int)(r % Wo); r /= Wo; oy = (int)(r % Ho); img = r / Ho; }
    else { ox = (int)(idx % Wo); long r = idx / Wo; oy = (int)(r % Ho); r /= Ho; c = (int)(r % C); img = r / C; }
    float fy = sy * ((float)oy + 0.5f) - 0.5f, fx = sx * ((float)ox + 0.5f) - 0.5f;
    fy = fy < 0.0f ? 0.0f : fy;
    fx = fx < 0.0f ? 0.0f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float g = ldx(dout, d_dtype, img * osn + c * osc + oy * osh + ox * osw);
    const long base = img * isn + c * isc;
    atomicAdd(din + base + (y0 + cy) * ish + (x0 + cx) * isw, g * (1.0f - ly) * (1.0f - lx));
    atomicAdd(din + base + (y0 + cy) * ish + (x1 + cx) * isw, g * (1.0f - ly) * lx);
    atomicAdd(din + base + (y1 + cy) * ish + (x0 + cx) * isw, g * ly * (1.0f - lx));
    atomicAdd(din + base + (y1 + cy) * ish + (x1 + cx) * isw, g * ly * lx);
  }
}

inline unsigned grid_for(long total, int block = 256) {
  long g = (total + block - 1) / block;
  return (unsigned)(g > 1048576 ? 1048576 : (g < 1 ? 1 : g));
}

// ---- hipFFT plans, cached per thread and per (device, H, W, batch, direction) ---------------------------------------------------
// A plan carries its stream (hipfftSetStream) and its work area, so two host threads must never share one: the cache is
// thread_local, which also makes it lock-free.  The device is part of the key: a plan belongs to the device it was made on.
int get_plan(int H, int W, long batch, int inverse, hipfftHandle* out) {
  thread_local std::map<std::tuple<int, int, int, long, int>, hipfftHandle> plans;
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto key = std::make_tuple(dev, H, W, batch, inverse);
  auto it = plans.find(key);
  if (it != plans.end()) { *out = it->second; return 0; }
  hipfftHandle p;
  int n[2] = {H, W};
  if (hipfftPlanMany(&p, 2, n, nullptr, 1, 0, nullptr, 1, 0, inverse ? HIPFFT_C2R : HIPFFT_R2C, (int)batch) != HIPFFT_SUCCESS) return -1;
  plans[key] = p;
  *out = p;
  return 0;
}

}  // namespace

extern "C" int tante_im2col(const void* x, int x_dtype, int nchw, int64_t n_img, int C, int H, int W, int kh, int kw, int sh, int sw,
                            int ph, int pw, int korder, void* cols, int cols_dtype, void* stream) {
  if (!x || !cols || n_img <= 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0)
    TANTE_FAIL(-1, "tante_im2col: bad argument");
  const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
  if (Ho <= 0 || Wo <= 0) TANTE_FAIL(-1, "tante_im2col: empty output");
  const long total = (long)n_img * Ho * Wo * C * kh * kw;
  const int eb = x_dtype == TANTE_BF16 ? 2 : 4;
  if (nchw && korder == 0 && (x_dtype == TANTE_F32 || (x_dtype == TANTE_BF16 && cols_dtype == TANTE_BF16)) && kh == kw && sh == kh && sw == kw &&
      ph == pw && ph < kh && (kh == 2 || kh == 4) && tante_opt("TANTE_IM2COL_TILED", 1)) {
    // channels per pass: the C x P segments of 64 floats (+ 1 pad) within 48 KB of LDS (three workgroups per CU and more)
    {
      int CC = C;
      while ((size_t)CC * kh * 65 * 4 > 48 * 1024) CC = (CC + 1) / 2;
      const size_t lds = (size_t)CC * kh * 65 * 4;
      const int TW = 64 / kh;
      const long blocks = (long)n_img * Ho * ((Wo + TW - 1) / TW);
      if (blocks > 2147483647L) TANTE_FAIL(-2, "tante_im2col: grid too large");
      const bool bf = cols_dtype == TANTE_BF16;
      hipStream_t s_ = (hipStream_t)stream;
      if (x_dtype == TANTE_BF16) {
        if (kh == 4) hipLaunchKernelGGL((im2col_nchw_tile_kernel<4, true, true>), dim3((unsigned)blocks), dim3(256), lds, s_, x, C, H, W, ph, Ho, Wo, CC, cols);
        else hipLaunchKernelGGL((im2col_nchw_tile_kernel<2, true, true>), dim3((unsigned)blocks), dim3(256), lds, s_, x, C, H, W, ph, Ho, Wo, CC, cols);
      } else if (kh == 4 && bf) hipLaunchKernelGGL((im2col_nchw_tile_kernel<4, true>), dim3((unsigned)blocks), dim3(256), lds, s_, (const float*)x, C, H, W, ph, Ho, Wo, CC, cols);
      else if (kh == 4) hipLaunchKernelGGL((im2col_nchw_tile_kernel<4, false>), dim3((unsigned)blocks), dim3(256), lds, s_, (const float*)x, C, H, W, ph, Ho, Wo, CC, cols);
      else if (bf) hipLaunchKernelGGL((im2col_nchw_tile_kernel<2, true>), dim3((unsigned)blocks), dim3(256), lds, s_, (const float*)x, C, H, W, ph, Ho, Wo, CC, cols);
      else hipLaunchKernelGGL((im2col_nchw_tile_kernel<2, false>), dim3((unsigned)blocks), dim3(256), lds, s_, (const float*)x, C, H, W, ph, Ho, Wo, CC, cols);
      TANTE_CHECK_LAUNCH();
      return 0;
    }
  }
  if (!nchw && korder == 1 && x_dtype == cols_dtype && C % (16 / eb) == 0 && (((uintptr_t)x | (uintptr_t)cols) & 15) == 0) {
    const long pieces = total / (16 / eb);
    if (eb == 2) hipLaunchKernelGGL(im2col_vec_kernel<2>, dim3(grid_for(pieces)), dim3(256), 0, (hipStream_t)stream, (const char*)x, (long)n_img, C, H, W,
                                    kh, kw, sh, sw, ph, pw, Ho, Wo, (char*)cols);
    else hipLaunchKernelGGL(im2col_vec_kernel<4>, dim3(grid_for(pieces)), dim3(256), 0, (hipStream_t)stream, (const char*)x, (long)n_img, C, H, W,
                            kh, kw, sh, sw, ph, pw, Ho, Wo, (char*)cols);
    TANTE_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, nchw, (long)n_img, C, H, W, kh, kw,
                     sh, sw, ph, pw, Ho, Wo, korder, cols, cols_dtype);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_avgpool_nhwc(const void* x, int x_dtype, int64_t n_img, int H, int W, int C, int Ht, int Wt, int act, void* y,
                                  int y_dtype, void* stream) {
  if (!x || !y || n_img <= 0 || H <= 0 || W <= 0 || C <= 0 || Ht <= 0 || Wt <= 0) TANTE_FAIL(-1, "tante_avgpool_nhwc: bad argument");
  hipLaunchKernelGGL(avgpool_kernel, dim3(grid_for((long)n_img * Ht * Wt * C)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, (long)n_img, H, W,
                     C, Ht, Wt, act, y, y_dtype);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_avgpool_nhwc_bwd(const void* dy, int dy_dtype, int64_t n_img, int H, int W, int C, int Ht, int Wt, void* dx, int dx_dtype,
                                      void* stream) {
  if (!dy || !dx || n_img <= 0 || H <= 0 || W <= 0 || C <= 0 || Ht <= 0 || Wt <= 0 || Ht > H || Wt > W) TANTE_FAIL(-1, "tante_avgpool_nhwc_bwd: bad argument");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long)n_img * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, dy_dtype, (long)n_img, H, W,
                     C, Ht, Wt, dx, dx_dtype);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_col2im_nhwc_sized(const void* cols, int cols_dtype, int64_t n_img, int Hi, int Wi, int P, int stride, int pad, int Cout,
                                       const float* bias, int Hf, int Wf, void* out, int out_dtype, void* stream) {
  if (!cols || !out || n_img <= 0 || Hi <= 0 || Wi <= 0 || P <= 0 || stride <= 0 || pad < 0 || Cout <= 0 || Hf <= 0 || Wf <= 0)
    TANTE_FAIL(-1, "tante_col2im_nhwc_sized: bad argument");
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((long)n_img * Hf * Wf * Cout)), dim3(256), 0, (hipStream_t)stream, cols, cols_dtype, (long)n_img,
                     Hi, Wi, P, stride, pad, Cout, bias, Hf, Wf, out, out_dtype);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_col2im_nhwc(const void* cols, int cols_dtype, int64_t n_img, int Hi, int Wi, int P, int stride, int pad, int Cout,
                                 const float* bias, void* out, int out_dtype, void* stream) {
  if (!cols || !out || n_img <= 0 || Hi <= 0 || Wi <= 0 || P <= 0 || stride <= 0 || pad < 0 || Cout <= 0) TANTE_FAIL(-1, "tante_col2im_nhwc: bad argument");
  const int Hf = (Hi - 1) * stride - 2 * pad + P, Wf = (Wi - 1) * stride - 2 * pad + P;
  if (Hf <= 0 || Wf <= 0) TANTE_FAIL(-1, "tante_col2im_nhwc: empty output");
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((long)n_img * Hf * Wf * Cout)), dim3(256), 0, (hipStream_t)stream, cols, cols_dtype, (long)n_img,
                     Hi, Wi, P, stride, pad, Cout, bias, Hf, Wf, out, out_dtype);
  TANTE_CHECK_LAUNCH();
  return 0;
}

// The same resize from a channels-LAST source into a channels-FIRST destination (the last stage of a padded decoder, enc_dec_cnn.py:164-184
// as used by enc_dec_fno.py: (n, 512, 512, 32) bf16 -> (n, 32, 512, 512) fp32): the generic kernel above walks the output, so a wave reads
// 64 pixels x 2 bytes at a stride of C elements -- a 64-byte segment fetched per value (158 us for 16.8 M outputs).  Here a workgroup owns
// 64 output columns of one output row: the two source rows' <= 66 pixels x CC channels arrive as contiguous runs (channels fastest)
// into LDS, and every channel leaves as 64 consecutive values.  Same expression, same order: bit-identical to the generic kernel.
// Needs a horizontal scale Wi / Wo <= 1 (the 66 source columns).
template <int CC>
__global__ __launch_bounds__(256) void resize_cl2cf_kernel(const void* __restrict__ in, int in_dtype, int C, int Hi, int Wi, int cy, int cx, long isn,
                                                           long ish, long isw, int Ho, int Wo, long osn, long osc, long osh, int act,
                                                           void* __restrict__ out, int out_dtype) {
  __shared__ float tile[2][66][CC + 1];
  const int tiles_w = (Wo + 63) / 64;
  const int bx = blockIdx.x % tiles_w, oy = (blockIdx.x / tiles_w) % Ho;
  const long img = blockIdx.x / ((long)tiles_w * Ho);
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  float fy = sy * ((float)oy + 0.5f) - 0.5f;
  fy = fy < 0.0f ? 0.0f : fy;
  const int y0 = (int)fy, y1 = y0 + (y0 < Hi - 1 ? 1 : 0);
  const float ly = fy - (float)y0;
  const int ox0 = bx * 64;
  float fx0 = sx * ((float)ox0 + 0.5f) - 0.5f;
  fx0 = fx0 < 0.0f ? 0.0f : fx0;
  const int xs0 = (int)fx0, ncol = min(66, Wi - xs0);
  const int ox = ox0 + (threadIdx.x & 63);
  float fx = sx * ((float)ox + 0.5f) - 0.5f;
  fx = fx < 0.0f ? 0.0f : fx;
  const int x0 = (int)fx, x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
  const float lx = fx - (float)x0;
  const int s0 = x0 - xs0, s1 = x1 - xs0;
  const bool vec16 = in_dtype == TANTE_BF16 && C % 8 == 0 && ((uintptr_t)in % 16) == 0 && isn % 8 == 0 && ish % 8 == 0 && isw % 8 == 0;
  for (int c0 = 0; c0 < C; c0 += CC) {
    const int cc = min(CC, C - c0);
    if (vec16) {      // bf16 pixels of whole 16-byte pieces: one load per 8 channels (the scalar form below: 16 two-byte loads per thread)
      constexpr int PCS = CC / 8;
      for (int e = threadIdx.x; e < 2 * 66 * PCS; e += 256) {
        const int r = e / (66 * PCS), rem = e - r * (66 * PCS), sc = rem / PCS, pc = rem - sc * PCS;
        if (sc < ncol && 8 * pc < cc) {
          const u32x4 u = *(const u32x4*)((const unsigned short*)in + img * isn + ((r ? y1 : y0) + cy) * ish + (xs0 + sc + cx) * isw + c0 + 8 * pc);
          float* t = &tile[r][sc][8 * pc];
#pragma unroll
          for (int q = 0; q < 4; ++q) { t[2 * q] = bf16_lo(u[q]); t[2 * q + 1] = bf16_hi(u[q]); }
        }
      }
    } else {
      for (int e = threadIdx.x; e < 2 * 66 * CC; e += 256) {
        const int r = e / (66 * CC), rem = e - r * (66 * CC), sc = rem / CC, c = rem - sc * CC;
        if (sc < ncol && c < cc) tile[r][sc][c] = ldx(in, in_dtype, img * isn + ((r ? y1 : y0) + cy) * ish + (xs0 + sc + cx) * isw + c0 + c);
      }
    }
    __syncthreads();
    if (ox < Wo) {
      for (int c = threadIdx.x >> 6; c < cc; c += 4) {
        const float v00 = tile[0][s0][c], v01 = tile[0][s1][c], v10 = tile[1][s0][c], v11 = tile[1][s1][c];
        const float v = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
        stx(out, out_dtype, img * osn + (c0 + c) * osc + oy * osh + ox, apply_act(v, act));
      }
    }
    __syncthreads();
  }
}

extern "C" int tante_resize_bilinear(const void* in, int in_dtype, int64_t n_img, int C, int Hi, int Wi, int crop_y, int crop_x,
                                     int64_t isn, int64_t isc, int64_t ish, int64_t isw, int Ho, int Wo, int64_t osn, int64_t osc,
                                     int64_t osh, int64_t osw, int act, void* out, int out_dtype, void* stream) {
  if (!in || !out || n_img <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) TANTE_FAIL(-1, "tante_resize_bilinear: bad argument");
  if (isc == 1 && osw == 1 && Wi <= Wo && tante_opt("TANTE_RESIZE_TILED", 1)) {
    const long blocks = (long)n_img * Ho * ((Wo + 63) / 64);
    if (blocks <= 2147483647L) {
      hipLaunchKernelGGL((resize_cl2cf_kernel<32>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, in_dtype, C, Hi, Wi, crop_y, crop_x,
                         (long)isn, (long)ish, (long)isw, Ho, Wo, (long)osn, (long)osc, (long)osh, act, out, out_dtype);
      TANTE_CHECK_LAUNCH();
      return 0;
    }
  }
  hipLaunchKernelGGL(resize_kernel, dim3(grid_for((long)n_img * C * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, in, in_dtype, (long)n_img, C,
                     Hi, Wi, crop_y, crop_x, (long)isn, (long)isc, (long)ish, (long)isw, Ho, Wo, (long)osn, (long)osc, (long)osh, (long)osw, act,
                     out, out_dtype, (int)(osc == 1));
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_layernorm_affine(const void* x, int x_dtype, int64_t M, int C, float eps, const float* gamma, const float* beta, void* y,
                                      int y_dtype, void* stream) {
  if (!x || !y || M <= 0 || C <= 0) TANTE_FAIL(-1, "tante_layernorm_affine: bad argument");
  const dim3 grid((unsigned)((M + 3) / 4));
  const bool vec = x_dtype == TANTE_F32 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) % 16) == 0;
  if (vec && C == 256)
    hipLaunchKernelGGL(ln_affine_vec_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (long)M, eps, gamma, beta, y, y_dtype);
  else if (vec && C == 512)
    hipLaunchKernelGGL(ln_affine_vec_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (long)M, eps, gamma, beta, y, y_dtype);
  else
    hipLaunchKernelGGL(ln_affine_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x_dtype, (long)M, C, eps, gamma, beta, y, y_dtype);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t tante_spectral_workspace_bytes(int64_t n, int Cin, int Cout, int H, int W) {
  const long Wf = W / 2 + 1;
  return (long)sizeof(float2) * n * H * Wf * ((long)Cin + Cout);
}

extern "C" int tante_spectral_layer(const float* x, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1, int wm2,
                                    int modes1, int modes2, const float* w0, const float* b0, int Cout, int act, float* out, void* work,
                                    int64_t work_bytes, void* stream) {
  return tante_spectral_layer_c(x, n, Cin, H, W, w_re, w_im, wm1, wm2, modes1, modes2, w0, b0, Cout, act, out, work, work_bytes, TANTE_F32, stream);
}

extern "C" int tante_spectral_layer_c(const float* x, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1, int wm2,
                                      int modes1, int modes2, const float* w0, const float* b0, int Cout, int act, float* out, void* work,
                                      int64_t work_bytes, int compute, void* stream) {
  if (compute != TANTE_F32 && compute != TANTE_BF16) TANTE_FAIL(-1, "tante_spectral_layer_c: compute must be TANTE_F32 or TANTE_BF16");
  if (!x || !w_re || !w_im || !w0 || !out || !work || n <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0)
    TANTE_FAIL(-1, "tante_spectral_layer: bad argument");
  if (work_bytes < tante_spectral_workspace_bytes(n, Cin, Cout, H, W)) TANTE_FAIL(-1, "tante_spectral_layer: workspace too small");
  const int Wf = W / 2 + 1;
  const int m1 = modes1 < H ? modes1 : H, m2 = modes2 < Wf ? modes2 : Wf;
  if (m1 > wm1 || m2 > wm2) TANTE_FAIL(-1, "tante_spectral_layer: weight holds fewer modes (%d, %d) than used (%d, %d)", wm1, wm2, m1, m2);
  hipStream_t s = (hipStream_t)stream;
  // few kept modes: both transforms as skinny fp32 matrix products, the 1x1 conv in the same output pass (spectral_dft.hip);
  // TANTE_SPECTRAL_DFT = 0 (tante_set_option) keeps the hipFFT path below for every shape (A/B timing, tests)
  if (tante_opt("TANTE_SPECTRAL_DFT", 1) && tante_spectral_dft_supported(n, Cin, Cout, H, W, m1, m2) &&
      tante_spectral_dft_workspace_bytes(n, Cin, Cout, H, m1, m2) <= work_bytes) {
    const int rc = tante_spectral_dft_forward(x, n, Cin, H, W, w_re, w_im, wm1, wm2, m1, m2, w0, b0, Cout, act, out, work, compute, s);
    if (rc) TANTE_FAIL(rc, "tante_spectral_layer: truncated-DFT launch failed");
    return 0;
  }
  float2* X = (float2*)work;
  float2* Y = X + (long)n * Cin * H * Wf;
  hipfftHandle fwd, inv;
  if (get_plan(H, W, (long)n * Cin, 0, &fwd) || get_plan(H, W, (long)n * Cout, 1, &inv)) TANTE_FAIL(-3, "tante_spectral_layer: hipfftPlanMany failed");
  if (hipfftSetStream(fwd, s) != HIPFFT_SUCCESS || hipfftSetStream(inv, s) != HIPFFT_SUCCESS) TANTE_FAIL(-3, "tante_spectral_layer: hipfftSetStream failed");
  if (hipfftExecR2C(fwd, (hipfftReal*)x, (hipfftComplex*)X) != HIPFFT_SUCCESS) TANTE_FAIL(-3, "tante_spectral_layer: R2C failed");
  // both 'ortho' factors (1/sqrt(HW) forward and inverse) ride the contraction
  hipLaunchKernelGGL(spectral_modes_kernel, dim3(grid_for((long)n * Cout * H * Wf)), dim3(256), 0, s, X, w_re, w_im, (long)n, Cin, Cout, H, Wf,
                     m1, m2, wm1, wm2, 1.0f / ((float)H * (float)W), Y);
  TANTE_CHECK_LAUNCH();
  if (hipfftExecC2R(inv, (hipfftComplex*)Y, (hipfftReal*)out) != HIPFFT_SUCCESS) TANTE_FAIL(-3, "tante_spectral_layer: C2R failed");
  const size_t lds = ((size_t)Cin * 64 + (size_t)Cout * Cin) * sizeof(float);
  if (lds > 160 * 1024) TANTE_FAIL(-2, "tante_spectral_layer: 1x1 weight %d x %d does not fit LDS", Cout, Cin);
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)conv1x1_add_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const long HW = (long)H * W;
  hipLaunchKernelGGL(conv1x1_add_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)n), dim3(256), lds, s, x, w0, b0, out, HW, Cin, Cout, act, out);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_spectral_bf16out_supported(int64_t n, int Cin, int Cout, int H, int W, int modes1, int modes2) {
  if (n <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  const int Wf = W / 2 + 1;
  const int m1 = modes1 < H ? modes1 : H, m2 = modes2 < Wf ? modes2 : Wf;
  return tante_opt("TANTE_SPECTRAL_DFT", 1) && tante_opt("TANTE_SPECTRAL_BF16OUT", 1) && tante_spectral_dft_bf16out_supported(n, Cin, Cout, H, W, m1, m2);
}

extern "C" int tante_spectral_layer_bf16out(const float* x, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im, int wm1, int wm2,
                                            int modes1, int modes2, const float* w0, const float* b0, int Cout, int act, void* out, void* work,
                                            int64_t work_bytes, void* stream) {
  if (!x || !w_re || !w_im || !w0 || !out || !work || n <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0)
    TANTE_FAIL(-1, "tante_spectral_layer_bf16out: bad argument");
  if (!tante_spectral_bf16out_supported(n, Cin, Cout, H, W, modes1, modes2)) TANTE_FAIL(-2, "tante_spectral_layer_bf16out: shape without the bf16-output form");
  if (((uintptr_t)x % 16) || ((uintptr_t)out % 16)) TANTE_FAIL(-1, "tante_spectral_layer_bf16out: x and out must be 16-byte aligned");
  const int Wf = W / 2 + 1;
  const int m1 = modes1 < H ? modes1 : H, m2 = modes2 < Wf ? modes2 : Wf;
  if (m1 > wm1 || m2 > wm2) TANTE_FAIL(-1, "tante_spectral_layer_bf16out: weight holds fewer modes (%d, %d) than used (%d, %d)", wm1, wm2, m1, m2);
  if (work_bytes < tante_spectral_dft_workspace_bytes(n, Cin, Cout, H, m1, m2)) TANTE_FAIL(-1, "tante_spectral_layer_bf16out: workspace too small");
  const int rc = tante_spectral_dft_forward(x, n, Cin, H, W, w_re, w_im, wm1, wm2, m1, m2, w0, b0, Cout, act, (float*)out, work, TANTE_BF16,
                                            (hipStream_t)stream, 1);
  if (rc) TANTE_FAIL(rc, "tante_spectral_layer_bf16out: truncated-DFT launch failed");
  return 0;
}

// tante_spectral_layer_c with (a) images of x `x_istride` elements apart (a frame of every batch item inside a rollout buffer) and (b) the
// output layout chosen: 0 fp32 (n, Cout, H, W), 1 bf16 (n, Cout, H, W), 2 fp32 channels-last rows ((n h w), Cout).  bf16 compute mode, shapes
// of the truncated-DFT path's split-bf16 kernels only: ask tante_spectral_layer_x_supported first (-2 otherwise).
extern "C" int tante_spectral_layer_x_supported(int64_t n, int Cin, int Cout, int H, int W, int modes1, int modes2, int strided, int out_mode) {
  if (n <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || out_mode < 0 || out_mode > 2) return 0;
  const int Wf = W / 2 + 1;
  const int m1 = modes1 < H ? modes1 : H, m2 = modes2 < Wf ? modes2 : Wf;
  if (!tante_opt("TANTE_SPECTRAL_DFT", 1) || (out_mode == 1 && !tante_opt("TANTE_SPECTRAL_BF16OUT", 1))) return 0;
  return tante_spectral_dft_x_supported(n, Cin, Cout, H, W, m1, m2, strided, out_mode == 1, out_mode == 2);
}

extern "C" int tante_spectral_layer_x(const float* x, int64_t x_istride, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im,
                                      int wm1, int wm2, int modes1, int modes2, const float* w0, const float* b0, int Cout, int act, void* out,
                                      int out_mode, void* work, int64_t work_bytes, void* stream) {
  if (!x || !w_re || !w_im || !w0 || !out || !work || n <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0)
    TANTE_FAIL(-1, "tante_spectral_layer_x: bad argument");
  if (x_istride < (int64_t)Cin * H * W || x_istride % 4) TANTE_FAIL(-1, "tante_spectral_layer_x: image stride must be >= Cin H W and a multiple of 4");
  const int strided = x_istride != (int64_t)Cin * H * W;
  if (!tante_spectral_layer_x_supported(n, Cin, Cout, H, W, modes1, modes2, strided, out_mode))
    TANTE_FAIL(-2, "tante_spectral_layer_x: shape / layout without a split-bf16 kernel (use tante_spectral_layer_c on a dense copy)");
  if (((uintptr_t)x % 16) || ((uintptr_t)out % 16) || ((uintptr_t)w0 % 16) || (b0 && ((uintptr_t)b0 % 16)))
    TANTE_FAIL(-1, "tante_spectral_layer_x: x, out, w0 and b0 must be 16-byte aligned");
  const int Wf = W / 2 + 1;
  const int m1 = modes1 < H ? modes1 : H, m2 = modes2 < Wf ? modes2 : Wf;
  if (m1 > wm1 || m2 > wm2) TANTE_FAIL(-1, "tante_spectral_layer_x: weight holds fewer modes (%d, %d) than used (%d, %d)", wm1, wm2, m1, m2);
  if (work_bytes < tante_spectral_dft_workspace_bytes(n, Cin, Cout, H, m1, m2)) TANTE_FAIL(-1, "tante_spectral_layer_x: workspace too small");
  const int rc = tante_spectral_dft_forward(x, n, Cin, H, W, w_re, w_im, wm1, wm2, m1, m2, w0, b0, Cout, act, (float*)out, work, TANTE_BF16,
                                            (hipStream_t)stream, out_mode == 1, (long)x_istride, out_mode == 2);
  if (rc) TANTE_FAIL(rc, "tante_spectral_layer_x: truncated-DFT launch failed");
  return 0;
}

extern "C" int tante_resize_bilinear_bwd(const void* dout, int d_dtype, int64_t n_img, int C, int Hi, int Wi, int crop_y, int crop_x, int64_t isn,
                                         int64_t isc, int64_t ish, int64_t isw, int Ho, int Wo, int64_t osn, int64_t osc, int64_t osh,
                                         int64_t osw, float* din, void* stream) {
  if (!dout || !din || n_img <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) TANTE_FAIL(-1, "tante_resize_bilinear_bwd: bad argument");
  hipLaunchKernelGGL(resize_bwd_kernel, dim3(grid_for((long)n_img * C * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, dout, d_dtype, (long)n_img, C,
                     Hi, Wi, crop_y, crop_x, (long)isn, (long)isc, (long)ish, (long)isw, Ho, Wo, (long)osn, (long)osc, (long)osh, (long)osw, din,
                     (int)(osc == 1));
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_spectral_layer_bwd(const float* x, const float* dy, int64_t n, int Cin, int H, int W, const float* w_re, const float* w_im,
                                        int wm1, int wm2, int modes1, int modes2, const float* w0t, int Cout, float* dx, float* dw_re,
                                        float* dw_im, void* work, int64_t work_bytes, void* stream) {
  if (!x || !dy || !w_re || !w_im || !w0t || !dx || !dw_re || !dw_im || !work || n <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0)
    TANTE_FAIL(-1, "tante_spectral_layer_bwd: bad argument");
  if (work_bytes < tante_spectral_workspace_bytes(n, Cin, Cout, H, W)) TANTE_FAIL(-1, "tante_spectral_layer_bwd: workspace too small");
  const int Wf = W / 2 + 1;
  const int m1 = modes1 < H ? modes1 : H, m2 = modes2 < Wf ? modes2 : Wf;
  if (m1 > wm1 || m2 > wm2) TANTE_FAIL(-1, "tante_spectral_layer_bwd: weight holds fewer modes than used");
  hipStream_t s = (hipStream_t)stream;
  float2* XS = (float2*)work;                       // Cin-sized spectrum: first dX, later X
  float2* R = XS + (long)n * Cin * H * Wf;          // Cout-sized spectrum of dy
  hipfftHandle fwd_in, fwd_out, inv_in;
  if (get_plan(H, W, (long)n * Cin, 0, &fwd_in) || get_plan(H, W, (long)n * Cout, 0, &fwd_out) || get_plan(H, W, (long)n * Cin, 1, &inv_in))
    TANTE_FAIL(-3, "tante_spectral_layer_bwd: hipfftPlanMany failed");
  if (hipfftSetStream(fwd_in, s) != HIPFFT_SUCCESS || hipfftSetStream(fwd_out, s) != HIPFFT_SUCCESS || hipfftSetStream(inv_in, s) != HIPFFT_SUCCESS)
    TANTE_FAIL(-3, "tante_spectral_layer_bwd: hipfftSetStream failed");
  const float scale = 1.0f / ((float)H * (float)W);
  if (hipfftExecR2C(fwd_out, (hipfftReal*)dy, (hipfftComplex*)R) != HIPFFT_SUCCESS) TANTE_FAIL(-3, "tante_spectral_layer_bwd: R2C(dy) failed");
  hipLaunchKernelGGL(spectral_adjoint_kernel, dim3(grid_for((long)n * Cin * H * Wf)), dim3(256), 0, s, R, w_re, w_im, (long)n, Cin, Cout, H, Wf, m1,
                     m2, wm1, wm2, scale, XS);
  TANTE_CHECK_LAUNCH();
  if (hipfftExecC2R(inv_in, (hipfftComplex*)XS, (hipfftReal*)dx) != HIPFFT_SUCCESS) TANTE_FAIL(-3, "tante_spectral_layer_bwd: C2R failed");
  // + the 1x1 conv's input gradient: dx += W0^T dy  (same kernel as the forward, roles of Cin / Cout swapped)
  const size_t lds = ((size_t)Cout * 64 + (size_t)Cin * Cout) * sizeof(float);
  if (lds > 160 * 1024) TANTE_FAIL(-2, "tante_spectral_layer_bwd: 1x1 weight %d x %d does not fit LDS", Cout, Cin);
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)conv1x1_add_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const long HW = (long)H * W;
  hipLaunchKernelGGL(conv1x1_add_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)n), dim3(256), lds, s, dy, w0t, (const float*)nullptr, dx, HW, Cout,
                     Cin, TANTE_ACT_NONE, dx);
  TANTE_CHECK_LAUNCH();
  if (hipfftExecR2C(fwd_in, (hipfftReal*)x, (hipfftComplex*)XS) != HIPFFT_SUCCESS) TANTE_FAIL(-3, "tante_spectral_layer_bwd: R2C(x) failed");
  hipLaunchKernelGGL(spectral_wgrad_kernel, dim3(grid_for((long)Cin * Cout * wm1 * wm2)), dim3(256), 0, s, XS, R, (long)n, Cin, Cout, H, W, Wf, m1, m2,
                     wm1, wm2, scale, dw_re, dw_im);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_cross_attention(const void* q, const void* k, const void* v, void* o, int dtype, int64_t n_batch, int n_head, int D, int Lq,
                                     int Lk, int64_t ldq, int64_t ldkv, int64_t ldo, void* stream) {
  return tante_cross_attention_q(q, k, v, o, dtype, n_batch, n_head, D, Lq, Lk, ldq, ldkv, ldo, Lq, stream);
}

extern "C" int tante_cross_attention_q(const void* q, const void* k, const void* v, void* o, int dtype, int64_t n_batch, int n_head, int D, int Lq,
                                       int Lk, int64_t ldq, int64_t ldkv, int64_t ldo, int64_t q_batch_rows, void* stream) {
  if (!q || !k || !v || !o || n_batch <= 0 || n_head <= 0 || Lq <= 0 || Lk <= 0) TANTE_FAIL(-1, "tante_cross_attention: bad argument");
  if (q_batch_rows != 0 && q_batch_rows < Lq) TANTE_FAIL(-1, "tante_cross_attention_q: q_batch_rows = %lld must be 0 (shared queries) or >= Lq", (long long)q_batch_rows);
  const long qbr = (long)q_batch_rows;
  if ((Lq + 255) / 256 > 65535 || n_batch * n_head > 2147483647L) TANTE_FAIL(-2, "tante_cross_attention: grid too large");
  const float scale = 1.0f / sqrtf((float)D);
  hipStream_t s = (hipStream_t)stream;
  const bool force_valu = tante_opt("TANTE_XATTN_VALU", 0) != 0;
  if (dtype == TANTE_BF16 && D == XD && Lk <= 512 && !force_valu && ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 4 == 0 &&
      ((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)o % 8) == 0) {
    const int Sp = ((Lk + 127) / 128) * 128;
    const size_t lds = (size_t)Sp * XD * 2 * 2;
    static size_t attr = 0;
    if (lds > attr) {
      hipFuncSetAttribute((const void*)xattn_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = lds;
    }
    // iterations per wave: as many as keep >= 512 workgroups in the launch
    int gpw = XGPW_MAX;
    while (gpw > 1 && n_batch * n_head * (long)((Lq + XWAVES * gpw * 16 * XQG - 1) / (XWAVES * gpw * 16 * XQG)) < 512) gpw >>= 1;
    gpw = tante_opt("TANTE_XATTN_GPW", gpw);
    if (gpw < 1 || gpw > 64) TANTE_FAIL(-1, "tante_cross_attention: TANTE_XATTN_GPW = %d out of range", gpw);
    const int XQ_PER_WG = XWAVES * gpw * 16 * XQG;
    if ((Lq + XQ_PER_WG - 1) / XQ_PER_WG > 65535) TANTE_FAIL(-2, "tante_cross_attention: grid too large");
    const dim3 mgrid((unsigned)(n_batch * n_head), (unsigned)((Lq + XQ_PER_WG - 1) / XQ_PER_WG));
    hipLaunchKernelGGL(xattn_mfma_kernel, mgrid, dim3(XWAVES * 64), lds, s, (const unsigned short*)q, (const unsigned short*)k, (const unsigned short*)v,
                       (unsigned short*)o, n_head, Lq, Lk, Sp, (long)ldq, (long)ldkv, (long)ldo, scale * 1.44269504088896340736f, qbr, gpw);
    TANTE_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid((unsigned)(n_batch * n_head), (unsigned)((Lq + 255) / 256));
#define TANTE_XA(DD) hipLaunchKernelGGL(cross_attn_kernel<DD>, grid, dim3(256), 0, s, q, k, v, o, dtype, n_head, Lq, Lk, (long)ldq, (long)ldkv, (long)ldo, scale, qbr)
  switch (D) {
    case 4: TANTE_XA(4); break;
    case 8: TANTE_XA(8); break;
    case 12: TANTE_XA(12); break;
    case 16: TANTE_XA(16); break;
    case 32: TANTE_XA(32); break;
    case 64: TANTE_XA(64); break;
    default: TANTE_FAIL(-2, "tante_cross_attention: head dim %d not in {4, 8, 12, 16, 32, 64}", D);
  }
#undef TANTE_XA
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_cross_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* dO, void* dq, float* dk,
                                         float* dv, float* stats, int dtype, int64_t n_batch, int n_head, int D, int Lq, int Lk, int64_t ldq,
                                         int64_t ldkv, int64_t ldo, int64_t ldg, void* stream) {
  if (!q || !k || !v || !o || !dO || !dq || !dk || !dv || !stats || n_batch <= 0 || n_head <= 0 || Lq <= 0 || Lk <= 0)
    TANTE_FAIL(-1, "tante_cross_attention_bwd: bad argument");
  if ((Lq + 255) / 256 > 65535 || (Lk + 255) / 256 > 65535 || (Lq + XB_QC - 1) / XB_QC > 65535) TANTE_FAIL(-2, "tante_cross_attention_bwd: grid too large");
  const float scale = 1.0f / sqrtf((float)D);
  hipStream_t s = (hipStream_t)stream;
  const dim3 ga((unsigned)(n_batch * n_head), (unsigned)((Lq + 255) / 256));
  const dim3 gb((unsigned)(n_batch * n_head), (unsigned)((Lk + 255) / 256), (unsigned)((Lq + XB_QC - 1) / XB_QC));
#define TANTE_XB(DD)                                                                                                                         \
  hipLaunchKernelGGL(xattn_bwd_dq_kernel<DD>, ga, dim3(256), 0, s, q, k, v, o, dO, dq, stats, dtype, n_head, Lq, Lk, (long)ldq, (long)ldkv,   \
                     (long)ldo, scale);                                                                                                       \
  hipLaunchKernelGGL(xattn_bwd_dkv_kernel<DD>, gb, dim3(256), 0, s, q, k, v, dO, stats, dk, dv, dtype, n_head, Lq, Lk, (long)ldq, (long)ldkv, \
                     (long)ldo, (long)ldg, scale)
  switch (D) {
    case 4: TANTE_XB(4); break;
    case 8: TANTE_XB(8); break;
    case 12: TANTE_XB(12); break;
    case 16: TANTE_XB(16); break;
    case 32: TANTE_XB(32); break;
    case 64: TANTE_XB(64); break;
    default: TANTE_FAIL(-2, "tante_cross_attention_bwd: head dim %d not in {4, 8, 12, 16, 32, 64}", D);
  }
#undef TANTE_XB
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_layernorm_affine_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, int64_t M, int C,
                                          float eps, float* dx, float* dgamma, float* dbeta, void* stream) {
  if (!dy || !x || !dx || M <= 0 || C <= 0) TANTE_FAIL(-1, "tante_layernorm_affine_bwd: bad argument");
  if (C > 1024) TANTE_FAIL(-2, "tante_layernorm_affine_bwd: C = %d > 1024", C);
  hipLaunchKernelGGL(ln_affine_bwd_kernel, dim3((unsigned)((M + 4 * LNB_RPW - 1) / (4 * LNB_RPW))), dim3(256), 0, (hipStream_t)stream, dy, dy_dtype,
                     x, x_dtype, gamma, (long)M, C, eps, dx, dgamma, dbeta);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_grid_embed_bwd(const float* coords, const float* grid, const float* latents, const float* out, const float* dout,
                                    int64_t N, int G, int LD, float eps, float* wsum_work, float* dlatents, float* dgrid, void* stream) {
  if (!coords || !grid || !latents || !out || !dout || !wsum_work || !dlatents || !dgrid || N <= 0 || G <= 0 || LD <= 0)
    TANTE_FAIL(-1, "tante_grid_embed_bwd: bad argument");
  if (LD > 1024 || N > 2147483647L) TANTE_FAIL(-2, "tante_grid_embed_bwd: latent_dim %d > 1024 or too many queries", LD);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(grid_embed_wsum_kernel, dim3((unsigned)N), dim3(256), 0, s, coords, grid, (long)N, G, eps, wsum_work);
  hipLaunchKernelGGL(grid_embed_bwd_kernel, dim3((unsigned)G), dim3(256), 0, s, coords, grid, latents, out, dout, wsum_work, (long)N, G, LD, eps,
                     dlatents, dgrid);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_grid_embed(const float* coords, const float* grid, const float* latents, int64_t N, int G, int LD, float eps, float* out,
                                void* stream) {
  if (!coords || !grid || !latents || !out || N <= 0 || G <= 0 || LD <= 0) TANTE_FAIL(-1, "tante_grid_embed: bad argument");
  if (LD > 1024) TANTE_FAIL(-2, "tante_grid_embed: latent_dim %d > 1024", LD);
  hipLaunchKernelGGL(grid_embed_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, coords, grid, latents, (long)N, G, LD, eps, out);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_fourier_embed(const float* coords, const float* kernel, int64_t N, int E, float* out, void* stream) {
  if (!coords || !kernel || !out || N <= 0 || E <= 0 || E % 2) TANTE_FAIL(-1, "tante_fourier_embed: bad argument");
  hipLaunchKernelGGL(fourier_embed_kernel, dim3(grid_for((long)N * (E / 2))), dim3(256), 0, (hipStream_t)stream, coords, kernel, (long)N, E / 2, out);
  TANTE_CHECK_LAUNCH();
  return 0;
}
