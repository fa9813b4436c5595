// Backward kernels of the TANTE train step (everything that is not a GEMM; the data-gradient GEMMs are tante_gemm with
// the *_T packings, the weight-gradient GEMM is wgrad.hip).
#include "common.hip.h"
#include "fused_common.hip.h"

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

// four consecutive elements at once (16 bytes of fp32, 8 of bf16): the streaming elementwise kernels of the train path move
// 25-50 MB per launch and were issuing one 2- or 4-byte access per thread
__device__ __forceinline__ f32x4 ld4(const void* p, int dtype, long i) {
  if (dtype == TANTE_BF16) {
    const u32x2 u = *(const u32x2*)((const unsigned short*)p + i);
    return f32x4{bf16_lo(u[0]), bf16_hi(u[0]), bf16_lo(u[1]), bf16_hi(u[1])};
  }
  return *(const f32x4*)((const float*)p + i);
}
__device__ __forceinline__ void st4(void* p, int dtype, long i, const f32x4& v) {
  if (dtype == TANTE_BF16) {
    u32x2 u;
    u[0] = pack_bf16x2(v[0], v[1]);
    u[1] = pack_bf16x2(v[2], v[3]);
    *(u32x2*)((unsigned short*)p + i) = u;
  } else {
    *(f32x4*)((float*)p + i) = v;
  }
}

// ---- LayerNorm without affine (gamma / beta are folded into the consumer's weight by the host) -------------------
// one wave per row; stats[row] = {mean, rstd}
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long M, int C, float eps, void* __restrict__ xhat,
                                                     int out_dtype, float* __restrict__ stats) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += xr[c];
  const float mean = wave_sum(s) / C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
  const float rstd = rsqrtf(wave_sum(q) / C + eps);
  for (int c = lane; c < C; c += 64) stx(xhat, out_dtype, row * C + c, (xr[c] - mean) * rstd);
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// C = 256 NV: a lane owns NV groups of 4 consecutive channels; the row lives in registers (one 16-byte load per group, one pass)
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(const float* __restrict__ x, long M, float eps, void* __restrict__ xhat, int out_dtype,
                                                         float* __restrict__ stats) {
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
    const long o = row * C + (i * 64 + lane) * 4;
    const f32x4 h = (v[i] - mean) * rstd;
    if (out_dtype == TANTE_BF16) {
      u32x2 u;
      u[0] = pack_bf16x2(h[0], h[1]);
      u[1] = pack_bf16x2(h[2], h[3]);
      *(u32x2*)((unsigned short*)xhat + o) = u;
    } else {
      *(f32x4*)((float*)xhat + o) = h;
    }
  }
  if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(const void* __restrict__ g, int g_dtype, const float* __restrict__ x,
                                                         const float* __restrict__ stats, const float* __restrict__ dskip, long M,
                                                         float* __restrict__ dx) {
  constexpr int C = 256 * NV;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  f32x4 gv[NV], xh[NV];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const long o = row * C + (i * 64 + lane) * 4;
    if (g_dtype == TANTE_BF16) {
      const u32x2 u = *(const u32x2*)((const unsigned short*)g + o);
      gv[i] = f32x4{bf16_lo(u[0]), bf16_hi(u[0]), bf16_lo(u[1]), bf16_hi(u[1])};
    } else {
      gv[i] = *(const f32x4*)((const float*)g + o);
    }
    xh[i] = (*(const f32x4*)(x + o) - mean) * rstd;
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1 += gv[i][j]; s2 += gv[i][j] * xh[i][j]; }
  }
  s1 = wave_sum(s1) / C; s2 = wave_sum(s2) / C;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const long o = row * C + (i * 64 + lane) * 4;
    f32x4 d = (gv[i] - s1 - xh[i] * s2) * rstd;
    if (dskip) d = d + *(const f32x4*)(dskip + o);
    *(f32x4*)(dx + o) = d;
  }
}

// dx = dskip + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = d xhat
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ g, int g_dtype, const float* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ dskip, long M, int C,
                                                     float* __restrict__ dx) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float mean = stats[2 * row], rstd = stats[2 * row + 1];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float gv = ldx(g, g_dtype, row * C + c), xh = (x[row * C + c] - mean) * rstd;
    s1 += gv; s2 += gv * xh;
  }
  s1 = wave_sum(s1) / C; s2 = wave_sum(s2) / C;
  for (int c = lane; c < C; c += 64) {
    const float gv = ldx(g, g_dtype, row * C + c), xh = (x[row * C + c] - mean) * rstd;
    const float d = rstd * (gv - s1 - xh * s2);
    dx[row * C + c] = (dskip ? dskip[row * C + c] : 0.f) + d;
  }
}

// ---- activations ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_f(float x, int act) { return apply_act(x, act); }
__global__ void act_fwd_kernel(const void* __restrict__ pre, int in_dtype, void* __restrict__ post, int out_dtype, long n, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stx(post, out_dtype, i, act_f(ldx(pre, in_dtype, i), act));
}
__global__ void act_fwd_vec_kernel(const void* __restrict__ pre, int in_dtype, void* __restrict__ post, int out_dtype, long n4, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = ld4(pre, in_dtype, 4 * i);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = act_f(v[j], act);
  st4(post, out_dtype, 4 * i, v);
}
__global__ void act_bwd_vec_kernel(const void* __restrict__ dpost, int d_dtype, const void* __restrict__ pre, int pre_dtype,
                                   void* __restrict__ dpre, int out_dtype, long n4, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 g = ld4(dpost, d_dtype, 4 * i);
  const f32x4 x = ld4(pre, pre_dtype, 4 * i);
#pragma unroll
  for (int j = 0; j < 4; ++j) g[j] *= act_df(x[j], act);
  st4(dpre, out_dtype, 4 * i, g);
}
__global__ void act_bwd_kernel(const void* __restrict__ dpost, int d_dtype, const void* __restrict__ pre, int pre_dtype,
                               void* __restrict__ dpre, int out_dtype, long n, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stx(dpre, out_dtype, i, ldx(dpost, d_dtype, i) * act_df(ldx(pre, pre_dtype, i), act));
}

// ---- out[c] = sum over (o, i) of x[(o*C + c)*inner + i]  (bias gradients) -------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x, int dtype, long outer, int C, long inner, long chunk,
                                                     float* __restrict__ out) {
  // grid.x: chunks of the (outer x inner) reduction, grid.y: channel groups of 64 (inner == 1) or single channels
  if (inner == 1) {
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C) {
      const long o0 = (long)blockIdx.x * chunk, o1 = min(outer, o0 + chunk);
      for (long o = o0 + sub; o < o1; o += 4) s += ldx(x, dtype, o * C + c);
    }
    __shared__ float part[256];
    part[threadIdx.x] = s;
    __syncthreads();
    if (sub == 0 && c < C) atomicAdd(&out[c], part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192]);
  } else {
    const int c = blockIdx.y;
    const long nch = (inner + chunk - 1) / chunk;           // `chunk` contiguous elements of one (o, c) slab per workgroup
    const long o = blockIdx.x / nch, i0 = (blockIdx.x % nch) * chunk, i1 = min(inner, i0 + chunk);
    float s = 0.f;
    const long base = (o * C + c) * inner;
    for (long i = i0 + threadIdx.x; i < i1; i += 256) s += ldx(x, dtype, base + i);
    s = wave_sum(s);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[c], part[0] + part[1] + part[2] + part[3]);
  }
}

// dense bf16 rows (inner == 1, C a multiple of 8 with 256 % (C / 8) == 0): 16 bytes = 8 channels per lane and row instead of one 2-byte
// element (the scalar kernel above read the deconvolution stages' bias gradients at 2.2 TB/s)
__global__ __launch_bounds__(256) void colsum_bf16_vec_kernel(const unsigned short* __restrict__ x, long outer, int C, long chunk, float* __restrict__ out) {
  __shared__ float acc[2048];
  const int gpr = C >> 3, rpp = 256 / gpr, cg = threadIdx.x % gpr, rsub = threadIdx.x / gpr;
  for (int i = threadIdx.x; i < C; i += 256) acc[i] = 0.f;
  __syncthreads();
  const long o0 = (long)blockIdx.x * chunk, o1 = min(outer, o0 + chunk);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  long o = o0 + rsub;
  for (; o + 3L * rpp < o1; o += 4L * rpp) {      // four rows in flight per lane: one load per trip left the pass latency-bound (0.7 TB/s)
    const u32x4 v0 = *(const u32x4*)(x + o * C + 8 * cg), v1 = *(const u32x4*)(x + (o + rpp) * C + 8 * cg);
    const u32x4 v2 = *(const u32x4*)(x + (o + 2L * rpp) * C + 8 * cg), v3 = *(const u32x4*)(x + (o + 3L * rpp) * C + 8 * cg);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s[2 * q] += (bf16_lo(v0[q]) + bf16_lo(v1[q])) + (bf16_lo(v2[q]) + bf16_lo(v3[q]));
      s[2 * q + 1] += (bf16_hi(v0[q]) + bf16_hi(v1[q])) + (bf16_hi(v2[q]) + bf16_hi(v3[q]));
    }
  }
  for (; o < o1; o += rpp) {
    const u32x4 v = *(const u32x4*)(x + o * C + 8 * cg);
#pragma unroll
    for (int q = 0; q < 4; ++q) { s[2 * q] += bf16_lo(v[q]); s[2 * q + 1] += bf16_hi(v[q]); }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) atomicAdd(&acc[8 * cg + e], s[e]);
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) atomicAdd(&out[i], acc[i]);
}

// ---- FiLM + positional epilogue as a stand-alone op (train path): y = v * a[t] + b[t] + s[hw], rows r = (b, t, hw) --
__global__ void film_pos_fwd_kernel(const float* __restrict__ v, const float* __restrict__ a, const float* __restrict__ b,
                                    const float* __restrict__ s, long rows, int C4, int T, long HW, float* __restrict__ y) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * C4) return;
  const long r = idx / C4;
  const int c4 = (int)(idx - r * C4);
  const long hw = r % HW;
  const int t = (int)((r / HW) % T);
  const f32x4 vv = ((const f32x4*)v)[idx];
  ((f32x4*)y)[idx] = vv * ((const f32x4*)a)[t * C4 + c4] + ((const f32x4*)b)[t * C4 + c4] + ((const f32x4*)s)[hw * C4 + c4];
}
// CViT's encoder: y[(b, hw), t] = v[(b, t), hw] + t_emb[t] + s_emb[hw]: the positional sums and the 'b t s d -> (b s) t d' regrouping that the
// time-aggregation attention reads, in one pass (one output row per thread quad group; reads stride HW rows, writes are consecutive)
__global__ void pos_embed_tmajor_kernel(const float* __restrict__ v, const float* __restrict__ te, const float* __restrict__ se, long rows,
                                        int C4, int T, long HW, float* __restrict__ y) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // output element: row (b, hw, t)
  if (idx >= rows * C4) return;
  const long r = idx / C4;
  const int c4 = (int)(idx - r * C4);
  const int t = (int)(r % T);
  const long bh = r / T, hw = bh % HW, b = bh / HW;
  const f32x4 vv = ((const f32x4*)v)[((b * T + t) * HW + hw) * C4 + c4];
  ((f32x4*)y)[idx] = vv + ((const f32x4*)te)[t * C4 + c4] + ((const f32x4*)se)[hw * C4 + c4];
}
// the same with the window given as T separate frame tensors (frame t of item b at f[t] + b * bstride[t], rows (hw, c)): the BPTT rollout
// keeps one encoding per frame and a window is any T of them -- stacking them first was a 25 MB copy per call
struct FilmFrames {
  const float* f[8];
  long bstride[8];
  float* dv[8];        // backward: the frames' gradients, contiguous (B, HW, C) each
  int acc = 0;         // backward: bit t -- dv[t] is added to, not written
};
__global__ void film_pos_fwd_frames_kernel(FilmFrames F, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ s,
                                           long rows, int C4, int T, long HW, float* __restrict__ y) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * C4) return;
  const long r = idx / C4;
  const int c4 = (int)(idx - r * C4);
  const long hw = r % HW, bt = r / HW;
  const int t = (int)(bt % T);
  const long bi = bt / T;
  const f32x4 vv = *(const f32x4*)(F.f[t] + bi * F.bstride[t] + (hw * C4 + c4) * 4);
  ((f32x4*)y)[idx] = vv * ((const f32x4*)a)[t * C4 + c4] + ((const f32x4*)b)[t * C4 + c4] + ((const f32x4*)s)[hw * C4 + c4];
}
// dv = dy * a[t];  da[t][c] += sum dy * v;  db[t][c] += sum dy   (one workgroup per (bt, hw chunk); thread = channel)
__global__ __launch_bounds__(256) void film_pos_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ v,
                                                           const float* __restrict__ a, long HW, int C, int T, long chunk,
                                                           float* __restrict__ dv, float* __restrict__ da, float* __restrict__ db) {
  const long bt = blockIdx.y;
  const int t = (int)(bt % T);
  const long h0 = (long)blockIdx.x * chunk, h1 = min(HW, h0 + chunk);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float av = a[t * C + c];
    float sa = 0.f, sb = 0.f;
    for (long hw = h0; hw < h1; ++hw) {
      const long i = (bt * HW + hw) * C + c;
      const float g = dy[i];
      dv[i] = g * av;
      sa += g * v[i];
      sb += g;
    }
    atomicAdd(&da[t * C + c], sa);
    atomicAdd(&db[t * C + c], sb);
  }
}
// the same for C = 256: a thread owns four channels (float4 streams: a row is one 1 KiB access of 64 lanes), four rows in flight per
// workgroup and eight per thread -- the kernel above reads 4 bytes per lane and 64 rows one after the other (2 TB/s by the counters)
template <bool FRAMES, int ROWS>
__global__ __launch_bounds__(256) void film_pos_bwd256_kernel(const float* __restrict__ dy, const float* __restrict__ v, const float* __restrict__ a,
                                                              long HW, int T, float* __restrict__ dv, float* __restrict__ da,
                                                              float* __restrict__ db, FilmFrames F) {
  constexpr int C4 = 64;
  __shared__ f32x4 red[2][3][C4];
  const long bt = blockIdx.y;
  const int t = (int)(bt % T), r = threadIdx.x >> 6, c4 = threadIdx.x & 63;
  const f32x4* vsrc = FRAMES ? (const f32x4*)(F.f[t] + (bt / T) * F.bstride[t]) - bt * HW * C4 : (const f32x4*)v;      // indexed by (bt HW + hw) C4 + c4 below
  f32x4* vdst = FRAMES ? (f32x4*)F.dv[t] + ((bt / T) * HW - bt * HW) * C4 : (f32x4*)dv;
  const long h0 = (long)blockIdx.x * ROWS;
  const f32x4 av = ((const f32x4*)a)[t * C4 + c4];
  f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, sb = sa;
  f32x4 g[ROWS / 4], x[ROWS / 4];
#pragma unroll
  for (int it = 0; it < ROWS / 4; ++it) {
    const long hw = h0 + it * 4 + r;
    const bool in = hw < HW;
    g[it] = in ? ((const f32x4*)dy)[(bt * HW + hw) * C4 + c4] : f32x4{0.f, 0.f, 0.f, 0.f};
    x[it] = in ? vsrc[(bt * HW + hw) * C4 + c4] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int it = 0; it < ROWS / 4; ++it) {
    const long hw = h0 + it * 4 + r;
    if (hw < HW) {
      f32x4 o = g[it] * av;
      if (FRAMES && ((F.acc >> t) & 1)) o += vdst[(bt * HW + hw) * C4 + c4];      // a frame that sits in several windows: the later uses add
      vdst[(bt * HW + hw) * C4 + c4] = o;
    }
    sa += g[it] * x[it];
    sb += g[it];
  }
  if (r) { red[0][r - 1][c4] = sa; red[1][r - 1][c4] = sb; }
  __syncthreads();
  if (r == 0) {
    sa += (red[0][0][c4] + red[0][1][c4]) + red[0][2][c4];
    sb += (red[1][0][c4] + red[1][1][c4]) + red[1][2][c4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      atomicAdd(&da[t * 256 + 4 * c4 + e], sa[e]);
      atomicAdd(&db[t * 256 + 4 * c4 + e], sb[e]);
    }
  }
}
// ds[hw][c] = sum over bt of dy[(bt*HW + hw)*C + c]
__global__ void film_pos_ds_kernel(const float* __restrict__ dy, long BT, long HW, int C, float* __restrict__ ds, int accumulate = 0) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= HW * C) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four images in flight per thread
  long bt = 0;
  for (; bt + 4 <= BT; bt += 4) {
    const float a = dy[bt * HW * C + idx], b = dy[(bt + 1) * HW * C + idx], c = dy[(bt + 2) * HW * C + idx], d = dy[(bt + 3) * HW * C + idx];
    s0 += a; s1 += b; s2 += c; s3 += d;
  }
  for (; bt < BT; ++bt) s0 += dy[bt * HW * C + idx];
  ds[idx] = (accumulate ? ds[idx] : 0.0f) + ((s0 + s1) + (s2 + s3));
}

// ---- Taylor sum backward: dd_k = sum_i c_ik dout_i;  dlast (+)= sum_i dout_i -----------------------------------------
struct TaylorBArgs {
  float* dd[8];
  float coef[8][8];
};
template <int NO>
__global__ void taylor_bwd_kernel(const float* __restrict__ dout, long dout_bstride, TaylorBArgs ta, int n_out, float* __restrict__ dlast,
                                  long dlast_bstride, int accumulate, long B, long frame4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * frame4) return;
  const long b = idx / frame4, f = idx - b * frame4;
  f32x4 acc[NO], sl = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NO; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < n_out) {
      const f32x4 g = *(const f32x4*)(dout + b * dout_bstride + ((long)i * frame4 + f) * 4);
      sl += g;
#pragma unroll
      for (int k = 0; k < NO; ++k) acc[k] += g * ta.coef[i][k];
    }
#pragma unroll
  for (int k = 0; k < NO; ++k) ((f32x4*)ta.dd[k])[idx] = acc[k];
  f32x4* dl = (f32x4*)(dlast + b * dlast_bstride + 4 * f);
  *dl = accumulate ? (*dl + sl) : sl;
}

// ---- attention backward (small sequences: L <= 128): one lane per token, two phases ----------------------------------
// phase A (lane = query i): softmax stats m_i, l_i over its keys, delta_i = do_i . o_i, dq_i
// phase B (lane = key j):   dk_j = scale * sum_i ds_ij q_i,  dv_j = sum_i p_ij do_i,  ds_ij = p_ij (do_i . v_j - delta_i)
// LDS rows hold the operands in their own dtype (bf16 rows halve the footprint: 4 workgroups per CU instead of 2)
template <bool LB16> struct AbwT { using T = float; };
template <> struct AbwT<true> { using T = unsigned short; };
template <bool LB16>
__device__ __forceinline__ f32x4 abw_ld4(const typename AbwT<LB16>::T* p) {
  if constexpr (LB16) {
    const u32x2 u = *(const u32x2*)p;
    return f32x4{bf16_lo(u[0]), bf16_hi(u[0]), bf16_lo(u[1]), bf16_hi(u[1])};
  } else {
    return *(const f32x4*)p;
  }
}
template <int D, bool LB16>
__global__ __launch_bounds__(128) void attn_bwd_small_kernel(const void* __restrict__ qkv, const void* __restrict__ dO, void* __restrict__ dqkv,
                                                             int dtype, int C, TanteSeq sq, int G, int causal, float scale, float p_drop,
                                                             unsigned long long seed) {
  using ET = typename AbwT<LB16>::T;
  constexpr int ST = D + (LB16 ? 8 : 4);   // row stride in elements: a multiple of 16 bytes
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];  // m, l, delta [128] floats, then Q, K, V, dO rows [128][ST]
  float* Ms = (float*)sm_raw;
  float* Ls = Ms + 128;
  float* Ds = Ls + 128;
  ET* Qs = (ET*)(Ds + 128);
  ET* Ks = Qs + 128 * ST;
  ET* Vs = Ks + 128 * ST;
  ET* Gs = Vs + 128 * ST;
  const int tid = threadIdx.x, h = blockIdx.y, L = sq.L;
  const int g = tid / L, l = tid - g * L, s = blockIdx.x * G + g;
  const bool live = (g < G) && (s < sq.nseq);
  auto token_of = [&](int t, bool& ok) -> long {   // token index of slot t of this workgroup (slots = G sequences x L positions)
    const int gg = t / L, ll = t - gg * L, ss = blockIdx.x * G + gg;
    ok = (gg < G) && (ss < sq.nseq);
    return ok ? (long)(ss / sq.n_s0) * sq.S1 + (long)(ss % sq.n_s0) * sq.S0 + (long)(ll / sq.n_l0) * sq.P1 + (long)(ll % sq.n_l0) * sq.P0 : 0;
  };
  bool dummy;
  const long tok = token_of(tid, dummy);
  // cooperative, vectorised staging: 16-byte pieces of the q / k / v / dO rows (a lane-per-token copy issues 4 D scalar loads
  // 3C elements apart -- that, not the arithmetic, was this kernel's time)
  {
    constexpr int EPC = 8;                         // elements per piece (bf16: 16 bytes; fp32: two 16-byte loads)
    constexpr int PPR = D / EPC > 0 ? D / EPC : 1; // pieces per row
    const int nslot = G * L;
    for (int idx = tid; idx < nslot * 4 * PPR; idx += 128) {
      const int pc = idx % PPR, mat = (idx / PPR) & 3, t = idx / (4 * PPR);
      bool ok;
      const long tk = token_of(t, ok);
      ET* dst = (mat == 0 ? Qs : mat == 1 ? Ks : mat == 2 ? Vs : Gs) + t * ST + pc * EPC;
      if (!ok) continue;
      const long e = (mat < 3 ? tk * 3L * C + (long)mat * C : tk * (long)C) + (long)h * D + pc * EPC;
      const void* src = mat < 3 ? qkv : dO;
      if constexpr (D >= EPC) {
        if constexpr (LB16) {
          *(u32x4*)dst = *(const u32x4*)((const unsigned short*)src + e);
        } else {
          const f32x4 f0 = *(const f32x4*)((const float*)src + e), f1 = *(const f32x4*)((const float*)src + e + 4);
          *(f32x4*)dst = f0;
          *(f32x4*)(dst + 4) = f1;
        }
      } else {
        for (int i = 0; i < D; ++i) {
          const float v = ldx(src, dtype, e + i);
          if constexpr (LB16) { __bf16 bb = (__bf16)v; dst[i] = __builtin_bit_cast(unsigned short, bb); }
          else dst[i] = v;
        }
      }
    }
  }
  __syncthreads();
  const int r0 = g * L;
  constexpr int D4 = D / 4;
  // LDS rows are read as float4 (ST * 4 bytes is a multiple of 16) and a lane's own rows live in registers: an FMA costs a
  // quarter of an LDS instruction instead of two.
  auto dot = [&](const f32x4 (&a)[D4], const ET* row) {
    float sacc = 0.f;
#pragma unroll
    for (int c = 0; c < D4; ++c) {
      const f32x4 r = abw_ld4<LB16>(row + 4 * c);
      sacc += a[c][0] * r[0] + a[c][1] * r[1] + a[c][2] * r[2] + a[c][3] * r[3];
    }
    return sacc;
  };
  float dq[D];
#pragma unroll
  for (int i = 0; i < D; ++i) dq[i] = 0.f;
  if (live) {
    // ---- phase A (lane = query): softmax statistics, then ONE pass that accumulates
    //   delta = sum_j p_j dp'_j,  a1 = sum_j p_j dp'_j k_j,  a2 = sum_j p_j k_j   ->   dq = scale (a1 - delta a2)
    // (dp'_j = do . v_j through the dropout mask, p_j the un-dropped probability)
    f32x4 qv[D4], gv[D4];
#pragma unroll
    for (int c = 0; c < D4; ++c) { qv[c] = abw_ld4<LB16>(Qs + tid * ST + 4 * c); gv[c] = abw_ld4<LB16>(Gs + tid * ST + 4 * c); }
    const int nk = causal ? l + 1 : L;
    float m = -INFINITY, lsum = 0.f;
    for (int j = 0; j < nk; ++j) {
      const float sc = dot(qv, Ks + (r0 + j) * ST) * scale;
      const float mn = fmaxf(m, sc);
      lsum = lsum * expf(m - mn) + expf(sc - mn);
      m = mn;
    }
    const float kscale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const unsigned long long mrow = (((unsigned long long)s * gridDim.y + h) * L + l) * L;   // this query's row of the dropout mask
    const float inv_l = 1.0f / lsum;
    float delta = 0.f;
    f32x4 a1[D4], a2[D4];
#pragma unroll
    for (int c = 0; c < D4; ++c) a1[c] = a2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nk; ++j) {
      const ET* kr = Ks + (r0 + j) * ST;
      const float sc = dot(qv, kr) * scale;
      float dp = dot(gv, Vs + (r0 + j) * ST);
      if (p_drop > 0.f) dp = dropout_keep(seed, mrow + j, p_drop) ? dp * kscale : 0.f;   // d(dropped prob) -> d(prob)
      const float p = expf(sc - m) * inv_l;
      const float pdp = p * dp;
      delta += pdp;
#pragma unroll
      for (int c = 0; c < D4; ++c) {
        const f32x4 kv4 = abw_ld4<LB16>(kr + 4 * c);
        a1[c] += kv4 * pdp;
        a2[c] += kv4 * p;
      }
    }
    Ms[tid] = m; Ls[tid] = inv_l; Ds[tid] = delta;
#pragma unroll
    for (int c = 0; c < D4; ++c)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) dq[4 * c + e4] = scale * (a1[c][e4] - delta * a2[c][e4]);
  }
  __syncthreads();
  // ---- phase B: this lane is key l of its sequence; queries i that see it: i >= l (causal) or all ----
  float dk[D], dv[D];
  {
    f32x4 kv[D4], vv[D4], dk4[D4], dv4[D4];
#pragma unroll
    for (int c = 0; c < D4; ++c) {
      kv[c] = abw_ld4<LB16>(Ks + tid * ST + 4 * c);
      vv[c] = abw_ld4<LB16>(Vs + tid * ST + 4 * c);
      dk4[c] = dv4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float ks2 = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int i = causal ? l : 0; live && i < L; ++i) {
      const int qi = r0 + i;
      const ET* qr = Qs + qi * ST;
      const ET* gr = Gs + qi * ST;
      f32x4 q4[D4], g4[D4];
      float sc = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < D4; ++c) {
        q4[c] = abw_ld4<LB16>(qr + 4 * c);
        g4[c] = abw_ld4<LB16>(gr + 4 * c);
        sc += q4[c][0] * kv[c][0] + q4[c][1] * kv[c][1] + q4[c][2] * kv[c][2] + q4[c][3] * kv[c][3];
        dp += g4[c][0] * vv[c][0] + g4[c][1] * vv[c][1] + g4[c][2] * vv[c][2] + g4[c][3] * vv[c][3];
      }
      const float p = expf(sc * scale - Ms[qi]) * Ls[qi];
      float pd = p;
      if (p_drop > 0.f) {
        const bool keep = dropout_keep(seed, (((unsigned long long)s * gridDim.y + h) * L + i) * L + l, p_drop);
        pd = keep ? p * ks2 : 0.f;
        dp = keep ? dp * ks2 : 0.f;
      }
      const float ds = p * (dp - Ds[qi]) * scale;
#pragma unroll
      for (int c = 0; c < D4; ++c) {
        dk4[c] += q4[c] * ds;
        dv4[c] += g4[c] * pd;
      }
    }
#pragma unroll
    for (int c = 0; c < D4; ++c)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) { dk[4 * c + e4] = dk4[c][e4]; dv[4 * c + e4] = dv4[c][e4]; }
  }
  // results -> LDS (the operand rows are dead once every lane has left phase B), then the same cooperative 16-byte pieces out
  __syncthreads();
#pragma unroll
  for (int i = 0; i < D; ++i) {
    if constexpr (LB16) {
      __bf16 a0 = (__bf16)dq[i], a1 = (__bf16)dk[i], a2 = (__bf16)dv[i];
      Qs[tid * ST + i] = __builtin_bit_cast(unsigned short, a0);
      Ks[tid * ST + i] = __builtin_bit_cast(unsigned short, a1);
      Vs[tid * ST + i] = __builtin_bit_cast(unsigned short, a2);
    } else {
      Qs[tid * ST + i] = dq[i]; Ks[tid * ST + i] = dk[i]; Vs[tid * ST + i] = dv[i];
    }
  }
  __syncthreads();
  {
    constexpr int EPC = 8;
    constexpr int PPR = D / EPC > 0 ? D / EPC : 1;
    const int nslot = G * L;
    for (int idx = tid; idx < nslot * 3 * PPR; idx += 128) {
      const int pc = idx % PPR, mat = (idx / PPR) % 3, t = idx / (3 * PPR);
      bool ok;
      const long tk = token_of(t, ok);
      if (!ok) continue;
      const ET* srcr = (mat == 0 ? Qs : mat == 1 ? Ks : Vs) + t * ST + pc * EPC;
      const long e = tk * 3L * C + (long)mat * C + (long)h * D + pc * EPC;
      if constexpr (D >= EPC) {
        if constexpr (LB16) {
          *(u32x4*)((unsigned short*)dqkv + e) = *(const u32x4*)srcr;
        } else {
          *(f32x4*)((float*)dqkv + e) = *(const f32x4*)srcr;
          *(f32x4*)((float*)dqkv + e + 4) = *(const f32x4*)(srcr + 4);
        }
      } else {
        for (int i = 0; i < D; ++i) {
          if constexpr (LB16) ((unsigned short*)dqkv)[e + i] = srcr[i];
          else stx(dqkv, dtype, e + i, srcr[i]);
        }
      }
    }
  }
}

// ---- attention backward on the matrix cores: bf16, head dim 32, L <= 64 ------------------------------------------------------------
// One wave per (unit, head); a unit is one sequence in NT = ceil(L / 16) tiles of 16 slots, or (L < 16) the 16 / L sequences that fit
// one tile, kept apart by a block-diagonal mask.  v_mfma_f32_16x16x32_bf16 computes X Y^T from two "row, 8 consecutive k" fragments,
// and head dim 32 is exactly one k step, so Q, K, V, dO fragments are plain 16-byte global loads and
//   S^T = K Q^T, dP^T = V dO^T   (accumulator: 4 keys x 1 query per lane)  -> row statistics m, 1/l, delta and dS^T -> dQ^T = K^T-frag x dS
//   S   = Q K^T, dP   = dO V^T   (accumulator: 4 queries x 1 key per lane) -> dS, dropped P                      -> dK^T, dV^T
// each score tile is computed in both orientations (2 MFMAs) instead of being transposed.  An accumulator pair (two 16-slot tiles) packs
// straight into the B operand of the second product with k order (tile a: 4 kk + r, tile b: 4 kk + r); the matching A operand - K^T, Q^T
// or dO^T in that k order - is two ds_read_b64_tr_b16 of the row-major LDS image (MI355X_MICROARCH.md "LDS": 4 rows x 16 columns per
// 16-lane group, column-major out).  The VALU kernel above spends ~260 FMAs per (query, key) pair; this one ~30 instructions.
__device__ __forceinline__ u32x2 abm_tr(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
template <int NT>
__global__ __launch_bounds__(256) void attn_bwd_mfma_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ dO,
                                                            unsigned short* __restrict__ dqkv, int C, int n_head, TanteSeq sq, int SPT, int causal,
                                                            float scale, float p_drop, unsigned long long seed,
                                                            const unsigned long long* __restrict__ seed_mix) {
  if (seed_mix) seed ^= *seed_mix;       // per-step word of a replayed train step (tante_set_seed_mix)
  constexpr int ROWS = NT * 16, NP = (NT + 1) / 2;
  // LDS rows are 96 bytes apart (64 of data): with 64-byte rows the transposing reads of the four 16-lane groups hit the same banks
  // (rows r and r + 4 are 256 bytes apart): PMC showed half of this kernel's LDS cycles as bank conflicts
  constexpr int RS = 48;   // row stride in bf16 elements
  constexpr int WAVE_LDS = 3 * ROWS * RS * 2 + 16 * RS * 2 + 3 * ROWS * 4 + NT * NT * 64;   // Q, K, dO images, one 16-row staging tile, the row statistics, the keep bits
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15, qq = l15 >> 2, pp = l15 & 3;
  const int h = blockIdx.y * 4 + wave;
  if (h >= n_head) return;                       // waves never meet at a barrier: each owns its slice of LDS
  char* base = sm_raw + wave * WAVE_LDS;
  unsigned short* Qs = (unsigned short*)base;
  unsigned short* Ks = Qs + ROWS * RS;
  unsigned short* Gs = Ks + ROWS * RS;
  unsigned short* Os = Gs + ROWS * RS;           // 16-row staging tile: V on the way in (it needs no image), the results on the way out
  float* St = (float*)(Os + 16 * RS);            // m [ROWS], 1/l [ROWS], delta [ROWS]
  // keep bits of the dropout mask, one byte per (score tile, key group kk, query): pass 1 draws them four keys at a time (one hash per
  // aligned run of mask indices) and pass 2, whose lanes hold four QUERIES of one key -- mask indices L apart, a hash and a 64-bit index
  // each: half of that pass's instructions -- reads them back transposed: a 32-bit word = the bytes of its four queries
  unsigned char* Mb = (unsigned char*)(St + 3 * ROWS);
  const int L = sq.L, unit = blockIdx.x;
  const bool big = L >= 16;
  const unsigned rcpL = (65536u + L - 1) / L;    // floor(slot / L) for slot < 16
  auto decode = [&](int slot, int& sl, int& pos, bool& live) {   // slot -> (sequence of the tile, position, exists)
    sl = big ? 0 : (int)(((unsigned)slot * rcpL) >> 16);
    pos = slot - sl * L;
    const int seq = big ? unit : unit * SPT + sl;
    live = (big ? pos < L : sl < SPT) && seq < sq.nseq;
  };
  long ctok[NT];
  int csl[NT], cpos[NT];
  bool clive[NT];
  u32x4 qf[NT], kf[NT], vf[NT], gf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    decode(t * 16 + l15, csl[t], cpos[t], clive[t]);
    const int seq = big ? unit : unit * SPT + csl[t];
    ctok[t] = clive[t] ? (long)(seq / sq.n_s0) * sq.S1 + (long)(seq % sq.n_s0) * sq.S0 + (long)(cpos[t] / sq.n_l0) * sq.P1 +
                             (long)(cpos[t] % sq.n_l0) * sq.P0 : 0;
  }
  // Memory is touched in ROW form: lanes 4 r .. 4 r + 3 = the 64 bytes of (token row r, this head) -- in operand layout (lane & 15 = row)
  // neighbouring lanes are different rows and the memory pipe works the instruction off lane by lane (tools/ubench/ta_cost.hip: 63
  // clocks against 17); with 4 loads + 6 eight-byte stores per 16 tokens that pipe, not HBM, set this kernel's time.  The rows go
  // through the LDS images (which the transposing reads need anyway) and come back as operand fragments.
  const int lr = lane >> 2, lc = lane & 3;
  long rtok[NT];
  bool rlive[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    rtok[t] = ((long)__shfl((int)(ctok[t] >> 32), lr) << 32) | (unsigned)__shfl((int)ctok[t], lr);
    rlive[t] = __shfl((int)clive[t], lr) != 0;
    u32x4 q4 = u32x4{0u, 0u, 0u, 0u}, k4 = q4, v4 = q4, g4 = q4;
    if (rlive[t]) {
      const unsigned short* r = qkv + rtok[t] * 3L * C + h * 32 + lc * 8;
      q4 = *(const u32x4*)r;
      k4 = *(const u32x4*)(r + C);
      v4 = *(const u32x4*)(r + 2 * C);
      g4 = *(const u32x4*)(dO + rtok[t] * (long)C + h * 32 + lc * 8);
    }
    *(u32x4*)(Qs + (t * 16 + lr) * RS + lc * 8) = q4;     // row-major images (64 data bytes per 96-byte row)
    *(u32x4*)(Ks + (t * 16 + lr) * RS + lc * 8) = k4;
    *(u32x4*)(Gs + (t * 16 + lr) * RS + lc * 8) = g4;
    *(u32x4*)(Os + lr * RS + lc * 8) = v4;
    vf[t] = *(const u32x4*)(Os + l15 * RS + kk * 8);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {   // operand fragments: row l15, dims 8 kk .. 8 kk + 7
    qf[t] = *(const u32x4*)(Qs + (t * 16 + l15) * RS + kk * 8);
    kf[t] = *(const u32x4*)(Ks + (t * 16 + l15) * RS + kk * 8);
    gf[t] = *(const u32x4*)(Gs + (t * 16 + l15) * RS + kk * 8);
  }
  // transposed fragment of X (row tile rt, 16-dim tile dt): lane 4 qq + pp of a 16-lane group supplies row 4 kk + qq, columns 4 pp .. 4 pp + 3
  const unsigned troff = (4 * kk + qq) * (RS * 2) + pp * 8;
  auto tfrag = [&](const unsigned short* X, int rt0, int rt1, int dt) {
    const unsigned a = lds_addr((const char*)X) + troff + dt * 32;
    u32x2 lo = abm_tr(a + rt0 * (16 * RS * 2)), hi = abm_tr(a + rt1 * (16 * RS * 2));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  const float c2 = scale * 1.4426950408889634f;
  const float ksc = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // one 16-token x 32-dim result tile: accumulator layout (token l15, dims 16 dt + 4 kk ..) -> the staging tile -> row form -> memory
  auto store_tile = [&](int mat, int t, const f32x4& v0, const f32x4& v1) {
    u32x2 u0, u1;
    u0[0] = pack_bf16x2(v0[0], v0[1]); u0[1] = pack_bf16x2(v0[2], v0[3]);
    u1[0] = pack_bf16x2(v1[0], v1[1]); u1[1] = pack_bf16x2(v1[2], v1[3]);
    *(u32x2*)(Os + l15 * RS + 4 * kk) = u0;
    *(u32x2*)(Os + l15 * RS + 16 + 4 * kk) = u1;
    const u32x4 row = *(const u32x4*)(Os + lr * RS + lc * 8);
    if (rlive[t]) *(u32x4*)(dqkv + rtok[t] * 3L * C + (long)mat * C + h * 32 + lc * 8) = row;
  };

  // ---- pass 1: queries in the columns ------------------------------------------------------------------------------------------
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const bool ilive = clive[it];
    const int ipos = cpos[it], isl = csl[it];
    const unsigned long long mrow = (((unsigned long long)(big ? unit : unit * SPT + isl) * n_head + h) * L + ipos) * L;
    f32x4 st[NT], dpt[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      st[jt] = mfma_bf16(kf[jt], qf[it], zero4);
      dpt[jt] = mfma_bf16(vf[jt], gf[it], zero4);
    }
    unsigned vm = 0;
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int jsl, jpos; bool jlive;
        decode(jt * 16 + 4 * kk + r, jsl, jpos, jlive);
        const bool valid = ilive && jlive && jsl == isl && (!causal || jpos <= ipos);
        vm |= (unsigned)valid << (jt * 4 + r);
        if (valid) mx = fmaxf(mx, st[jt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mm = (mx == -INFINITY) ? 0.f : mx;
    float lsum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = ((vm >> (jt * 4 + r)) & 1) ? __builtin_amdgcn_exp2f((st[jt][r] - mm) * c2) : 0.f;
        st[jt][r] = e;
        lsum += e;
      }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    const float inv_l = lsum > 0.f ? 1.0f / lsum : 0.f;
    float delta = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      unsigned m4 = 0xfu;
      if (p_drop > 0.f) {
        if ((L & 3) == 0) {      // the lane's four keys are one aligned run of mask indices of one sequence
          int jsl, jpos; bool jlive;
          decode(jt * 16 + 4 * kk, jsl, jpos, jlive);
          m4 = dropout_keep4(seed, mrow + jpos, p_drop);
        } else {
          m4 = 0u;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int jsl, jpos; bool jlive;
            decode(jt * 16 + 4 * kk + r, jsl, jpos, jlive);
            m4 |= (unsigned)dropout_keep(seed, mrow + jpos, p_drop) << r;
          }
        }
        Mb[(it * NT + jt) * 64 + kk * 16 + l15] = (unsigned char)m4;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = st[jt][r] * inv_l;
        float dp = dpt[jt][r];
        if (p_drop > 0.f) dp = ((m4 >> r) & 1u) ? dp * ksc : 0.f;             // d(dropped prob) -> d(prob)
        st[jt][r] = p;
        dpt[jt][r] = dp;
        delta += p * dp;
      }
    }
    delta += __shfl_xor(delta, 16);
    delta += __shfl_xor(delta, 32);
    if (kk == 0) {
      St[it * 16 + l15] = mm;
      St[ROWS + it * 16 + l15] = inv_l;
      St[2 * ROWS + it * 16 + l15] = delta;
    }
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dpt[jt][r] = st[jt][r] * (dpt[jt][r] - delta) * scale;   // dS^T
    f32x4 dq[2] = {zero4, zero4};
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      constexpr int dummy = 0; (void)dummy;
      const int j0 = 2 * jp, j1 = (2 * jp + 1 < NT) ? 2 * jp + 1 : 2 * jp;
      const u32x4 pf = pack8(dpt[j0], (2 * jp + 1 < NT) ? dpt[j1] : zero4);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma_bf16(tfrag(Ks, j0, j1, dt), pf, dq[dt]);
    }
    store_tile(0, it, dq[0], dq[1]);
  }

  // ---- pass 2: keys in the columns ---------------------------------------------------------------------------------------------
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const bool jlive = clive[jt];
    const int jpos = cpos[jt], jsl = csl[jt];
    f32x4 sv[NT], dp[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      sv[it] = mfma_bf16(qf[it], kf[jt], zero4);
      dp[it] = mfma_bf16(gf[it], vf[jt], zero4);
    }
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const f32x4 m4 = *(const f32x4*)(St + it * 16 + 4 * kk), il4 = *(const f32x4*)(St + ROWS + it * 16 + 4 * kk);
      const f32x4 de4 = *(const f32x4*)(St + 2 * ROWS + it * 16 + 4 * kk);
      const unsigned kw = p_drop > 0.f ? *(const unsigned*)(Mb + (it * NT + jt) * 64 + (l15 >> 2) * 16 + 4 * kk) >> (l15 & 3) : 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int isl, ipos; bool ilive;
        decode(it * 16 + 4 * kk + r, isl, ipos, ilive);
        const bool valid = ilive && jlive && isl == jsl && (!causal || jpos <= ipos);
        const float p = valid ? __builtin_amdgcn_exp2f((sv[it][r] - m4[r]) * c2) * il4[r] : 0.f;
        float d = dp[it][r], pd = p;
        if (p_drop > 0.f) {
          const bool keep = (kw >> (8 * r)) & 1u;          // pass 1's draw for (query 4 kk + r, this key)
          pd = keep ? p * ksc : 0.f;
          d = keep ? d * ksc : 0.f;
        }
        sv[it][r] = pd;                              // dropped probabilities (dV)
        dp[it][r] = p * (d - de4[r]) * scale;       // dS (dK)
      }
    }
    f32x4 dk[2] = {zero4, zero4}, dv[2] = {zero4, zero4};
#pragma unroll
    for (int ip = 0; ip < NP; ++ip) {
      const int i0 = 2 * ip, i1 = (2 * ip + 1 < NT) ? 2 * ip + 1 : 2 * ip;
      const u32x4 pfs = pack8(dp[i0], (2 * ip + 1 < NT) ? dp[i1] : zero4);
      const u32x4 pfp = pack8(sv[i0], (2 * ip + 1 < NT) ? sv[i1] : zero4);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dk[dt] = mfma_bf16(tfrag(Qs, i0, i1, dt), pfs, dk[dt]);
        dv[dt] = mfma_bf16(tfrag(Gs, i0, i1, dt), pfp, dv[dt]);
      }
    }
    store_tile(1, jt, dk[0], dk[1]);
    store_tile(2, jt, dv[0], dv[1]);
  }
}

// ---- the same for whole sequences of 2 - 4 tiles (16 < L <= 64), NT waves per (sequence, head) -----------------------------------------------
// In the kernel above ONE wave walks all NT x NT score tiles of its (sequence, head) twice: at L = 48 that is a 54 k-cycle serial chain at
// two waves per SIMD (200 registers, 16 KB of LDS per wave) -- 1.7 TB/s by the counters where the L <= 16 form reaches 4.1.  Here the NT
// waves of a head share the row images in LDS (each loads and later stores ONE 16-row tile): wave ti handles query tile ti in pass 1
// (row statistics, keep bits, dQ) and key tile ti in pass 2 (dK, dV), with one barrier between the passes for the statistics and keep
// bits of the other tiles.  A third of the chain per wave, fragments of the other tiles from LDS instead of a register file of their own.
template <int NT, int HG>   // HG heads per workgroup
__global__ __launch_bounds__(64 * NT * HG) void attn_bwd_split_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ dO,
                                                                      unsigned short* __restrict__ dqkv, int C, int n_head, TanteSeq sq, int causal,
                                                                      float scale, float p_drop, unsigned long long seed,
                                                                      const unsigned long long* __restrict__ seed_mix) {
  if (seed_mix) seed ^= *seed_mix;
  constexpr int ROWS = NT * 16, NP = (NT + 1) / 2, RS = 48;
  constexpr int HEAD_LDS = 4 * ROWS * RS * 2 + NT * 16 * RS * 2 + 3 * ROWS * 4 + NT * NT * 64;   // Q, K, V, dO images, NT staging tiles, statistics, keep bits
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int tid = threadIdx.x, lane = tid & 63, kk = lane >> 4, l15 = lane & 15, qq = l15 >> 2, pp = l15 & 3;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), hg = wave / NT, ti = wave - hg * NT;
  const int h = blockIdx.y * HG + hg;                      // n_head % HG == 0 (host): every wave has a head, every wave reaches the barriers
  char* base = sm_raw + hg * HEAD_LDS;
  unsigned short* Qs = (unsigned short*)base;
  unsigned short* Ks = Qs + ROWS * RS;
  unsigned short* Vs = Ks + ROWS * RS;
  unsigned short* Gs = Vs + ROWS * RS;
  unsigned short* Os = Gs + ROWS * RS + ti * 16 * RS;      // this wave's staging tile
  float* St = (float*)(Gs + ROWS * RS + NT * 16 * RS);     // m [ROWS], 1/l [ROWS], delta [ROWS]
  unsigned char* Mb = (unsigned char*)(St + 3 * ROWS);
  const int L = sq.L, seq = blockIdx.x;
  // ---- this wave's 16 rows (tile ti) in row form: lanes 4 r .. 4 r + 3 = the 64 bytes of (token row r, this head) ------------------------
  const int lr = lane >> 2, lc = lane & 3, rpos = ti * 16 + lr;
  const bool rlive = rpos < L;
  const long rtok = rlive ? (long)(seq / sq.n_s0) * sq.S1 + (long)(seq % sq.n_s0) * sq.S0 + (long)(rpos / sq.n_l0) * sq.P1 + (long)(rpos % sq.n_l0) * sq.P0 : 0;
  {
    u32x4 q4 = u32x4{0u, 0u, 0u, 0u}, k4 = q4, v4 = q4, g4 = q4;
    if (rlive) {
      const unsigned short* r = qkv + rtok * 3L * C + h * 32 + lc * 8;
      q4 = *(const u32x4*)r;
      k4 = *(const u32x4*)(r + C);
      v4 = *(const u32x4*)(r + 2 * C);
      g4 = *(const u32x4*)(dO + rtok * (long)C + h * 32 + lc * 8);
    }
    *(u32x4*)(Qs + rpos * RS + lc * 8) = q4;
    *(u32x4*)(Ks + rpos * RS + lc * 8) = k4;
    *(u32x4*)(Vs + rpos * RS + lc * 8) = v4;
    *(u32x4*)(Gs + rpos * RS + lc * 8) = g4;
  }
  __syncthreads();
  auto rfrag = [&](const unsigned short* X, int t) { return *(const u32x4*)(X + (t * 16 + l15) * RS + kk * 8); };   // row l15 of tile t, dims 8 kk ..
  const unsigned troff = (4 * kk + qq) * (RS * 2) + pp * 8;
  auto tfrag = [&](const unsigned short* X, int rt0, int rt1, int dt) {
    const unsigned a = lds_addr((const char*)X) + troff + dt * 32;
    u32x2 lo = abm_tr(a + rt0 * (16 * RS * 2)), hi = abm_tr(a + rt1 * (16 * RS * 2));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
  };
  const float c2 = scale * 1.4426950408889634f;
  const float ksc = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  auto store_tile = [&](int mat, const f32x4& v0, const f32x4& v1) {      // a 16-token x 32-dim result of tile ti: accumulator layout -> rows
    u32x2 u0, u1;
    u0[0] = pack_bf16x2(v0[0], v0[1]); u0[1] = pack_bf16x2(v0[2], v0[3]);
    u1[0] = pack_bf16x2(v1[0], v1[1]); u1[1] = pack_bf16x2(v1[2], v1[3]);
    *(u32x2*)(Os + l15 * RS + 4 * kk) = u0;
    *(u32x2*)(Os + l15 * RS + 16 + 4 * kk) = u1;
    const u32x4 row = *(const u32x4*)(Os + lr * RS + lc * 8);
    if (rlive) *(u32x4*)(dqkv + rtok * 3L * C + (long)mat * C + h * 32 + lc * 8) = row;
  };
  const int mypos = ti * 16 + l15;                 // pass 1: this lane's query; pass 2: this lane's key
  const bool mylive = mypos < L;

  // ---- pass 1: the queries of tile ti in the columns ---------------------------------------------------------------------------------------
  {
    const u32x4 qf = rfrag(Qs, ti), gf = rfrag(Gs, ti);
    const unsigned long long mrow = (((unsigned long long)seq * n_head + h) * L + mypos) * L;
    f32x4 st[NT], dpt[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      st[jt] = mfma_bf16(rfrag(Ks, jt), qf, zero4);
      dpt[jt] = mfma_bf16(rfrag(Vs, jt), gf, zero4);
    }
    unsigned vm = 0;
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jpos = jt * 16 + 4 * kk + r;
        const bool valid = mylive && jpos < L && (!causal || jpos <= mypos);
        vm |= (unsigned)valid << (jt * 4 + r);
        if (valid) mx = fmaxf(mx, st[jt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mm = (mx == -INFINITY) ? 0.f : mx;
    float lsum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = ((vm >> (jt * 4 + r)) & 1) ? __builtin_amdgcn_exp2f((st[jt][r] - mm) * c2) : 0.f;
        st[jt][r] = e;
        lsum += e;
      }
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    const float inv_l = lsum > 0.f ? 1.0f / lsum : 0.f;
    float delta = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      unsigned m4 = 0xfu;
      if (p_drop > 0.f) {
        if ((L & 3) == 0) {
          m4 = dropout_keep4(seed, mrow + jt * 16 + 4 * kk, p_drop);
        } else {
          m4 = 0u;
#pragma unroll
          for (int r = 0; r < 4; ++r) m4 |= (unsigned)dropout_keep(seed, mrow + jt * 16 + 4 * kk + r, p_drop) << r;
        }
        Mb[(ti * NT + jt) * 64 + kk * 16 + l15] = (unsigned char)m4;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = st[jt][r] * inv_l;
        float dp = dpt[jt][r];
        if (p_drop > 0.f) dp = ((m4 >> r) & 1u) ? dp * ksc : 0.f;
        st[jt][r] = p;
        dpt[jt][r] = dp;
        delta += p * dp;
      }
    }
    delta += __shfl_xor(delta, 16);
    delta += __shfl_xor(delta, 32);
    if (kk == 0) {
      St[mypos] = mm;
      St[ROWS + mypos] = inv_l;
      St[2 * ROWS + mypos] = delta;
    }
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dpt[jt][r] = st[jt][r] * (dpt[jt][r] - delta) * scale;   // dS^T
    f32x4 dq[2] = {zero4, zero4};
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const int j0 = 2 * jp, j1 = (2 * jp + 1 < NT) ? 2 * jp + 1 : 2 * jp;
      const u32x4 pf = pack8(dpt[j0], (2 * jp + 1 < NT) ? dpt[j1] : zero4);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma_bf16(tfrag(Ks, j0, j1, dt), pf, dq[dt]);
    }
    store_tile(0, dq[0], dq[1]);
  }
  __syncthreads();      // every tile's statistics and keep bits are in LDS

  // ---- pass 2: the keys of tile ti in the columns --------------------------------------------------------------------------------------------
  {
    const u32x4 kf = rfrag(Ks, ti), vf = rfrag(Vs, ti);
    f32x4 sv[NT], dp[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      sv[it] = mfma_bf16(rfrag(Qs, it), kf, zero4);
      dp[it] = mfma_bf16(rfrag(Gs, it), vf, zero4);
    }
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const f32x4 m4 = *(const f32x4*)(St + it * 16 + 4 * kk), il4 = *(const f32x4*)(St + ROWS + it * 16 + 4 * kk);
      const f32x4 de4 = *(const f32x4*)(St + 2 * ROWS + it * 16 + 4 * kk);
      const unsigned kw = p_drop > 0.f ? *(const unsigned*)(Mb + (it * NT + ti) * 64 + (l15 >> 2) * 16 + 4 * kk) >> (l15 & 3) : 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ipos = it * 16 + 4 * kk + r;
        const bool valid = ipos < L && mylive && (!causal || mypos <= ipos);
        const float p = valid ? __builtin_amdgcn_exp2f((sv[it][r] - m4[r]) * c2) * il4[r] : 0.f;
        float d = dp[it][r], pd = p;
        if (p_drop > 0.f) {
          const bool keep = (kw >> (8 * r)) & 1u;
          pd = keep ? p * ksc : 0.f;
          d = keep ? d * ksc : 0.f;
        }
        sv[it][r] = pd;
        dp[it][r] = p * (d - de4[r]) * scale;
      }
    }
    f32x4 dk[2] = {zero4, zero4}, dv[2] = {zero4, zero4};
#pragma unroll
    for (int ip = 0; ip < NP; ++ip) {
      const int i0 = 2 * ip, i1 = (2 * ip + 1 < NT) ? 2 * ip + 1 : 2 * ip;
      const u32x4 pfs = pack8(dp[i0], (2 * ip + 1 < NT) ? dp[i1] : zero4);
      const u32x4 pfp = pack8(sv[i0], (2 * ip + 1 < NT) ? sv[i1] : zero4);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dk[dt] = mfma_bf16(tfrag(Qs, i0, i1, dt), pfs, dk[dt]);
        dv[dt] = mfma_bf16(tfrag(Gs, i0, i1, dt), pfp, dv[dt]);
      }
    }
    store_tile(1, dk[0], dk[1]);
    store_tile(2, dv[0], dv[1]);
  }
}

template <int NT, int HG>
void launch_attn_bwd_split(const void* qkv, const void* dO, void* dqkv, int C, int n_head, const TanteSeq& sq, int causal, float p_drop,
                           unsigned long long seed, hipStream_t s) {
  const size_t lds = (size_t)HG * (4 * NT * 16 * 96 + NT * 16 * 96 + 3 * NT * 16 * 4 + NT * NT * 64);
  static TantePerDevice attr;
  attr.once([&] { (void)hipFuncSetAttribute((const void*)attn_bwd_split_kernel<NT, HG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  hipLaunchKernelGGL((attn_bwd_split_kernel<NT, HG>), dim3(sq.nseq, n_head / HG), dim3(64 * NT * HG), lds, s, (const unsigned short*)qkv,
                     (const unsigned short*)dO, (unsigned short*)dqkv, C, n_head, sq, causal, 1.0f / sqrtf(32.0f), p_drop, seed, tante_seed_mix_ptr());
}

template <int NT>
void launch_attn_bwd_mfma(const void* qkv, const void* dO, void* dqkv, int C, int n_head, const TanteSeq& sq, int SPT, int units, int causal,
                          float p_drop, unsigned long long seed, hipStream_t s) {
  const size_t lds = 4 * (size_t)(3 * NT * 16 * 96 + 16 * 96 + 3 * NT * 16 * 4 + NT * NT * 64);      // = 4 waves x WAVE_LDS
  static TantePerDevice attr;
  attr.once([&] { (void)hipFuncSetAttribute((const void*)attn_bwd_mfma_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  hipLaunchKernelGGL((attn_bwd_mfma_kernel<NT>), dim3(units, (n_head + 3) / 4), dim3(256), lds, s, (const unsigned short*)qkv,
                     (const unsigned short*)dO, (unsigned short*)dqkv, C, n_head, sq, SPT, causal, 1.0f / sqrtf(32.0f), p_drop, seed, tante_seed_mix_ptr());
}
bool try_attn_bwd_mfma(const void* qkv, const void* dO, void* dqkv, int dtype, int C, int n_head, const TanteSeq& sq, int causal, float p_drop,
                       unsigned long long seed, hipStream_t s) {
  const bool off = tante_opt("TANTE_ATTN_BWD_VALU", 0) != 0;
  if (off || dtype != TANTE_BF16 || C != n_head * 32 || sq.L > 64 || sq.L < 1) return false;
  if (((uintptr_t)qkv | (uintptr_t)dO | (uintptr_t)dqkv) & 15) return false;
  const int L = sq.L;
  const int SPT = L >= 16 ? 1 : 16 / L;
  const int units = L >= 16 ? sq.nseq : (sq.nseq + SPT - 1) / SPT;
  const bool no_split = tante_opt("TANTE_ATTN_BWD_NO_SPLIT", 0) != 0;
  if (!no_split && L > 16) {      // whole sequences of 2 - 4 tiles: NT waves per (sequence, head)
    const int hg_env = tante_opt("TANTE_ATTN_BWD_HG", 1);      // heads per workgroup: 29.2 / 31.2 / 36.4 us at 1 / 2 / 4 (L = 48, cfg3)
    const int nt = (L + 15) / 16;
    if (hg_env == 2 && n_head % 2 == 0) {
      if (nt == 2) launch_attn_bwd_split<2, 2>(qkv, dO, dqkv, C, n_head, sq, causal, p_drop, seed, s);
      else if (nt == 3) launch_attn_bwd_split<3, 2>(qkv, dO, dqkv, C, n_head, sq, causal, p_drop, seed, s);
      else launch_attn_bwd_split<4, 2>(qkv, dO, dqkv, C, n_head, sq, causal, p_drop, seed, s);
    } else if (hg_env == 4 && n_head % 4 == 0) {
      if (nt == 2) launch_attn_bwd_split<2, 4>(qkv, dO, dqkv, C, n_head, sq, causal, p_drop, seed, s);
      else if (nt == 3) launch_attn_bwd_split<3, 4>(qkv, dO, dqkv, C, n_head, sq, causal, p_drop, seed, s);
      else launch_attn_bwd_split<4, 4>(qkv, dO, dqkv, C, n_head, sq, causal, p_drop, seed, s);
    } else {
      if (nt == 2) launch_attn_bwd_split<2, 1>(qkv, dO, dqkv, C, n_head, sq, causal, p_drop, seed, s);
      else if (nt == 3) launch_attn_bwd_split<3, 1>(qkv, dO, dqkv, C, n_head, sq, causal, p_drop, seed, s);
      else launch_attn_bwd_split<4, 1>(qkv, dO, dqkv, C, n_head, sq, causal, p_drop, seed, s);
    }
    return true;
  }
  switch (L >= 16 ? (L + 15) / 16 : 1) {
    case 1: launch_attn_bwd_mfma<1>(qkv, dO, dqkv, C, n_head, sq, SPT, units, causal, p_drop, seed, s); break;
    case 2: launch_attn_bwd_mfma<2>(qkv, dO, dqkv, C, n_head, sq, SPT, units, causal, p_drop, seed, s); break;
    case 3: launch_attn_bwd_mfma<3>(qkv, dO, dqkv, C, n_head, sq, SPT, units, causal, p_drop, seed, s); break;
    default: launch_attn_bwd_mfma<4>(qkv, dO, dqkv, C, n_head, sq, SPT, units, causal, p_drop, seed, s); break;
  }
  return true;
}

// ---- axis propagator weight gradients: dW[a][j] (+)= sum_{o,i} U[o][a][i] V[o][j][i],  db[a] (+)= sum_{o,i} U[o][a][i] ---------------
// U, V are (outer, n, inner) fp32 in the residual stream's own layout: the contraction index i is the contiguous one, so both MFMA
// operands are natural "row, consecutive k" fragments read straight from global memory (a lane loads 4 consecutive i of row l15; MFMA s
// of the four uses element s of every lane, i.e. k slot kk <-> i = i0 + 4 kk + s on both sides).  fp32 MFMA (16x16x4): the propagators
// stay in fp32 in both compute modes.  The generic weight-gradient kernel gathers these operands element by element with stride
// `inner` (85 us for 2 x 25 MB); this one streams them.
// Round 2: (1) three chunks in flight per wave and up to four workgroups per CU (two chunks x eight waves per CU left the kernel
// latency-bound at 1.1 - 2 TB/s); (2) short axes (n <= 8, the temporal propagator's n = 4) pack G = 16 / npad SEGMENTS of the
// contraction into the 16 rows of the MFMA tile -- row (s, a) = line a of segment s, both operands alike -- and keep the G diagonal
// blocks of the product: a quarter of the load instructions, each one whole; (3) the workgroups' n x n partials no longer go to dW by
// global atomics on the same n^2 addresses from every workgroup (the kernel's time grew linearly with the grid: 24 / 35 / 46 us at
// 512 / 1024 / 1536 workgroups -- device-scope atomics are served behind the XCDs' L2s): each workgroup STORES its partial to a
// caller-owned workspace, and the last of every AW_GS workgroups to finish (an arrival counter per group) sums the group's partials
// and adds ONE n x n to dW: a sixteenth of the atomics, no second launch.
constexpr int AW_GS = TANTE_AW_GS;                        // workgroups per reduction group (common.hip.h: axis_bwd.hip shares the workspace)
constexpr int AW_MAXWG = TANTE_AW_MAXWG;                  // grid cap with a workspace
constexpr int AW_SLAB = TANTE_AW_SLAB;                    // floats per partial: dW (n <= 64) then db
constexpr long AW_WS_FLOATS = TANTE_AW_WS_FLOATS;         // partials + one arrival counter per group

template <int NT, int G>
__global__ __launch_bounds__(512) void axis_wgrad_kernel(const float* __restrict__ U, const float* __restrict__ V, long outer, int n, long inner,
                                                         float* __restrict__ dW, float* __restrict__ db, long n_chunks, float* __restrict__ ws) {
  static_assert(G == 1 || NT == 1, "segments are packed into a single 16-row tile");
  constexpr int NP = NT * 16, NPAD = 16 / G, CW = 16 * G;      // CW: contraction elements per chunk
  __shared__ float red[NP * NP + NP];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  const int nthr = blockDim.x, nwv = nthr >> 6;      // 4 waves, or 8 for the long axes (fewer, larger partials)
  for (int i = tid; i < NP * NP + NP; i += nthr) red[i] = 0.f;
  __syncthreads();
  const long cpo = inner / CW;   // chunks of the contraction per outer index
  f32x4 acc[NT][NT], accb[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // this lane's row of tile t: line (t * 16 + l15) % NPAD-wise -> (segment, line)
  auto load = [&](long c, f32x4 (&a)[NT], f32x4 (&b)[NT]) {
    const long o = c / cpo;
    const long base = o * n * inner + (c - o * cpo) * CW + kk * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int row = t * 16 + l15, seg = G > 1 ? row / NPAD : 0, line = G > 1 ? row % NPAD : row;
      a[t] = b[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (line < n) {
        a[t] = *(const f32x4*)(U + base + (long)line * inner + 16 * seg);
        b[t] = *(const f32x4*)(V + base + (long)line * inner + 16 * seg);
      }
    }
  };
  auto fma_chunk = [&](const f32x4 (&a)[NT], const f32x4 (&b)[NT]) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int at = 0; at < NT; ++at) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) acc[at][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[at][s4], b[jt][s4], acc[at][jt], 0, 0, 0);
        if (db) accb[at] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[at][s4], 1.0f, accb[at], 0, 0, 0);
      }
  };
  const long stride = (long)gridDim.x * nwv;
  long c = (long)blockIdx.x * nwv + wave;
  f32x4 a0[NT], b0[NT], a1[NT], b1[NT], a2[NT], b2[NT];
  if (c < n_chunks) load(c, a0, b0);
  if (c + stride < n_chunks) load(c + stride, a1, b1);
  while (c < n_chunks) {        // three chunks per trip: two later chunks' loads fly under this chunk's MFMAs
    if (c + 2 * stride < n_chunks) load(c + 2 * stride, a2, b2);
    fma_chunk(a0, b0);
    if (c + stride >= n_chunks) break;
    if (c + 3 * stride < n_chunks) load(c + 3 * stride, a0, b0);
    fma_chunk(a1, b1);
    if (c + 2 * stride >= n_chunks) break;
    if (c + 4 * stride < n_chunks) load(c + 4 * stride, a1, b1);
    fma_chunk(a2, b2);
    c += 3 * stride;
  }
  // D[row][col]: lane holds col = l15, rows 4 kk + r.  G > 1: row = (s, a), col = (s', j): the diagonal blocks s == s' are the product
  if constexpr (G == 1 && NT <= 3) {
    // the waves' partial tiles are summed pairwise through LDS in register order (one 16-byte piece per lane and tile), and wave 0 writes
    // the workgroup's tile: 36 LDS float atomics per lane and wave on the SAME 2 304 addresses were most of this kernel's time at n = 48
    __shared__ f32x4 slab[4][(NT * NT + NT) * 64];
    for (int half = nwv >> 1; half >= 1; half >>= 1) {
      if (wave >= half && wave < 2 * half) {
        f32x4* sl = slab[wave - half];
#pragma unroll
        for (int at = 0; at < NT; ++at) {
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) sl[(at * NT + jt) * 64 + lane] = acc[at][jt];
          sl[(NT * NT + at) * 64 + lane] = accb[at];
        }
      }
      __syncthreads();
      if (wave < half) {
        const f32x4* sl = slab[wave];
#pragma unroll
        for (int at = 0; at < NT; ++at) {
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) acc[at][jt] += sl[(at * NT + jt) * 64 + lane];
          accb[at] += sl[(NT * NT + at) * 64 + lane];
        }
      }
      __syncthreads();
    }
    if (wave == 0) {
#pragma unroll
      for (int at = 0; at < NT; ++at) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(at * 16 + 4 * kk + r) * NP + jt * 16 + l15] = acc[at][jt][r];
        if (db && l15 == 0)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[NP * NP + at * 16 + 4 * kk + r] = accb[at][r];
      }
    }
  } else
#pragma unroll
  for (int at = 0; at < NT; ++at) {
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (G > 1) {
          const int row = 4 * kk + r;
          if (row / NPAD == l15 / NPAD) atomicAdd(&red[(row % NPAD) * NP + (l15 % NPAD)], acc[at][jt][r]);
        } else {
          atomicAdd(&red[(at * 16 + 4 * kk + r) * NP + jt * 16 + l15], acc[at][jt][r]);
        }
      }
    if (db && l15 == 0)
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&red[NP * NP + (G > 1 ? (4 * kk + r) % NPAD : at * 16 + 4 * kk + r)], accb[at][r]);
  }
  __syncthreads();
  if (!ws) {      // no workspace: every workgroup adds its partial straight into the parameter's gradient
    for (int i = tid; i < n * n; i += nthr) atomicAdd(&dW[i], red[(i / n) * NP + (i % n)]);
    if (db && tid < n) atomicAdd(&db[tid], red[NP * NP + tid]);
    return;
  }
  // agent-coherent (write-through) stores + a wait for their acknowledgement instead of a release fence: on this part a device-scope
  // release writes back the whole L2 (buffer_wbl2), and a thousand workgroups doing that quadrupled the kernel's time
  float* mine = ws + (long)blockIdx.x * AW_SLAB;
  for (int i = tid; i < n * n; i += nthr) __hip_atomic_store(mine + i, red[(i / n) * NP + (i % n)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (db && tid < n) __hip_atomic_store(mine + 64 * 64 + tid, red[NP * NP + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0): this thread's stores have reached the coherence point
  __syncthreads();
  const int grp = blockIdx.x / AW_GS, first = grp * AW_GS, members = min(AW_GS, (int)gridDim.x - first);
  unsigned* counter = (unsigned*)(ws + (long)AW_MAXWG * AW_SLAB) + grp;
  if (tid == 0) s_last = atomicAdd(counter, 1u) == (unsigned)(members - 1);
  __syncthreads();
  if (!s_last) return;
  for (int i = tid; i < n * n + (db ? n : 0); i += nthr) {
    const int off = i < n * n ? i : 64 * 64 + (i - n * n);
    float v[AW_GS];
#pragma unroll
    for (int m = 0; m < AW_GS; ++m)      // agent-scope loads (the partials of other XCDs' workgroups), all in flight at once
      v[m] = m < members ? __hip_atomic_load(ws + (long)(first + m) * AW_SLAB + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < AW_GS; ++m) sum += v[m];
    atomicAdd(i < n * n ? dW + i : db + (i - n * n), sum);
  }
  if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- LayerNorm affine folded into the consumer's weight (train path): We = W diag(gamma), be = b + W beta, and the fold's backward ------
// One launch each.  fwd: blocks [0, nbe) write We (float4 pieces), the blocks after them one be row per wave.
__device__ __forceinline__ void fold_fwd_body(int bid, const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ gamma,
                                              const float* __restrict__ beta, int N, int K, float* __restrict__ We, float* __restrict__ be, int nbe) {
  if (bid < nbe) {
    const long i4 = (long)bid * 256 + threadIdx.x;
    if (i4 * 4 >= (long)N * K) return;
    const int k = (int)((i4 * 4) % K);
    const f32x4 w = *(const f32x4*)(W + i4 * 4), g = *(const f32x4*)(gamma + k);
    *(f32x4*)(We + i4 * 4) = w * g;
    return;
  }
  const int n = (bid - nbe) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += W[(long)n * K + k] * beta[k];
  s = wave_sum(s);
  if (lane == 0) be[n] = (b ? b[n] : 0.f) + s;
}
__global__ __launch_bounds__(256) void fold_fwd_kernel(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int N, int K, float* __restrict__ We,
                                                       float* __restrict__ be, int nbe) {
  fold_fwd_body((int)blockIdx.x, W, b, gamma, beta, N, K, We, be, nbe);
}
// every fold of a model in one launch (tante_fold_fwd_multi): 18 launches of ~4.7 us per train step otherwise
constexpr int FOLDF_MAX = 24;
struct FoldFwdBatch {
  const float* W[FOLDF_MAX]; const float* b[FOLDF_MAX]; const float* gamma[FOLDF_MAX]; const float* beta[FOLDF_MAX];
  float* We[FOLDF_MAX]; float* be[FOLDF_MAX];
  int N[FOLDF_MAX], K[FOLDF_MAX], nbe[FOLDF_MAX], first[FOLDF_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void fold_fwd_multi_kernel(FoldFwdBatch B) {
  int e = 0;
  while (e + 1 < B.n && (int)blockIdx.x >= B.first[e + 1]) ++e;
  fold_fwd_body((int)blockIdx.x - B.first[e], B.W[e], B.b[e], B.gamma[e], B.beta[e], B.N[e], B.K[e], B.We[e], B.be[e], B.nbe[e]);
}
// bwd, from the accumulated gradients GW (N, K), Gb (N) of the folded pair:
//   dW += GW diag(gamma) + Gb beta^T,  db += Gb,  dgamma[k] += sum_n GW[n][k] W[n][k],  dbeta[k] += sum_n W[n][k] Gb[n]
// One pass: block (slab of FB_ROWS rows, 256-column chunk), thread = column: the elementwise update and the slab's share of the two
// column reductions (atomics) from the same loads.  CLEAR: the accumulators are zeroed as they are read, so the caller can keep them
// across steps without a fill per weight and step (GW by its only reader; Gb only when there is a single column chunk, K <= 256).
constexpr int FB_ROWS = 16;
__device__ __forceinline__ void fold_bwd_body(int bid, float* __restrict__ GW, float* __restrict__ Gb, const float* __restrict__ W,
                                              const float* __restrict__ gamma, const float* __restrict__ beta, int N, int K,
                                              float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dgamma,
                                              float* __restrict__ dbeta, int clear) {
  const int kb = (K + 255) / 256;
  const int slab = bid / kb, kc = bid % kb, k = kc * 256 + threadIdx.x;
  const int n0 = slab * FB_ROWS, n1 = min(N, n0 + FB_ROWS);
  __shared__ float gbs[FB_ROWS];
  if (threadIdx.x < n1 - n0) gbs[threadIdx.x] = Gb[n0 + threadIdx.x];
  __syncthreads();
  if (kc == 0 && threadIdx.x < n1 - n0) {
    if (db) db[n0 + threadIdx.x] += gbs[threadIdx.x];
    if (clear && kb == 1) Gb[n0 + threadIdx.x] = 0.0f;
  }
  if (k >= K) return;
  const float g = gamma[k], bt = beta[k];
  float sg = 0.f, sb = 0.f;
#pragma unroll 4
  for (int n = n0; n < n1; ++n) {
    const long i = (long)n * K + k;
    const float gw = GW[i], w = W[i], gb = gbs[n - n0];
    dW[i] += gw * g + bt * gb;
    sg += gw * w;
    sb += w * gb;
    if (clear) GW[i] = 0.0f;
  }
  atomicAdd(&dgamma[k], sg);
  atomicAdd(&dbeta[k], sb);
}
__global__ __launch_bounds__(256) void fold_bwd_kernel(float* __restrict__ GW, float* __restrict__ Gb, const float* __restrict__ W,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int N, int K,
                                                       float* __restrict__ dW, float* __restrict__ db, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, int clear) {
  fold_bwd_body((int)blockIdx.x, GW, Gb, W, gamma, beta, N, K, dW, db, dgamma, dbeta, clear);
}
// every fold of a model in ONE launch (tante_fold_bwd_multi): a fold's backward is 1 - 3 MB of elementwise work that takes 11 us as a launch
// of its own (its latency chain, not its bytes), and a train step has two per TransformerBlock
constexpr int FOLD_MAX = 24;
struct FoldBatch {
  float* GW[FOLD_MAX]; float* Gb[FOLD_MAX];
  const float* W[FOLD_MAX]; const float* gamma[FOLD_MAX]; const float* beta[FOLD_MAX];
  float* dW[FOLD_MAX]; float* db[FOLD_MAX]; float* dgamma[FOLD_MAX]; float* dbeta[FOLD_MAX];
  int N[FOLD_MAX], K[FOLD_MAX], first[FOLD_MAX + 1];      // first[e]: the first block of entry e
  int n;
};
__global__ __launch_bounds__(256) void fold_bwd_multi_kernel(FoldBatch B, int clear) {
  int e = 0;
  while (e + 1 < B.n && (int)blockIdx.x >= B.first[e + 1]) ++e;      // block-uniform
  fold_bwd_body((int)blockIdx.x - B.first[e], B.GW[e], B.Gb[e], B.W[e], B.gamma[e], B.beta[e], B.N[e], B.K[e], B.dW[e], B.db[e], B.dgamma[e],
                B.dbeta[e], clear);
}

// ---- axis propagator backward: y = x + W2 gelu(W1 x + b1) + b2 along an axis of (outer, n, inner) -----------------------
// one lane per column; writes dx = dy + W1^T (gelu'(pre) * (W2^T dy)), and materialises h = gelu(pre) and dpre for the
// weight-gradient GEMMs (same (outer, n, inner) layout, fp32)
template <int N>
__global__ __launch_bounds__(256) void axis_mlp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, long outer, int n,
                                                           long inner, const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, float* __restrict__ dx, float* __restrict__ hbuf,
                                                           float* __restrict__ dpre) {
  __shared__ float w1s[N * N], w2s[N * N], b1s[N];
  for (int idx = threadIdx.x; idx < N * N; idx += 256) {
    const int j = idx / N, a = idx % N;
    const bool in = j < n && a < n;
    w1s[idx] = in ? w1[j * n + a] : 0.f;   // [j][a]
    w2s[idx] = in ? w2[j * n + a] : 0.f;   // [a_out][j_hidden] stored as [row][col]
  }
  if (threadIdx.x < N) b1s[threadIdx.x] = threadIdx.x < n ? b1[threadIdx.x] : 0.f;
  __syncthreads();
  const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= outer * inner) return;
  const long o = col / inner, i = col - o * inner;
  const long base = o * (long)n * inner + i;
  float xv[N], gy[N];
#pragma unroll
  for (int a = 0; a < N; ++a) {
    xv[a] = a < n ? x[base + (long)a * inner] : 0.f;
    gy[a] = a < n ? dy[base + (long)a * inner] : 0.f;
  }
  float gx[N];
#pragma unroll
  for (int a = 0; a < N; ++a) gx[a] = gy[a];
  for (int j = 0; j < n; ++j) {
    float pre = b1s[j];
#pragma unroll
    for (int a = 0; a < N; ++a) pre += w1s[j * N + a] * xv[a];
    float dh = 0.f;  // (W2^T dy)_j = sum_a W2[a][j] dy_a
#pragma unroll
    for (int a = 0; a < N; ++a) dh += w2s[a * N + j] * gy[a];
    const float dp = dh * act_df(pre, TANTE_ACT_GELU_ERF);
    hbuf[base + (long)j * inner] = gelu_erf_f(pre);
    dpre[base + (long)j * inner] = dp;
#pragma unroll
    for (int a = 0; a < N; ++a) gx[a] += w1s[j * N + a] * dp;
  }
#pragma unroll
  for (int a = 0; a < N; ++a)
    if (a < n) dx[base + (long)a * inner] = gx[a];
}

// ---- residual dropout: out = res + keep * y / (1 - p)  (self.drop(y) of attn_backbone.py:81-82) and its backward -----------------
__global__ void dropout_add_kernel(const void* __restrict__ y, int y_dtype, const float* __restrict__ res, float p, unsigned long long seed,
                                   long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = dropout_keep(seed, (unsigned long long)i, p) ? ldx(y, y_dtype, i) / (1.0f - p) : 0.0f;
  out[i] = res[i] + v;
}
__global__ void dropout_bwd_kernel(const float* __restrict__ dout, float p, unsigned long long seed, long n, void* __restrict__ dy, int y_dtype) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  stx(dy, y_dtype, i, dropout_keep(seed, (unsigned long long)i, p) ? dout[i] / (1.0f - p) : 0.0f);
}

__global__ void dropout_add_vec_kernel(const void* __restrict__ y, int y_dtype, const float* __restrict__ res, float p, unsigned long long seed,
                                       long n4, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = ld4(y, y_dtype, 4 * i);
  f32x4 r = *(const f32x4*)(res + 4 * i);
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] += dropout_keep(seed, (unsigned long long)(4 * i + j), p) ? v[j] / (1.0f - p) : 0.0f;   // the scalar kernel's expression
  *(f32x4*)(out + 4 * i) = r;
}
__global__ void dropout_bwd_vec_kernel(const float* __restrict__ dout, float p, unsigned long long seed, long n4, void* __restrict__ dy, int y_dtype) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 g = *(const f32x4*)(dout + 4 * i);
#pragma unroll
  for (int j = 0; j < 4; ++j) g[j] = dropout_keep(seed, (unsigned long long)(4 * i + j), p) ? g[j] / (1.0f - p) : 0.0f;
  st4(dy, y_dtype, 4 * i, g);
}

template <int N>
void launch_axis_bwd(const float* x, const float* dy, long outer, int n, long inner, const float* w1, const float* b1, const float* w2,
                     float* dx, float* h, float* dpre, hipStream_t s) {
  const long cols = outer * inner;
  hipLaunchKernelGGL(axis_mlp_bwd_kernel<N>, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, x, dy, outer, n, inner, w1, b1, w2, dx, h, dpre);
}

template <int D, bool LB16>
void launch_attn_bwd_t(const void* qkv, const void* dO, void* dqkv, int dtype, int C, int n_head, const TanteSeq& sq, int causal, float p_drop,
                       unsigned long long seed, hipStream_t s) {
  const int G = 128 / sq.L;
  const size_t lds = 3 * 128 * sizeof(float) + 4 * 128 * (size_t)(D + (LB16 ? 8 : 4)) * (LB16 ? 2 : 4);
  static TantePerDevice attr;
  if (lds > 64 * 1024) attr.once([&] {
    hipFuncSetAttribute((const void*)attn_bwd_small_kernel<D, LB16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL((attn_bwd_small_kernel<D, LB16>), dim3((sq.nseq + G - 1) / G, n_head), dim3(128), lds, s, qkv, dO, dqkv, dtype, C, sq, G,
                     causal, 1.0f / sqrtf((float)D), p_drop, seed);
}
template <int D>
void launch_attn_bwd(const void* qkv, const void* dO, void* dqkv, int dtype, int C, int n_head, const TanteSeq& sq, int causal, float p_drop,
                     unsigned long long seed, hipStream_t s) {
  if (dtype == TANTE_BF16) launch_attn_bwd_t<D, true>(qkv, dO, dqkv, dtype, C, n_head, sq, causal, p_drop, seed, s);
  else launch_attn_bwd_t<D, false>(qkv, dO, dqkv, dtype, C, n_head, sq, causal, p_drop, seed, s);
}

}  // namespace

extern "C" int tante_layernorm_fwd(const float* x, int64_t M, int C, float eps, void* xhat, int out_dtype, float* stats, void* stream) {
  if (!x || !xhat || !stats || M <= 0 || C <= 0) TANTE_FAIL(-1, "tante_layernorm_fwd: bad argument");
  const dim3 grid((unsigned)((M + 3) / 4));
  const bool vec = ((uintptr_t)x % 16) == 0 && ((uintptr_t)xhat % 16) == 0;
  if (vec && C == 256) hipLaunchKernelGGL(ln_fwd_vec_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, (long)M, eps, xhat, out_dtype, stats);
  else if (vec && C == 512) hipLaunchKernelGGL(ln_fwd_vec_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, (long)M, eps, xhat, out_dtype, stats);
  else hipLaunchKernelGGL(ln_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (long)M, C, eps, xhat, out_dtype, stats);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_layernorm_bwd(const void* dxhat, int g_dtype, const float* x, const float* stats, const float* dskip, int64_t M, int C,
                                   float* dx, void* stream) {
  if (!dxhat || !x || !stats || !dx || M <= 0 || C <= 0) TANTE_FAIL(-1, "tante_layernorm_bwd: bad argument");
  const dim3 grid((unsigned)((M + 3) / 4));
  const bool vec = (((uintptr_t)x | (uintptr_t)dxhat | (uintptr_t)dx | (uintptr_t)dskip) % 16) == 0;
  if (vec && C == 256) hipLaunchKernelGGL(ln_bwd_vec_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, dxhat, g_dtype, x, stats, dskip, (long)M, dx);
  else if (vec && C == 512) hipLaunchKernelGGL(ln_bwd_vec_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, dxhat, g_dtype, x, stats, dskip, (long)M, dx);
  else hipLaunchKernelGGL(ln_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dxhat, g_dtype, x, stats, dskip, (long)M, C, dx);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_act_fwd(const void* pre, int in_dtype, void* post, int out_dtype, int64_t n, int act, void* stream) {
  if (!pre || !post || n <= 0) TANTE_FAIL(-1, "tante_act_fwd: bad argument");
  if (n % 4 == 0 && (((uintptr_t)pre | (uintptr_t)post) % 16) == 0)
    hipLaunchKernelGGL(act_fwd_vec_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, in_dtype, post, out_dtype,
                       (long)(n / 4), act);
  else
    hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, in_dtype, post, out_dtype, (long)n, act);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_act_bwd(const void* dpost, int d_dtype, const void* pre, int pre_dtype, void* dpre, int out_dtype, int64_t n, int act,
                             void* stream) {
  if (!dpost || !pre || !dpre || n <= 0) TANTE_FAIL(-1, "tante_act_bwd: bad argument");
  if (n % 4 == 0 && (((uintptr_t)dpost | (uintptr_t)pre | (uintptr_t)dpre) % 16) == 0)
    hipLaunchKernelGGL(act_bwd_vec_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dpost, d_dtype, pre, pre_dtype,
                       dpre, out_dtype, (long)(n / 4), act);
  else
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dpost, d_dtype, pre, pre_dtype, dpre,
                       out_dtype, (long)n, act);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int64_t tante_axis_wgrad_workspace_bytes(void) { return AW_WS_FLOATS * (int64_t)sizeof(float); }

extern "C" int tante_axis_wgrad_ws(const float* U, const float* V, int64_t outer, int n, int64_t inner, float* dW, float* db, int accumulate,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
  if (!U || !V || !dW || outer <= 0 || n <= 0 || inner <= 0) TANTE_FAIL(-1, "tante_axis_wgrad: bad argument");
  if (n > 64) TANTE_FAIL(-2, "tante_axis_wgrad: axis length %d > 64", n);
  if (inner % 16 || (((uintptr_t)U | (uintptr_t)V) & 15)) TANTE_FAIL(-2, "tante_axis_wgrad: inner must be a multiple of 16 and the operands 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  float* ws = (workspace && workspace_bytes >= tante_axis_wgrad_workspace_bytes() && ((uintptr_t)workspace % 16) == 0) ? (float*)workspace : nullptr;
  if (!accumulate) {
    if (tante_zero_async(dW, (size_t)n * n * sizeof(float), s) != hipSuccess) TANTE_FAIL(-3, "tante_axis_wgrad: memset failed");
    if (db && tante_zero_async(db, (size_t)n * sizeof(float), s) != hipSuccess) TANTE_FAIL(-3, "tante_axis_wgrad: memset failed");
  }
  // short axes: G segments of the contraction per MFMA tile (inner must hold whole 16 G chunks)
  const int g = (n <= 4 && inner % 64 == 0) ? 4 : (n <= 8 && inner % 32 == 0) ? 2 : 1;
  const long n_chunks = (long)outer * (inner / (16 * g));
  // long axes (n > 16: partials of up to 16 KB): 8 waves per workgroup and one workgroup per CU -- the epilogue (partial store, group sum,
  // atomics) is per workgroup and was what the kernel's time followed (n = 48: 39 / 66 / 85 us at 256 / 512 / 683 workgroups of 4 waves)
  const int nthr = n > 16 ? 512 : 256;
  long wgs = (n_chunks + 3 * (nthr / 64) - 1) / (3 * (nthr / 64));          // at least three chunks per wave
  const long wg_cap = tante_opt("TANTE_AXIS_WGRAD_WGS", 0);
  const long cap = wg_cap ? (wg_cap > AW_MAXWG && ws ? AW_MAXWG : wg_cap) : (n > 16 ? 256 : 512);
  if (wgs > cap) wgs = cap;
  if (wgs < 1) wgs = 1;
#define TANTE_AW(NTV, GV) hipLaunchKernelGGL((axis_wgrad_kernel<NTV, GV>), dim3((unsigned)wgs), dim3(nthr), 0, s, U, V, (long)outer, n, (long)inner, dW, db, n_chunks, ws)
  switch ((n + 15) / 16) {
    case 1: if (g == 4) TANTE_AW(1, 4); else if (g == 2) TANTE_AW(1, 2); else TANTE_AW(1, 1); break;
    case 2: TANTE_AW(2, 1); break;
    case 3: TANTE_AW(3, 1); break;
    default: TANTE_AW(4, 1); break;
  }
#undef TANTE_AW
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_axis_wgrad(const float* U, const float* V, int64_t outer, int n, int64_t inner, float* dW, float* db, int accumulate,
                                void* stream) {
  return tante_axis_wgrad_ws(U, V, outer, n, inner, dW, db, accumulate, nullptr, 0, stream);
}
extern "C" int tante_fold_fwd(const float* W, const float* b, const float* gamma, const float* beta, int N, int K, float* We, float* be,
                              void* stream) {
  if (!W || !gamma || !beta || !We || !be || N <= 0 || K <= 0 || K % 4) TANTE_FAIL(-1, "tante_fold_fwd: bad argument (K must be a multiple of 4)");
  const int nbe = (int)(((long)N * K / 4 + 255) / 256);
  hipLaunchKernelGGL(fold_fwd_kernel, dim3(nbe + (N + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, b, gamma, beta, N, K, We, be, nbe);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_fold_fwd_multi(const TanteFoldFwd* folds, int n, void* stream) {
  if (!folds || n <= 0) TANTE_FAIL(-1, "tante_fold_fwd_multi: bad argument");
  for (int g = 0; g < n; g += FOLDF_MAX) {
    FoldFwdBatch B;
    const int m = n - g < FOLDF_MAX ? n - g : FOLDF_MAX;
    int blocks = 0;
    for (int e = 0; e < m; ++e) {
      const TanteFoldFwd& f = folds[g + e];
      if (!f.W || !f.gamma || !f.beta || !f.We || !f.be || f.N <= 0 || f.K <= 0 || f.K % 4) TANTE_FAIL(-1, "tante_fold_fwd_multi: bad entry %d", g + e);
      B.W[e] = f.W; B.b[e] = f.b; B.gamma[e] = f.gamma; B.beta[e] = f.beta; B.We[e] = f.We; B.be[e] = f.be;
      B.N[e] = f.N; B.K[e] = f.K; B.nbe[e] = (int)(((long)f.N * f.K / 4 + 255) / 256); B.first[e] = blocks;
      blocks += B.nbe[e] + (f.N + 3) / 4;
    }
    B.first[m] = blocks;
    B.n = m;
    hipLaunchKernelGGL(fold_fwd_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, B);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}
static int fold_bwd_impl(float* GW, float* Gb, const float* W, const float* gamma, const float* beta, int N, int K, float* dW, float* db,
                         float* dgamma, float* dbeta, int clear, void* stream) {
  if (!GW || !Gb || !W || !gamma || !beta || !dW || !dgamma || !dbeta || N <= 0 || K <= 0 || K % 4)
    TANTE_FAIL(-1, "tante_fold_bwd: bad argument (K must be a multiple of 4)");
  const int kb = (K + 255) / 256;
  hipLaunchKernelGGL(fold_bwd_kernel, dim3(((N + FB_ROWS - 1) / FB_ROWS) * kb), dim3(256), 0, (hipStream_t)stream, GW, Gb, W, gamma, beta, N, K, dW,
                     db, dgamma, dbeta, clear);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_fold_bwd(const float* GW, const float* Gb, const float* W, const float* gamma, const float* beta, int N, int K, float* dW,
                              float* db, float* dgamma, float* dbeta, void* stream) {
  return fold_bwd_impl((float*)GW, (float*)Gb, W, gamma, beta, N, K, dW, db, dgamma, dbeta, 0, stream);
}
extern "C" int tante_fold_bwd_clear(float* GW, float* Gb, const float* W, const float* gamma, const float* beta, int N, int K, float* dW, float* db,
                                    float* dgamma, float* dbeta, void* stream) {
  if (K > 256) TANTE_FAIL(-2, "tante_fold_bwd_clear: K <= 256 (one column chunk reads the bias accumulator)");
  return fold_bwd_impl(GW, Gb, W, gamma, beta, N, K, dW, db, dgamma, dbeta, 1, stream);
}
extern "C" int tante_fold_bwd_multi(const TanteFold* folds, int n, int clear, void* stream) {
  if (!folds || n <= 0) TANTE_FAIL(-1, "tante_fold_bwd_multi: bad argument");
  for (int g = 0; g < n; g += FOLD_MAX) {
    FoldBatch B;
    const int m = n - g < FOLD_MAX ? n - g : FOLD_MAX;
    int blocks = 0;
    for (int e = 0; e < m; ++e) {
      const TanteFold& f = folds[g + e];
      if (!f.GW || !f.Gb || !f.W || !f.gamma || !f.beta || !f.dW || !f.dgamma || !f.dbeta || f.N <= 0 || f.K <= 0 || f.K % 4)
        TANTE_FAIL(-1, "tante_fold_bwd_multi: bad entry %d (K must be a multiple of 4)", g + e);
      if (clear && f.K > 256) TANTE_FAIL(-2, "tante_fold_bwd_multi: clear needs K <= 256 (one column chunk reads the bias accumulator)");
      B.GW[e] = f.GW; B.Gb[e] = f.Gb; B.W[e] = f.W; B.gamma[e] = f.gamma; B.beta[e] = f.beta;
      B.dW[e] = f.dW; B.db[e] = f.db; B.dgamma[e] = f.dgamma; B.dbeta[e] = f.dbeta;
      B.N[e] = f.N; B.K[e] = f.K; B.first[e] = blocks;
      blocks += ((f.N + FB_ROWS - 1) / FB_ROWS) * ((f.K + 255) / 256);
    }
    B.first[m] = blocks;
    B.n = m;
    hipLaunchKernelGGL(fold_bwd_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, B, clear);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_colsum(const void* x, int dtype, int64_t outer, int C, int64_t inner, float* out, int accumulate, void* stream) {
  if (!x || !out || outer <= 0 || C <= 0 || inner <= 0) TANTE_FAIL(-1, "tante_colsum: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && tante_zero_async(out, (size_t)C * sizeof(float), s) != hipSuccess) TANTE_FAIL(-3, "tante_colsum: memset failed");
  if (inner == 1 && dtype == TANTE_BF16 && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0 && ((uintptr_t)x % 16) == 0) {
    long chunks = 1024;                                      // at least 512 rows per workgroup: every workgroup ends in C same-address atomics (9.4 MB of 256-channel rows: 17.6 / 11.5 / 9.7 / 10.7 / 15.1 us at 128 / 256 / 512 / 1024 / 2048 rows)
    const long cs_rows = tante_opt("TANTE_COLSUM_ROWS", 512);
    if (chunks > (outer + cs_rows - 1) / cs_rows) chunks = (outer + cs_rows - 1) / cs_rows;
    const long chunk = (outer + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_bf16_vec_kernel, dim3((unsigned)((outer + chunk - 1) / chunk)), dim3(256), 0, s, (const unsigned short*)x, (long)outer, C,
                       chunk, out);
  } else if (inner == 1) {
    // enough workgroups to fill the chip: (row chunks) x (64-channel groups) ~ 2048, at least 64 rows per chunk
    const long groups = (C + 63) / 64;
    long chunks = 2048 / groups;
    if (chunks < 1) chunks = 1;
    if (chunks > (outer + 63) / 64) chunks = (outer + 63) / 64;
    const long chunk = (outer + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((outer + chunk - 1) / chunk), (unsigned)((C + 63) / 64)), dim3(256), 0, s, x, dtype,
                       (long)outer, C, (long)inner, chunk, out);
  } else {
    const long chunk = 4096;
    const long nch = (inner + chunk - 1) / chunk;
    if (outer * nch > 2000000000L) TANTE_FAIL(-2, "tante_colsum: too large");
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)(outer * nch), (unsigned)C), dim3(256), 0, s, x, dtype, (long)outer, C, (long)inner, chunk, out);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_film_pos_fwd(const float* v, const float* a, const float* b, const float* s_emb, int64_t rows, int C, int T, int64_t HW,
                                  float* y, void* stream) {
  if (!v || !a || !b || !s_emb || !y || rows <= 0 || C <= 0 || C % 4 || T <= 0 || HW <= 0) TANTE_FAIL(-1, "tante_film_pos_fwd: bad argument");
  const long n4 = rows * (C / 4);
  hipLaunchKernelGGL(film_pos_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, a, b, s_emb, (long)rows, C / 4,
                     T, (long)HW, y);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_pos_embed_tmajor(const float* v, const float* t_emb, const float* s_emb, int64_t B, int T, int64_t HW, int C, float* y,
                                      void* stream) {
  if (!v || !t_emb || !s_emb || !y || B <= 0 || C <= 0 || C % 4 || T <= 0 || HW <= 0) TANTE_FAIL(-1, "tante_pos_embed_tmajor: bad argument");
  const long rows = (long)B * T * HW, n4 = rows * (C / 4);
  hipLaunchKernelGGL(pos_embed_tmajor_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, t_emb, s_emb, rows, C / 4, T,
                     (long)HW, y);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_film_pos_bwd(const float* dy, const float* v, const float* a, int64_t BT, int64_t HW, int C, int T, float* dv, float* da,
                                  float* db, float* ds, void* stream) {
  if (!dy || !v || !a || !dv || !da || !db || !ds || BT <= 0 || HW <= 0 || C <= 0 || T <= 0) TANTE_FAIL(-1, "tante_film_pos_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (tante_zero_async(da, (size_t)T * C * sizeof(float), s) != hipSuccess || tante_zero_async(db, (size_t)T * C * sizeof(float), s) != hipSuccess)
    TANTE_FAIL(-3, "tante_film_pos_bwd: memset failed");
  const long chunk = 64;
  if (C == 256 && ((((uintptr_t)dy | (uintptr_t)v | (uintptr_t)a | (uintptr_t)dv) & 15) == 0))
    hipLaunchKernelGGL((film_pos_bwd256_kernel<false, 64>), dim3((unsigned)((HW + 63) / 64), (unsigned)BT), dim3(256), 0, s, dy, v, a, (long)HW, T, dv, da, db,
                       FilmFrames{});
  else
    hipLaunchKernelGGL(film_pos_bwd_kernel, dim3((unsigned)((HW + chunk - 1) / chunk), (unsigned)BT), dim3(256), 0, s, dy, v, a, (long)HW, C, T, chunk,
                       dv, da, db);
  hipLaunchKernelGGL(film_pos_ds_kernel, dim3((unsigned)((HW * C + 255) / 256)), dim3(256), 0, s, dy, (long)BT, (long)HW, C, ds);
  TANTE_CHECK_LAUNCH();
  return 0;
}
static int film_frames_arg(const TanteFrames* fr, int T, int C, float* const* dv, FilmFrames& F) {
  if (!fr || T <= 0 || T > 8) return -1;
  F.acc = 0;
  for (int t = 0; t < 8; ++t) {
    F.f[t] = t < T ? fr->f[t] : nullptr;
    F.bstride[t] = t < T ? (long)fr->bstride[t] : 0;
    F.dv[t] = (dv && t < T) ? dv[t] : nullptr;
    if (t < T && (!F.f[t] || ((uintptr_t)F.f[t] & 15) || F.bstride[t] % 4 || (dv && (!F.dv[t] || ((uintptr_t)F.dv[t] & 15))))) return -1;
  }
  return 0;
}
extern "C" int tante_film_pos_fwd_frames(const TanteFrames* frames, const float* a, const float* b, const float* s_emb, int64_t B, int T, int64_t HW,
                                         int C, float* y, void* stream) {
  FilmFrames F;
  if (!a || !b || !s_emb || !y || B <= 0 || HW <= 0 || C <= 0 || C % 4 || film_frames_arg(frames, T, C, nullptr, F))
    TANTE_FAIL(-1, "tante_film_pos_fwd_frames: bad argument (1 <= T <= 8, C %% 4 == 0, 16-byte aligned frames)");
  const long rows = B * T * HW, n4 = rows * (C / 4);
  hipLaunchKernelGGL(film_pos_fwd_frames_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, F, a, b, s_emb, rows, C / 4, T,
                     (long)HW, y);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_film_pos_bwd_frames(const float* dy, const TanteFrames* frames, const float* a, int64_t B, int64_t HW, int C, int T,
                                         float* const* dv, float* da, float* db, float* ds, void* stream) {
  FilmFrames F;
  if (!dy || !a || !dv || !da || !db || !ds || B <= 0 || HW <= 0 || film_frames_arg(frames, T, C, dv, F))
    TANTE_FAIL(-1, "tante_film_pos_bwd_frames: bad argument");
  if (C != 256 || ((uintptr_t)dy & 15)) TANTE_FAIL(-2, "tante_film_pos_bwd_frames: C = 256 only (got %d)", C);
  hipStream_t s = (hipStream_t)stream;
  if (tante_zero_async(da, (size_t)T * C * sizeof(float), s) != hipSuccess || tante_zero_async(db, (size_t)T * C * sizeof(float), s) != hipSuccess)
    TANTE_FAIL(-3, "tante_film_pos_bwd_frames: clear failed");
  const long BT = B * T;
  const int rows_env = tante_opt("TANTE_FILM_BWD_ROWS", 64);      // 64 rows per workgroup: 20.7 us against 30.2 at 32 (every workgroup ends in 512 same-address atomics)
  if (rows_env != 16 && rows_env != 32)
    hipLaunchKernelGGL((film_pos_bwd256_kernel<true, 64>), dim3((unsigned)((HW + 63) / 64), (unsigned)BT), dim3(256), 0, s, dy, (const float*)nullptr, a, (long)HW, T,
                       (float*)nullptr, da, db, F);
  else if (rows_env == 16)
    hipLaunchKernelGGL((film_pos_bwd256_kernel<true, 16>), dim3((unsigned)((HW + 15) / 16), (unsigned)BT), dim3(256), 0, s, dy, (const float*)nullptr, a, (long)HW, T,
                       (float*)nullptr, da, db, F);
  else
    hipLaunchKernelGGL((film_pos_bwd256_kernel<true, 32>), dim3((unsigned)((HW + 31) / 32), (unsigned)BT), dim3(256), 0, s, dy, (const float*)nullptr, a, (long)HW, T,
                       (float*)nullptr, da, db, F);
  hipLaunchKernelGGL(film_pos_ds_kernel, dim3((unsigned)((HW * C + 255) / 256)), dim3(256), 0, s, dy, BT, (long)HW, C, ds);
  TANTE_CHECK_LAUNCH();
  return 0;
}
/* tante_film_pos_bwd_frames with accumulation: bit t of dv_acc_mask -- dv[t] is ADDED to (a frame encoding that sits in several windows of
 * a BPTT rollout receives one gradient per window: the uses accumulate in place instead of autograd summing fresh tensors);
 * acc_flags bit 0: da / db are added to (the FiLM tables are shared by every call of the rollout), bit 1: ds is added to. */
extern "C" int tante_film_pos_bwd_frames_acc(const float* dy, const TanteFrames* frames, const float* a, int64_t B, int64_t HW, int C, int T,
                                             float* const* dv, int dv_acc_mask, float* da, float* db, float* ds, int acc_flags, void* stream) {
  FilmFrames F;
  if (!dy || !a || !dv || !da || !db || !ds || B <= 0 || HW <= 0 || film_frames_arg(frames, T, C, dv, F))
    TANTE_FAIL(-1, "tante_film_pos_bwd_frames_acc: bad argument");
  if (C != 256 || ((uintptr_t)dy & 15)) TANTE_FAIL(-2, "tante_film_pos_bwd_frames_acc: C = 256 only (got %d)", C);
  F.acc = dv_acc_mask;
  hipStream_t s = (hipStream_t)stream;
  if (!(acc_flags & 1) && (tante_zero_async(da, (size_t)T * C * sizeof(float), s) != hipSuccess || tante_zero_async(db, (size_t)T * C * sizeof(float), s) != hipSuccess))
    TANTE_FAIL(-3, "tante_film_pos_bwd_frames_acc: clear failed");
  const long BT = B * T;
  hipLaunchKernelGGL((film_pos_bwd256_kernel<true, 64>), dim3((unsigned)((HW + 63) / 64), (unsigned)BT), dim3(256), 0, s, dy, (const float*)nullptr, a, (long)HW, T,
                     (float*)nullptr, da, db, F);
  hipLaunchKernelGGL(film_pos_ds_kernel, dim3((unsigned)((HW * C + 255) / 256)), dim3(256), 0, s, dy, BT, (long)HW, C, ds, (acc_flags >> 1) & 1);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_taylor_bwd(const float* dout, int64_t dout_bstride, float* const* dderivs, int n_order, double dt, int n_out, float* dlast,
                                int64_t dlast_bstride, int accumulate, int64_t B, int64_t frame, void* stream) {
  if (!dout || !dderivs || !dlast) TANTE_FAIL(-1, "tante_taylor_bwd: null pointer");
  if (n_order < 1 || n_order > 8 || n_out < 1 || n_out > 8) TANTE_FAIL(-2, "tante_taylor_bwd: order and n_out must be in 1..8");
  if (frame % 4 || dout_bstride % 4 || dlast_bstride % 4) TANTE_FAIL(-2, "tante_taylor_bwd: sizes must be multiples of 4 floats");
  TaylorBArgs ta;
  for (int k = 0; k < 8; ++k) ta.dd[k] = k < n_order ? dderivs[k] : nullptr;
  for (int i = 0; i < 8; ++i) {
    double fact = 1.0;
    for (int k = 0; k < 8; ++k) {
      fact *= (double)(k + 1);
      double p = 1.0;
      for (int e = 0; e <= k; ++e) p *= (double)(i + 1) * dt;
      ta.coef[i][k] = (float)(p / fact);
    }
  }
  const long n4 = B * (frame / 4);
  const dim3 grid((unsigned)((n4 + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
#define TANTE_TB(NO) \
  case NO: hipLaunchKernelGGL(taylor_bwd_kernel<NO>, grid, dim3(256), 0, s, dout, (long)dout_bstride, ta, n_out, dlast, (long)dlast_bstride, accumulate, (long)B, (long)(frame / 4)); break;
  switch (n_order) { TANTE_TB(1) TANTE_TB(2) TANTE_TB(3) TANTE_TB(4) TANTE_TB(5) TANTE_TB(6) TANTE_TB(7) TANTE_TB(8) }
#undef TANTE_TB
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_attention_bwd(const void* qkv, const void* dO, void* dqkv, int dtype, int C, int n_head, const TanteSeq* seq, int causal,
                                   float p_drop, uint64_t seed_, void* stream) {
  const unsigned long long seed = (unsigned long long)seed_;
  if (!qkv || !dO || !dqkv || !seq) TANTE_FAIL(-1, "tante_attention_bwd: null pointer");
  if (n_head <= 0 || C % n_head) TANTE_FAIL(-1, "tante_attention_bwd: bad heads");
  if (seq->L > 128) TANTE_FAIL(-2, "tante_attention_bwd: sequences longer than 128 are not on the train path yet (L=%d)", seq->L);
  hipStream_t s = (hipStream_t)stream;
  if (try_attn_bwd_mfma(qkv, dO, dqkv, dtype, C, n_head, *seq, causal, p_drop, seed, s)) {
    TANTE_CHECK_LAUNCH();
    return 0;
  }
  switch (C / n_head) {
    case 4: launch_attn_bwd<4>(qkv, dO, dqkv, dtype, C, n_head, *seq, causal, p_drop, seed, s); break;
    case 8: launch_attn_bwd<8>(qkv, dO, dqkv, dtype, C, n_head, *seq, causal, p_drop, seed, s); break;
    case 16: launch_attn_bwd<16>(qkv, dO, dqkv, dtype, C, n_head, *seq, causal, p_drop, seed, s); break;
    case 32: launch_attn_bwd<32>(qkv, dO, dqkv, dtype, C, n_head, *seq, causal, p_drop, seed, s); break;
    case 64: launch_attn_bwd<64>(qkv, dO, dqkv, dtype, C, n_head, *seq, causal, p_drop, seed, s); break;
    default: TANTE_FAIL(-2, "tante_attention_bwd: head dim %d unsupported", C / n_head);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_axis_mlp_bwd(const float* x, const float* dy, int64_t outer, int n, int64_t inner, const float* w1, const float* b1,
                                  const float* w2, float* dx, float* h, float* dpre, void* stream) {
  if (!x || !dy || !w1 || !b1 || !w2 || !dx || !h || !dpre || outer <= 0 || n <= 0 || inner <= 0) TANTE_FAIL(-1, "tante_axis_mlp_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (n <= 4) launch_axis_bwd<4>(x, dy, outer, n, inner, w1, b1, w2, dx, h, dpre, s);
  else if (n <= 8) launch_axis_bwd<8>(x, dy, outer, n, inner, w1, b1, w2, dx, h, dpre, s);
  else if (n <= 16) launch_axis_bwd<16>(x, dy, outer, n, inner, w1, b1, w2, dx, h, dpre, s);
  else if (n <= 32) launch_axis_bwd<32>(x, dy, outer, n, inner, w1, b1, w2, dx, h, dpre, s);
  else if (n <= 48) launch_axis_bwd<48>(x, dy, outer, n, inner, w1, b1, w2, dx, h, dpre, s);
  else if (n <= 64) launch_axis_bwd<64>(x, dy, outer, n, inner, w1, b1, w2, dx, h, dpre, s);
  else TANTE_FAIL(-2, "tante_axis_mlp_bwd: axis length %d > 64 is not on the train path yet", n);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_dropout_add(const void* y, int y_dtype, const float* res, float p, uint64_t seed, int64_t n, float* out, void* stream) {
  if (!y || !res || !out || n <= 0 || p < 0.0f || p >= 1.0f) TANTE_FAIL(-1, "tante_dropout_add: bad argument");
  if (n % 4 == 0 && (((uintptr_t)y | (uintptr_t)res | (uintptr_t)out) % 16) == 0)
    hipLaunchKernelGGL(dropout_add_vec_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, y_dtype, res, p,
                       (unsigned long long)seed, (long)(n / 4), out);
  else
    hipLaunchKernelGGL(dropout_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, y_dtype, res, p,
                       (unsigned long long)seed, (long)n, out);
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_dropout_bwd(const float* dout, float p, uint64_t seed, int64_t n, void* dy, int y_dtype, void* stream) {
  if (!dout || !dy || n <= 0 || p < 0.0f || p >= 1.0f) TANTE_FAIL(-1, "tante_dropout_bwd: bad argument");
  if (n % 4 == 0 && (((uintptr_t)dout | (uintptr_t)dy) % 16) == 0)
    hipLaunchKernelGGL(dropout_bwd_vec_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, p,
                       (unsigned long long)seed, (long)(n / 4), dy, y_dtype);
  else
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, p, (unsigned long long)seed,
                       (long)n, dy, y_dtype);
  TANTE_CHECK_LAUNCH();
  return 0;
}
