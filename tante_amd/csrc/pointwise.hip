// HBM-bound stages of the TANTE path: axis propagators, FiLM tables, Taylor sum, step-size reduction.
#include "common.hip.h"
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <algorithm>

int tante_axis_mlp_mfma_supported(int64_t outer, int n, int64_t inner, const void* x, const void* w1, const void* w2);      // axis_mfma.hip
int tante_axis_mlp_mfma(float* x, int64_t outer, int n, int64_t inner, const float* w1, const float* b1, const float* w2, const float* b2, hipStream_t s);

namespace {

// ---- axis propagator:  x += W2 gelu_erf(W1 x_line + b1) + b2  along a strided axis -------------------
// x is viewed as (outer, n, inner) fp32; one lane owns one (outer, inner) column, i.e. a line of n
// values with stride `inner`; consecutive lanes own consecutive inner indices, so every load/store
// instruction of a wave covers 256 contiguous bytes.  The line, the hidden vector and both n x n
// weight matrices are small: the line lives in registers (N is a template parameter so the register
// arrays are statically indexed) and the weights are read with wave-uniform addresses (scalar loads).
template <int N>
__global__ __launch_bounds__(256) void axis_mlp_kernel(float* __restrict__ x, long outer, int n, long inner,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2) {
  // weights zero-padded to N x N in LDS; w2 stored transposed so that both inner loops read one
  // contiguous row per hidden unit j with a wave-uniform address (ds_read_b128 broadcast)
  __shared__ __attribute__((aligned(16))) float w1s[N * N];
  __shared__ __attribute__((aligned(16))) float w2ts[N * N];
  __shared__ __attribute__((aligned(16))) float b1s[N];
  __shared__ __attribute__((aligned(16))) float b2s[N];
  for (int idx = threadIdx.x; idx < N * N; idx += 256) {
    const int j = idx / N, a = idx % N;
    const bool in = (j < n) && (a < n);
    w1s[idx] = in ? w1[j * n + a] : 0.0f;
    w2ts[idx] = in ? w2[a * n + j] : 0.0f;
  }
  if (threadIdx.x < N) {
    b1s[threadIdx.x] = (threadIdx.x < n) ? b1[threadIdx.x] : 0.0f;
    b2s[threadIdx.x] = (threadIdx.x < n) ? b2[threadIdx.x] : 0.0f;
  }
  __syncthreads();
  const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= outer * inner) return;
  const long o = col / inner, i = col - o * inner;
  float* p = x + o * (long)n * inner + i;
  float v[N], acc[N];
#pragma unroll
  for (int a = 0; a < N; ++a) {
    v[a] = (a < n) ? p[(long)a * inner] : 0.0f;
    acc[a] = b2s[a];
  }
  for (int j = 0; j < n; ++j) {
    float s = b1s[j];
#pragma unroll
    for (int a4 = 0; a4 < N / 4; ++a4) {
      const f32x4 w = *(const f32x4*)(w1s + j * N + 4 * a4);
      s += w[0] * v[4 * a4] + w[1] * v[4 * a4 + 1] + w[2] * v[4 * a4 + 2] + w[3] * v[4 * a4 + 3];
    }
    const float h = gelu_erf_f(s);
#pragma unroll
    for (int a4 = 0; a4 < N / 4; ++a4) {
      const f32x4 w = *(const f32x4*)(w2ts + j * N + 4 * a4);
      acc[4 * a4] += w[0] * h;
      acc[4 * a4 + 1] += w[1] * h;
      acc[4 * a4 + 2] += w[2] * h;
      acc[4 * a4 + 3] += w[3] * h;
    }
  }
#pragma unroll
  for (int a = 0; a < N; ++a)
    if (a < n) p[(long)a * inner] = v[a] + acc[a];
}

// Short axes (n <= 8, i.e. the temporal propagator): four adjacent columns per thread as float4 (16-byte loads / stores along the
// contiguous inner axis), weights in registers via LDS broadcast.  FAST: polynomial GELU (bf16 compute mode), else erff.
template <int N, bool FAST>
__global__ __launch_bounds__(256) void axis_mlp_vec_kernel(float* __restrict__ x, long outer, int n, long inner4,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           const float* __restrict__ src = nullptr) {   // src: out of place (x = result only)
  __shared__ float w1s[N * N], w2s[N * N], b1s[N], b2s[N];
  for (int idx = threadIdx.x; idx < N * N; idx += 256) {
    const int j = idx / N, a = idx % N;
    const bool in = (j < n) && (a < n);
    w1s[idx] = in ? w1[j * n + a] : 0.0f;
    w2s[idx] = in ? w2[j * n + a] : 0.0f;
  }
  if (threadIdx.x < N) {
    b1s[threadIdx.x] = (threadIdx.x < n) ? b1[threadIdx.x] : 0.0f;
    b2s[threadIdx.x] = (threadIdx.x < n) ? b2[threadIdx.x] : 0.0f;
  }
  __syncthreads();
  const long col = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= outer * inner4) return;
  const long o = col / inner4, i = col - o * inner4;
  f32x4* p = (f32x4*)x + o * (long)n * inner4 + i;
  const f32x4* q = src ? (const f32x4*)src + o * (long)n * inner4 + i : p;
  f32x4 v[N], h[N];
#pragma unroll
  for (int a = 0; a < N; ++a) v[a] = (a < n) ? q[(long)a * inner4] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < N; ++j) {
    f32x4 s = splat4(b1s[j]);
#pragma unroll
    for (int a = 0; a < N; ++a) s += splat4(w1s[j * N + a]) * v[a];
    if constexpr (FAST) h[j] = gelu_poly4<false>(s);
    else h[j] = f32x4{gelu_erf_f(s[0]), gelu_erf_f(s[1]), gelu_erf_f(s[2]), gelu_erf_f(s[3])};
  }
#pragma unroll
  for (int a = 0; a < N; ++a) {
    if (a < n) {
      f32x4 acc = v[a] + splat4(b2s[a]);
#pragma unroll
      for (int j = 0; j < N; ++j) acc += splat4(w2s[a * N + j]) * h[j];
      p[(long)a * inner4] = acc;
    }
  }
}

// generic fallback for long axes (n > 64): line and hidden vector live in LDS, [n][64 lanes]
__global__ __launch_bounds__(64) void axis_mlp_lds_kernel(float* __restrict__ x, long outer, int n, long inner,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2) {
  extern __shared__ float sm[];  // v[n][64], h[n][64]
  float* vs = sm;
  float* hs = sm + (long)n * 64;
  const int lane = threadIdx.x;
  const long col = (long)blockIdx.x * 64 + lane;
  const bool live = col < outer * inner;
  const long o = live ? col / inner : 0, i = live ? col - o * inner : 0;
  float* p = x + o * (long)n * inner + i;
  for (int a = 0; a < n; ++a) vs[a * 64 + lane] = live ? p[(long)a * inner] : 0.0f;
  for (int j = 0; j < n; ++j) {
    float s = b1[j];
    for (int a = 0; a < n; ++a) s += w1[j * n + a] * vs[a * 64 + lane];
    hs[j * 64 + lane] = gelu_erf_f(s);
  }
  if (!live) return;
  for (int a = 0; a < n; ++a) {
    float s = b2[a];
    for (int j = 0; j < n; ++j) s += w2[a * n + j] * hs[j * 64 + lane];
    p[(long)a * inner] = vs[a * 64 + lane] + s;
  }
}

// ---- fused vertical + horizontal propagators on MFMA -------------------------------------------------------
// x (BT, nH, nW, C) fp32, in place:  x += MLP_H(x) along h, then x += MLP_W(x) along w  (attn_backbone.py:140-143).
// One workgroup owns one (bt, 16-channel tile): the nH x nW x 16 plane sits in LDS (64 B per token, so global
// traffic is whole 64-byte segments), both axes are processed there and the plane is read and written ONCE.
// The n x n contractions run on the matrix cores: for a fixed w (phase H) or h (phase W) the MFMA B operand is
// the 16 channels x n line, D = W1 . line, GELU on the accumulators, and -- as in the fused block kernel -- the
// accumulator tiles are re-used directly as the next MFMA's B operand with the k order permuted (W2's columns
// are loaded in that order), so the hidden layer never leaves registers.
// bf16 compute: v_mfma_f32_16x16x32_bf16; fp32 compute: v_mfma_f32_16x16x4_f32 (exact fp32).  n <= 64.
template <bool BF16, int MT>
struct AxisW {  // A-operand fragments of one n x n weight (n <= 16 * MT), zero padded
  static constexpr int KB = BF16 ? (MT + 1) / 2 : MT;   // k-blocks: 32 wide (bf16) / 16 wide (fp32)
  u32x4 f[MT][KB];
};

// the staged n x n weight: bf16 in the bf16 path (it is rounded to bf16 for the MFMA anyway; halves the staging LDS), fp32 otherwise
template <bool BF16> struct AxisStage { using T = float; };
template <> struct AxisStage<true> { using T = unsigned short; };
template <bool BF16>
__device__ __forceinline__ float axis_wld(const typename AxisStage<BF16>::T* ws, int i) {
  if constexpr (BF16) return __uint_as_float(((unsigned)ws[i]) << 16);
  else return ws[i];
}
template <bool BF16>
__device__ __forceinline__ void axis_wst(typename AxisStage<BF16>::T* ws, int i, float v) {
  if constexpr (BF16) { __bf16 b = (__bf16)v; ws[i] = __builtin_bit_cast(unsigned short, b); }
  else ws[i] = v;
}

// row stride of the staged n x n weight (elements): bf16 rows are padded by 8 so that the 16-byte fragment reads of 8 consecutive rows
// land in 8 different bank groups (a 64-byte row stride made them 4-way conflicted: PMC showed 64 % of this kernel's LDS cycles as conflicts)
template <bool BF16>
__host__ __device__ __forceinline__ int axis_wstride(int n) { return BF16 ? n + 8 : n; }

// build the fragments from the weight staged in LDS (ws: n rows of axis_wstride(n), row-major)
template <bool BF16, int MT, bool KPERM>
__device__ __forceinline__ void load_axis_weight(const typename AxisStage<BF16>::T* ws, int n, int l15, int kk, AxisW<BF16, MT>& aw) {
  const int NS = axis_wstride<BF16>(n);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 16 + l15;
#pragma unroll
    for (int kb = 0; kb < AxisW<BF16, MT>::KB; ++kb) {
      if constexpr (BF16) {
        if (n % 8 == 0) {     // whole 16-byte (8-byte when k-permuted) pieces of a row: vector reads
          u32x4 f = u32x4{0u, 0u, 0u, 0u};
          if constexpr (!KPERM) {
            const int k0 = kb * 32 + 8 * kk;
            if (row < n && k0 + 8 <= n) f = *(const u32x4*)(ws + row * NS + k0);
          } else {
            const int ka = kb * 32 + 4 * kk, kb2 = kb * 32 + 16 + 4 * kk;
            if (row < n && ka + 4 <= n) { const u32x2 a = *(const u32x2*)(ws + row * NS + ka); f[0] = a[0]; f[1] = a[1]; }
            if (row < n && kb2 + 4 <= n) { const u32x2 b = *(const u32x2*)(ws + row * NS + kb2); f[2] = b[0]; f[3] = b[1]; }
          }
          aw.f[mt][kb] = f;
          continue;
        }
      }
      float v[8];
#pragma unroll
      for (int e = 0; e < (BF16 ? 8 : 4); ++e) {
        int k;
        if constexpr (BF16) k = KPERM ? (kb * 32 + (e < 4 ? 4 * kk + e : 16 + 4 * kk + (e - 4))) : (kb * 32 + 8 * kk + e);
        else k = KPERM ? (kb * 16 + 4 * kk + e) : (kb * 16 + 4 * e + kk);
        v[e] = (row < n && k < n) ? axis_wld<BF16>(ws, row * NS + k) : 0.0f;
      }
      if constexpr (BF16) {
        aw.f[mt][kb][0] = pack_bf16x2(v[0], v[1]); aw.f[mt][kb][1] = pack_bf16x2(v[2], v[3]);
        aw.f[mt][kb][2] = pack_bf16x2(v[4], v[5]); aw.f[mt][kb][3] = pack_bf16x2(v[6], v[7]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) aw.f[mt][kb][e] = __float_as_uint(v[e]);
      }
    }
  }
}

// one line group: xin = this lane's k elements of the line (operand order), out = W2 gelu(W1 x + b1) + b2
template <bool BF16, int MT>
__device__ __forceinline__ void axis_mlp_mfma(const AxisW<BF16, MT>& w1, const AxisW<BF16, MT>& w2, const float (&b1)[MT][4],
                                              const float (&b2)[MT][4], const float (&xin)[16 * ((MT + 1) / 2)], f32x4 (&out)[MT]) {
  constexpr int KB = AxisW<BF16, MT>::KB;
  f32x4 d1[MT + 1];  // +1: the bf16 pack below reads tile MT when MT is odd (zeros)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) d1[mt] = f32x4{b1[mt][0], b1[mt][1], b1[mt][2], b1[mt][3]};
  d1[MT] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (BF16) {
    u32x4 xb[KB];
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) {
      xb[ks][0] = pack_bf16x2(xin[8 * ks], xin[8 * ks + 1]); xb[ks][1] = pack_bf16x2(xin[8 * ks + 2], xin[8 * ks + 3]);
      xb[ks][2] = pack_bf16x2(xin[8 * ks + 4], xin[8 * ks + 5]); xb[ks][3] = pack_bf16x2(xin[8 * ks + 6], xin[8 * ks + 7]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ks = 0; ks < KB; ++ks)
        d1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1.f[mt][ks]), __builtin_bit_cast(bf16x8, xb[ks]), d1[mt], 0, 0, 0);
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          d1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w1.f[mt][kb][e]), xin[4 * kb + e], d1[mt], 0, 0, 0);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if constexpr (BF16) d1[mt] = gelu_poly4<false>(d1[mt]);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) d1[mt][j] = gelu_erf_f(d1[mt][j]);
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) out[mt] = f32x4{b2[mt][0], b2[mt][1], b2[mt][2], b2[mt][3]};
  if constexpr (BF16) {  // the hidden accumulators ARE the B operand (k order = accumulator order, matched by w2)
    u32x4 hb[KB];
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) {
      const f32x4 lo = d1[2 * ks], hi = d1[2 * ks + 1];
      hb[ks][0] = pack_bf16x2(lo[0], lo[1]); hb[ks][1] = pack_bf16x2(lo[2], lo[3]);
      hb[ks][2] = pack_bf16x2(hi[0], hi[1]); hb[ks][3] = pack_bf16x2(hi[2], hi[3]);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int ks = 0; ks < KB; ++ks)
        out[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2.f[mt][ks]), __builtin_bit_cast(bf16x8, hb[ks]), out[mt], 0, 0, 0);
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          out[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w2.f[mt][kb][e]), d1[kb][e], out[mt], 0, 0, 0);
  }
}

// one phase: lines of length n along an axis with element stride `ls` floats; line group g starts at g * gs
constexpr int AXT = 512;      // threads per workgroup of the fused H+W kernel (8 waves share the line groups of a phase)
constexpr int AXWS = 18;      // LDS floats per token (16 channels + 2): the w-strided reads of phase W hit 64 distinct banks
__host__ __device__ inline int axis_row_stride(int nW) {   // LDS floats per h row: = 2 (mod 8) spreads the h-strided reads of phase H
  int rs = nW * AXWS;
  while (rs % 8 != 2) ++rs;
  return rs;
}

// gout != nullptr (the LAST phase): the updated lines go straight to global memory (position stride gls, group stride ggs, channel
// l15) instead of back into the LDS plane -- saves the write-back, the barrier and the LDS -> global pass that would follow
template <bool BF16, int MT>
__device__ __forceinline__ void axis_phase(float* plane, float* wst_raw, const float* __restrict__ gw1, const float* __restrict__ gb1,
                                           const float* __restrict__ gw2, const float* __restrict__ gb2, int n, int ngroups, int ls,
                                           int gs, int tid, float* __restrict__ gout = nullptr, long gls = 0, long ggs = 0) {
  using ST = typename AxisStage<BF16>::T;
  const int lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  // stage this axis' weights in LDS: [b1 | b2] fp32, then w1 | w2 in the staging type
  float* bst = wst_raw;
  ST* wst = (ST*)(wst_raw + 2 * n);
  const int NS = axis_wstride<BF16>(n);
  for (int i = tid; i < n * n; i += AXT) {
    const int r = i / n, c = i - r * n;
    axis_wst<BF16>(wst, r * NS + c, gw1[i]);
    axis_wst<BF16>(wst, n * NS + r * NS + c, gw2[i]);
  }
  if (tid < n) { bst[tid] = gb1[tid]; bst[n + tid] = gb2[tid]; }
  __syncthreads();
  AxisW<BF16, MT> w1, w2;
  load_axis_weight<BF16, MT, false>(wst, n, l15, kk, w1);
  load_axis_weight<BF16, MT, true>(wst + n * NS, n, l15, kk, w2);
  float b1[MT][4], b2[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = mt * 16 + kk * 4 + j;
      b1[mt][j] = r < n ? bst[r] : 0.0f;
      b2[mt][j] = r < n ? bst[n + r] : 0.0f;
    }
  constexpr int NX = 16 * ((MT + 1) / 2);
  // software pipeline over this wave's line groups: the LDS gather of group g + 1 is in flight under the MFMAs / GELU of group g
  auto gather = [&](int g, float (&xv)[NX]) {
    const float* base = plane + g * gs + l15;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int p = BF16 ? ((j >> 3) * 32 + kk * 8 + (j & 7)) : ((j >> 2) * 16 + (j & 3) * 4 + kk);
      xv[j] = p < n ? base[p * ls] : 0.0f;
    }
  };
  float xa[NX], xb[NX];
  if (wave < ngroups) gather(wave, xa);
  for (int g = wave; g < ngroups; g += AXT / 64) {
    const int gn = g + AXT / 64;
    if (gn < ngroups) gather(gn, xb);
    float* base = plane + g * gs + l15;
    f32x4 out[MT];
    axis_mlp_mfma<BF16, MT>(w1, w2, b1, b2, xa, out);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = mt * 16 + kk * 4 + j;
        if (p < n) {
          if (gout) gout[(long)g * ggs + (long)p * gls + l15] = base[p * ls] + out[mt][j];
          else base[p * ls] += out[mt][j];       // this line group is owned by this wave
        }
      }
#pragma unroll
    for (int j = 0; j < NX; ++j) xa[j] = xb[j];
  }
  if (!gout) __syncthreads();
}

// optional source of the plane: the encoder's per-frame output before FiLM (a frame-major cache kept by the rollout loop), with
// FiLM(t) + positional embeddings applied while the plane is loaded -- x[b, t] = src[t][b] * a[t] + b[t] + s_emb  (tante.py:136-141)
struct AxisSrc {
  const float* src;          // nullptr: read x in place
  long t_stride, b_stride;   // elements between frames / batch items of src
  const float *fa, *fb, *se; // (T, C), (T, C), (nH * nW, C)
  int T;
};

template <bool BF16, int MT>
__global__ __launch_bounds__(AXT) void axis_hw_kernel(float* __restrict__ x, AxisSrc S, int nH, int nW, int C, const float* __restrict__ wh1,
                                                      const float* __restrict__ bh1, const float* __restrict__ wh2,
                                                      const float* __restrict__ bh2, const float* __restrict__ ww1,
                                                      const float* __restrict__ bw1, const float* __restrict__ ww2,
                                                      const float* __restrict__ bw2, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float plane[];  // [nH][rs], token (h, w) at h * rs + w * AXWS; then the weight staging
  const int tid = threadIdx.x;
  const int ctiles = C / 16;
  const long bt = blockIdx.x / ctiles;
  const int c0 = (blockIdx.x % ctiles) * 16;
  float* gx = x + bt * (long)nH * nW * C + c0;
  const int rs = axis_row_stride(nW);
  float* wst = plane + ((nH * rs + 3) & ~3);     // 16-byte aligned: the weight fragments are read as 16-byte pieces
  // ---- load the plane: 4 threads x 16 B per token ---------------------------------------------------------
  const float* lsrc = gx;
  const float *fa = nullptr, *fb = nullptr;
  if (S.src) {
    const long b = bt / S.T, t = bt - b * S.T;
    lsrc = S.src + t * S.t_stride + b * S.b_stride + c0;
    fa = S.fa + t * C + c0;
    fb = S.fb + t * C + c0;
  }
  if (!(dbg & 1))
  for (int i = tid; i < nH * nW * 4; i += AXT) {
    const int tokn = i >> 2, q = i & 3, h = tokn / nW, w = tokn - h * nW;
    f32x4 v = *(const f32x4*)(lsrc + (long)tokn * C + q * 4);
    if (S.src) {
      const f32x4 a = *(const f32x4*)(fa + q * 4), b = *(const f32x4*)(fb + q * 4), sv = *(const f32x4*)(S.se + (long)tokn * C + c0 + q * 4);
      v = v * a + b + sv;      // the encoder epilogue's expression, term for term
    }
    float2* d = (float2*)(plane + h * rs + w * AXWS + q * 4);   // 8-byte aligned: rs, AXWS even
    d[0] = make_float2(v[0], v[1]);
    d[1] = make_float2(v[2], v[3]);
  }
  __syncthreads();
  // phase H: lines along h (stride rs), one group per w;  phase W: lines along w (stride AXWS), one group per h
  if (!(dbg & 2)) axis_phase<BF16, MT>(plane, wst, wh1, bh1, wh2, bh2, nH, nW, rs, AXWS, tid);
  if (!(dbg & 4) && !(dbg & 1)) {   // phase W writes its lines (w positions of row h, 16 channels) to global memory itself
    axis_phase<BF16, MT>(plane, wst, ww1, bw1, ww2, bw2, nW, nH, AXWS, rs, tid, gx, (long)C, (long)nW * C);
    return;
  }
  if (!(dbg & 4)) axis_phase<BF16, MT>(plane, wst, ww1, bw1, ww2, bw2, nW, nH, AXWS, rs, tid);
  // ---- store the plane (ablation builds only: the W phase above normally stores) -------------------------------
  if (!(dbg & 1))
  for (int i = tid; i < nH * nW * 4; i += AXT) {
    const int tokn = i >> 2, q = i & 3, h = tokn / nW, w = tokn - h * nW;
    const float2* sp = (const float2*)(plane + h * rs + w * AXWS + q * 4);
    const float2 a = sp[0], b = sp[1];
    *(f32x4*)(gx + (long)tokn * C + q * 4) = f32x4{a.x, a.y, b.x, b.y};
  }
}

// ---- the same stage for whole 16-row tiles (bf16 compute, nH = 16 MTH, nW = 16 MTW): the shapes the rollouts run ----------
// Every size is a compile-time constant, so the generic kernel's `p < n` predication (a third of its instructions were EXEC
// bookkeeping and address arithmetic: 2 000 VALU per wave for 24 MFMAs) disappears, and the plane layout is chosen for WIDE accesses
// on both sides.  Global side: a workgroup owns CT = 32 channels of a plane when that fits the LDS (whole 128-byte lines per token:
// in-kernel stamps showed the 64-byte pieces of 16-channel tiles loading at 2.2 - 2.7 TB/s), else 16; the plane arrives by LDS-DMA
// (`global_load_lds`, lane-linear: one instruction = 1 KiB = 8 / 16 consecutive tokens of a row, no register round trip) and the
// workgroup id is turned into XCD-major order so that the channel tiles of one plane share an L2.  LDS side: a token's CT channels are
// contiguous with no padding, rows are padded to RS = 32 (mod 64) floats, and one wave iteration handles 32 lines as two MFMA column
// sets: lane (l15, kk) owns a channel PAIR (CT = 32: channels 2 l15, + 1 of one line group; CT = 16: channels 2 (l15 & 7), + 1 of
// group l15 >> 3 of a group pair), so every gather, residual read and write-back is a ds_read_b64 / ds_write_b64 (set 0 takes .x,
// set 1 takes .y).  The k slot (kk, e) of the 32-wide MFMA chunk ks holds line position 32 ks + 4 e + kk (W1's columns are loaded in
// that order): with it the 32 lanes of a ds_read_b64 service group cover all 64 banks in both phases (kk moves by one token along
// w, or by RS = 32 (mod 64) floats along h; the other lane bits fill the 32 floats in between).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__host__ __device__ constexpr int axe_rs(int nW, int ct) { return nW * ct + 32; }

template <int MT>
struct AxeW {   // A-operand fragments of W1 (k = line position) and W2 (k = hidden unit, accumulator order) + this lane's bias rows
  static constexpr int KB = (MT + 1) / 2;
  u32x4 a1[MT][KB], a2[MT][KB];
  f32x4 b1[MT], b2[MT];
};

// The two n x n weights of a phase reach the lanes through LDS: the workgroup reads them coalesced (every lane gathering its own
// fragment elements from global memory cost 100 scattered load instructions per wave -- stamps showed them, not the plane, holding
// the memory pipe for the first third of the kernel) and scatters them as bf16 into FRAGMENT order, so that afterwards a lane's
// A operand is one conflict-free ds_read_b128 per (row tile, k chunk).  Layout (u16 units): w1 at 0, w2 at MT KB 512, fragment
// (mt, ks) at (mt KB + ks) 512, lane l at + 8 l, k slot e at + e; then (byte offset axe_wbytes - 8 N) b1 | b2 as floats.
template <int MT>
__host__ __device__ constexpr int axe_wbytes() { return 2 * MT * ((MT + 1) / 2) * 1024 + 2 * 16 * MT * 4; }

template <int MT, int NT>
__device__ __forceinline__ void axe_stage_weights(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                  const float* __restrict__ b2, char* wst, int tid) {
  constexpr int N = 16 * MT, KB = (MT + 1) / 2;
  unsigned short* f = (unsigned short*)wst;
  for (int i = tid; i < N * N; i += NT) {
    const int row = i / N, col = i - row * N;          // N is a compile-time constant
    const int mt = row >> 4, l15 = row & 15;
    {  // w1[row][col]: col is the line position p = 32 ks + 4 e + kk
      const int ks = col >> 5, e = (col & 31) >> 2, kk = col & 3;
      __bf16 v = (__bf16)w1[i];
      f[(mt * KB + ks) * 512 + (kk * 16 + l15) * 8 + e] = __builtin_bit_cast(unsigned short, v);
    }
    {  // w2[row][col]: col is the hidden unit j = (2 ks + (e >> 2)) 16 + 4 kk + (e & 3)
      const int tile = col >> 4, ks = tile >> 1, kk = (col & 15) >> 2, e = (tile & 1) * 4 + (col & 3);
      __bf16 v = (__bf16)w2[i];
      f[(MT * KB + mt * KB + ks) * 512 + (kk * 16 + l15) * 8 + e] = __builtin_bit_cast(unsigned short, v);
    }
  }
  float* bs = (float*)(wst + 2 * MT * KB * 1024);
  if (tid < N) { bs[tid] = b1[tid]; bs[N + tid] = b2[tid]; }
}

template <int MT>
__device__ __forceinline__ void axe_load_weights(const char* wst, int lane, int kk, AxeW<MT>& W) {
  constexpr int N = 16 * MT, KB = AxeW<MT>::KB;
  const u32x4* f = (const u32x4*)wst;
  const float* bs = (const float*)(wst + 2 * MT * KB * 1024);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) {
      W.a1[mt][ks] = f[(mt * KB + ks) * 64 + lane];
      W.a2[mt][ks] = f[(MT * KB + mt * KB + ks) * 64 + lane];
      if (2 * ks + 1 >= MT) {    // odd MT: the upper half of the last k chunk does not exist (and was never written)
        W.a1[mt][ks][2] = W.a1[mt][ks][3] = 0u;
        W.a2[mt][ks][2] = W.a2[mt][ks][3] = 0u;
      }
    }
    W.b1[mt] = *(const f32x4*)(bs + mt * 16 + 4 * kk);
    W.b2[mt] = *(const f32x4*)(bs + N + mt * 16 + 4 * kk);
  }
}

// one phase over the LDS plane: lines of N = 16 MT positions with element stride LS floats; NG line groups GS floats apart.
// GOUT: the updated lines go to global memory (position stride gls, group stride ggs floats) instead of back into the plane.
// IL line groups are worked on at once, stage by stage (gather | pack | first contraction | GELU | pack | second contraction | update):
// one group is a chain of dependent LDS reads, packs and 4 - 16 MFMAs that a wave sits out latency by latency (stamps: 3 k cycles per
// group for ~100 instructions); two independent chains interleave.  IL = 2 is used by the 8-wave form (256 registers per lane).
template <int MT, int CT, int NWV, int NG, int LS, int GS, bool GOUT, int IL = 1>
__device__ __forceinline__ void axe_phase(float* plane, const AxeW<MT>& W, int wave, int l15, int kk, float* __restrict__ gout, long gls,
                                          long ggs) {
  constexpr int KB = AxeW<MT>::KB, NIT = CT == 32 ? NG : NG / 2, NE = 8 * KB;
  static_assert(CT == 32 || NG % 2 == 0, "16-channel tiles process line groups in pairs");
  // this lane's (line group within the iteration, channel pair)
  const int gsel = CT == 32 ? 0 : (l15 >> 3), cp = CT == 32 ? 2 * l15 : 2 * (l15 & 7);
  constexpr int GPI = CT == 32 ? 1 : 2;     // line groups per iteration
  auto gather = [&](int it, f32x2 (&xv)[NE]) {
    const float* base = plane + (GPI * it + gsel) * GS + cp + kk * LS;
#pragma unroll
    for (int ks = 0; ks < KB; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (ks * 32 + 4 * e < 16 * MT) xv[ks * 8 + e] = *(const f32x2*)(base + (ks * 32 + 4 * e) * LS);
        else xv[ks * 8 + e] = f32x2{0.f, 0.f};
  };
  f32x2 xa[IL][NE], xn[IL][NE];
#pragma unroll
  for (int g = 0; g < IL; ++g)
    if (wave + g * NWV < NIT) gather(wave + g * NWV, xa[g]);
  for (int it0 = wave; it0 < NIT; it0 += IL * NWV) {
#pragma unroll
    for (int g = 0; g < IL; ++g)
      if (it0 + (IL + g) * NWV < NIT) gather(it0 + (IL + g) * NWV, xn[g]);
    f32x4 out[IL][2][MT];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      u32x4 xb[IL][KB];
#pragma unroll
      for (int g = 0; g < IL; ++g)
#pragma unroll
        for (int ks = 0; ks < KB; ++ks)
#pragma unroll
          for (int q = 0; q < 4; ++q) xb[g][ks][q] = pack_bf16x2(xa[g][ks * 8 + 2 * q][a], xa[g][ks * 8 + 2 * q + 1][a]);
      f32x4 d1[IL][2 * KB];
#pragma unroll
      for (int g = 0; g < IL; ++g)
#pragma unroll
        for (int mt = 0; mt < 2 * KB; ++mt) d1[g][mt] = mt < MT ? W.b1[mt < MT ? mt : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KB; ++ks)
#pragma unroll
          for (int g = 0; g < IL; ++g)
            d1[g][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, W.a1[mt][ks]), __builtin_bit_cast(bf16x8, xb[g][ks]), d1[g][mt], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < IL; ++g)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) d1[g][mt] = gelu_poly4<false>(d1[g][mt]);
      u32x4 hb[IL][KB];
#pragma unroll
      for (int g = 0; g < IL; ++g)
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) {
          const f32x4 lo = d1[g][2 * ks], hi = d1[g][2 * ks + 1];
          hb[g][ks] = u32x4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
        }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int g = 0; g < IL; ++g) out[g][a][mt] = W.b2[mt];
#pragma unroll
        for (int ks = 0; ks < KB; ++ks)
#pragma unroll
          for (int g = 0; g < IL; ++g)
            out[g][a][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, W.a2[mt][ks]), __builtin_bit_cast(bf16x8, hb[g][ks]), out[g][a][mt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < IL; ++g) {
      const int it = it0 + g * NWV;
      if (it < NIT) {
        float* rbase = plane + (GPI * it + gsel) * GS + cp + 4 * kk * LS;   // this lane's output rows: positions 16 mt + 4 kk + r
        float* gbase = GOUT ? gout + (long)(GPI * it + gsel) * ggs + cp + 4 * kk * gls : nullptr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            f32x2 v = *(const f32x2*)(rbase + (16 * mt + r) * LS);
            v[0] += out[g][0][mt][r];
            v[1] += out[g][1][mt][r];
#ifdef AXE_PLAIN_STORE
            if constexpr (GOUT) *(f32x2*)(gbase + (long)(16 * mt + r) * gls) = v;
#else       // write-through (common.hip.h): the finished plane streams out, nothing dirty is left for the kernel-end release
            if constexpr (GOUT) st_wt8(gbase + (long)(16 * mt + r) * gls, __builtin_bit_cast(u32x2, v));
#endif
            else *(f32x2*)(rbase + (16 * mt + r) * LS) = v;      // these lines belong to this wave alone
          }
      }
    }
#pragma unroll
    for (int g = 0; g < IL; ++g)
#pragma unroll
      for (int j = 0; j < NE; ++j) xa[g][j] = xn[g][j];
  }
}

// diagnostic builds (-DTANTE_ABLATE) only: per-wave shader-clock stamps at the stage boundaries (tools/axe_stamps.py); empty otherwise
#ifdef TANTE_ABLATE
unsigned long long* g_axe_stamps = nullptr;
#define AXE_STAMP(k)                                                                  \
  do {                                                                                \
    if (stamps) {                                                                     \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                     \
      if (lane == 0) stamps[((long)blockIdx.x * 16 + wave) * 12 + (k)] = t_;          \
    }                                                                                 \
  } while (0)
#else
#define AXE_STAMP(k)
#endif

template <int MTH, int MTW, int CT, int NT>
__global__ __launch_bounds__(NT) void axis_hw_exact_kernel(float* __restrict__ x, AxisSrc S, int C, const float* __restrict__ wh1,
                                                           const float* __restrict__ bh1, const float* __restrict__ wh2,
                                                           const float* __restrict__ bh2, const float* __restrict__ ww1,
                                                           const float* __restrict__ bw1, const float* __restrict__ ww2,
                                                           const float* __restrict__ bw2, unsigned long long* stamps,
                                                           const float* __restrict__ xin, float* __restrict__ xmid) {
  // xin (training forward): read the planes from here instead of x (out of place: the input stays intact for the backward pass);
  // xmid: the planes after the H propagator, i.e. the W propagator's input, which its backward needs
  constexpr int NH = 16 * MTH, NW = 16 * MTW, RS = axe_rs(NW, CT), NWV = NT / 64;
  constexpr int AXE_IL = (NT == 512 && CT == 32 && MTH <= 2 && MTW <= 2) ? 2 : 1;      // two line groups at once where 256 registers allow
  constexpr int TPI = 256 / CT, LPT = CT / 4;    // tokens per DMA instruction (1 KiB), lanes per token
  extern __shared__ __attribute__((aligned(16))) float plane[];  // [NH][RS], token (h, w) at h * RS + w * CT
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  const int ctiles = C / CT;
  // workgroup id -> (plane, channel tile): consecutive ids go to consecutive XCDs, so id = 8 slot + xcd is turned into xcd-major order:
  // the channel tiles of one plane then run on ONE XCD at about the same time and its L2 sees whole token rows
  unsigned wg = blockIdx.x;
  if (gridDim.x % 8 == 0) wg = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const long bt = wg / ctiles;
  const int c0 = (wg % ctiles) * CT;
  float* gx = x + bt * (long)NH * NW * C + c0;
  AXE_STAMP(0);
#ifdef TANTE_ABLATE
  if (stamps && lane == 0) stamps[((long)blockIdx.x * 16 + wave) * 12 + 8] = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- the plane: one instruction = TPI tokens of a row (lanes LPT t .. LPT t + LPT - 1 = the CT channels of token t) ------------
  constexpr int IPR = NW / TPI;                  // instructions per row
  if (!S.src) {
    const float* gin = xin ? xin + (gx - x) : gx;
    for (int q = wave; q < NH * IPR; q += NWV) {
      const int h = q / IPR, j = q - h * IPR;
      const float* g = gin + ((long)h * NW + j * TPI + lane / LPT) * C + (lane % LPT) * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(plane + h * RS + j * 256), 16, 0, 0);
    }
  } else {     // the frame-major pre-FiLM encoder cache: FiLM(t) + positional embedding while loading (the encoder epilogue's expression)
    const long b = bt / S.T, t = bt - b * S.T;
    const float* lsrc = S.src + t * S.t_stride + b * S.b_stride + c0;
    const int q4 = (lane % LPT) * 4;
    const f32x4 fa = *(const f32x4*)(S.fa + t * C + c0 + q4), fb = *(const f32x4*)(S.fb + t * C + c0 + q4);
    for (int q = wave; q < NH * IPR; q += NWV) {
      const int h = q / IPR, j = q - h * IPR, tokn = h * NW + j * TPI + lane / LPT;
      const f32x4 v = *(const f32x4*)(lsrc + (long)tokn * C + q4), sv = *(const f32x4*)(S.se + (long)tokn * C + c0 + q4);
      *(f32x4*)(plane + h * RS + j * 256 + lane * 4) = v * fa + fb + sv;
    }
  }
  char* wstH = (char*)(plane + NH * RS);
  char* wstW = wstH + axe_wbytes<MTH>();
  axe_stage_weights<MTH, NT>(wh1, bh1, wh2, bh2, wstH, tid);
  axe_stage_weights<MTW, NT>(ww1, bw1, ww2, bw2, wstW, tid);
  AXE_STAMP(1);
  __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): this wave's part of the plane has landed
  AXE_STAMP(2);
  __syncthreads();
  AXE_STAMP(3);
  // phase H: lines along h (stride RS), one group per w (stride CT);  phase W: lines along w (stride CT), one group per h
  {
    AxeW<MTH> WH;
    axe_load_weights<MTH>(wstH, lane, kk, WH);
    axe_phase<MTH, CT, NWV, NW, RS, CT, false, AXE_IL>(plane, WH, wave, l15, kk, nullptr, 0, 0);
  }
  AXE_STAMP(4);
  AxeW<MTW> WW;
  axe_load_weights<MTW>(wstW, lane, kk, WW);
  __syncthreads();
  AXE_STAMP(5);
  if (xmid) {      // the plane as the H propagator left it, row form (the mirror image of the LDS-DMA pass above)
    float* gm = xmid + (gx - x);
    for (int q = wave; q < NH * IPR; q += NWV) {
      const int h = q / IPR, j = q - h * IPR;
      *(f32x4*)(gm + ((long)h * NW + j * TPI + lane / LPT) * C + (lane % LPT) * 4) = *(const f32x4*)(plane + h * RS + j * 256 + lane * 4);
    }
  }
  axe_phase<MTW, CT, NWV, NH, CT, RS, true, AXE_IL>(plane, WW, wave, l15, kk, gx, (long)C, (long)NW * C);
  AXE_STAMP(6);
#ifdef TANTE_ABLATE
  if (stamps) {
    __builtin_amdgcn_s_waitcnt(0x0070);
    AXE_STAMP(7);
    if (lane == 0) stamps[((long)blockIdx.x * 16 + wave) * 12 + 9] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ---- FiLM tables ------------------------------------------------------------------------------------
// a[r][c] = 1 + W2s relu(w0s * t[r] + b0s) + b2s ; b[r][c] = W2h relu(w0h * t[r] + b0h) + b2h (+ add[r][c])
__global__ void film_table_kernel(const float* __restrict__ t, int rows, int C, const float* __restrict__ sc_w0,
                                  const float* __restrict__ sc_b0, const float* __restrict__ sc_w2, const float* __restrict__ sc_b2,
                                  const float* __restrict__ sh_w0, const float* __restrict__ sh_b0, const float* __restrict__ sh_w2,
                                  const float* __restrict__ sh_b2, const float* __restrict__ add, float* __restrict__ a_out,
                                  float* __restrict__ b_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * C) return;
  const int r = idx / C, c = idx % C, Hd = C / 2;
  const float tv = t[r];
  float sa = sc_b2[c], sb = sh_b2[c];
  for (int j = 0; j < Hd; ++j) {
    sa += sc_w2[c * Hd + j] * fmaxf(sc_w0[j] * tv + sc_b0[j], 0.0f);
    sb += sh_w2[c * Hd + j] * fmaxf(sh_w0[j] * tv + sh_b0[j], 0.0f);
  }
  a_out[idx] = 1.0f + sa;
  b_out[idx] = sb + (add ? add[idx] : 0.0f);
}

// Backward of the tables: per MLP  out[r][c] = b2[c] + sum_j w2[c][j] relu(w0[j] t[r] + b0[j]);  dOut = dA (scale) / dB (shift), (rows, C).
// grid = (Hd / 8, 2 MLPs): a workgroup owns 8 hidden units j of one MLP; thread c owns output channel c (C <= 256 per pass):
//   gw2[c][j] = sum_r dO[r][c] h[r][j]                      (8 consecutive floats per thread)
//   dh[r][j]  = sum_c dO[r][c] w2[c][j]  -> [h > 0]        (a sum over the workgroup's threads: wave DPP + LDS)
//   gw0[j] = sum_r dh[r][j] t[r],  gb0[j] = sum_r dh[r][j];  gb2[c] = sum_r dO[r][c]  (by the j = 0 workgroup)
// Results are ADDED onto the gradient buffers when `acc` (the parameters' .grad slots) and written otherwise.  rows <= 8.
__global__ __launch_bounds__(256) void film_table_bwd_kernel(const float* __restrict__ t, int rows, int C, const float* __restrict__ sc_w0,
                                                            const float* __restrict__ sc_b0, const float* __restrict__ sc_w2,
                                                            const float* __restrict__ sh_w0, const float* __restrict__ sh_b0,
                                                            const float* __restrict__ sh_w2, const float* __restrict__ dA,
                                                            const float* __restrict__ dB, float* g_sc_w0, float* g_sc_b0, float* g_sc_w2,
                                                            float* g_sc_b2, float* g_sh_w0, float* g_sh_b0, float* g_sh_w2, float* g_sh_b2,
                                                            int acc) {
  __shared__ float red[4][8][8];      // [wave][row][j]
  const int Hd = C / 2, tid = threadIdx.x, m = blockIdx.y, j0 = blockIdx.x * 8;
  const float* w0 = m ? sh_w0 : sc_w0;
  const float* b0 = m ? sh_b0 : sc_b0;
  const float* w2 = m ? sh_w2 : sc_w2;
  const float* dOut = m ? dB : dA;
  float* gw0 = m ? g_sh_w0 : g_sc_w0;
  float* gb0 = m ? g_sh_b0 : g_sc_b0;
  float* gw2 = m ? g_sh_w2 : g_sc_w2;
  float* gb2 = m ? g_sh_b2 : g_sc_b2;
  float tv[8], h[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    tv[r] = r < rows ? t[r] : 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[r][j] = r < rows ? fmaxf(w0[j0 + j] * tv[r] + b0[j0 + j], 0.0f) : 0.0f;
  }
  float part[8][8];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) part[r][j] = 0.0f;
  for (int c = tid; c < C; c += 256) {
    float d[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = r < rows ? dOut[r * C + c] : 0.0f;
    float w[8];
    const f32x4 wa = *(const f32x4*)(w2 + (long)c * Hd + j0), wb = *(const f32x4*)(w2 + (long)c * Hd + j0 + 4);
    w[0] = wa[0]; w[1] = wa[1]; w[2] = wa[2]; w[3] = wa[3]; w[4] = wb[0]; w[5] = wb[1]; w[6] = wb[2]; w[7] = wb[3];
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float sw = 0.0f;
#pragma unroll
      for (int r = 0; r < 8; ++r) { sw += d[r] * h[r][j]; part[r][j] += d[r] * w[j]; }
      g[j] = sw;
    }
    float* o = gw2 + (long)c * Hd + j0;
    f32x4 oa = f32x4{g[0], g[1], g[2], g[3]}, ob = f32x4{g[4], g[5], g[6], g[7]};
    if (acc) { oa += *(const f32x4*)o; ob += *(const f32x4*)(o + 4); }
    *(f32x4*)o = oa; *(f32x4*)(o + 4) = ob;
    if (blockIdx.x == 0) {
      float sb = 0.0f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sb += d[r];
      gb2[c] = (acc ? gb2[c] : 0.0f) + sb;
    }
  }
  // dh[r][j]: sum of part over the workgroup's threads
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = part[r][j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
      if ((tid & 63) == 0) red[tid >> 6][r][j] = v;
    }
  __syncthreads();
  if (tid < 8) {
    const int j = tid;
    float s0 = 0.0f, s1 = 0.0f;
    for (int r = 0; r < rows; ++r) {
      const float dh = (red[0][r][j] + red[1][r][j]) + (red[2][r][j] + red[3][r][j]);
      const float hv = fmaxf(w0[j0 + j] * t[r] + b0[j0 + j], 0.0f);
      const float dp = hv > 0.0f ? dh : 0.0f;      // relu'(pre) = [pre > 0] = [h > 0]
      s0 += dp * t[r];
      s1 += dp;
    }
    gw0[j0 + j] = (acc ? gw0[j0 + j] : 0.0f) + s0;
    gb0[j0 + j] = (acc ? gb0[j0 + j] : 0.0f) + s1;
  }
}

__global__ void film_apply_kernel(const float* __restrict__ x, long x_bstride, float* __restrict__ y, long rows, int C4,
                                  long rows_per, const float* __restrict__ a, const float* __restrict__ b) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 per thread
  if (idx >= rows * C4) return;
  const long r = idx / C4;
  const int c4 = (int)(idx - r * C4);
  const long g = r / rows_per, i = r - g * rows_per;
  const f32x4 xv = *(const f32x4*)(x + g * x_bstride + (i * C4 + c4) * 4);
  const f32x4 av = ((const f32x4*)a)[g * C4 + c4], bv = ((const f32x4*)b)[g * C4 + c4];
  ((f32x4*)y)[idx] = xv * av + bv;
}

// out[i] = z[i * E + E - 1]  (letter 'C': keep the last lifted channel, attn_backbone.py:188)
__global__ void gather_last_kernel(const float* __restrict__ z, long n, int E, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = z[i * E + (E - 1)];
}

// ---- formatter input: 'b t h w c -> b t c h w' + nan_to_num in one pass (data/datamodule.py:184-192) --------------------------------
// x: (n_img, HW, D) channels-last; out image i at out + (i / T) * out_bstride + (i % T) * D * HW, (D, HW) channels-first.
// A workgroup moves 256 pixels of one image: coalesced reads of 256 * D consecutive floats into LDS, coalesced writes of D rows of 256.
__device__ __forceinline__ float nan_to_num_f(float v) {
  if (v != v) return 0.0f;                                   // torch.nan_to_num defaults: nan -> 0, +-inf -> +-FLT_MAX
  return fminf(fmaxf(v, -3.40282346638528859812e+38f), 3.40282346638528859812e+38f);
}
__global__ __launch_bounds__(256) void format_input_kernel(const float* __restrict__ x, long HW, int D, int T, float* __restrict__ out,
                                                           long out_bstride) {
  extern __shared__ float fsm[];                             // [256][D + 1]
  const long img = blockIdx.y, p0 = (long)blockIdx.x * 256;
  const int np = (int)min(256L, HW - p0), tid = threadIdx.x;
  const float* src = x + (img * HW + p0) * D;
  for (int i = tid; i < np * D; i += 256) fsm[(i / D) * (D + 1) + (i % D)] = src[i];
  __syncthreads();
  float* dst = out + (img / T) * out_bstride + (img % T) * (long)D * HW + p0;
  for (int d = 0; d < D; ++d)
    if (tid < np) dst[(long)d * HW + tid] = nan_to_num_f(fsm[tid * (D + 1) + d]);
}

// D = 4 (cfg2 / cfg3: four fields): a lane takes FOUR consecutive pixels -- four 16-byte loads of 64 contiguous bytes, a 4 x 4 transpose in
// registers, one 16-byte store per channel row.  The generic kernel above moves 4 bytes per lane and instruction through LDS: 39.7 us per
// rollout at cfg2 for 12.6 MB (r06_rollout_kernel_stats.csv).  Needs HW % 4 == 0 and 16-byte aligned rows.
__global__ __launch_bounds__(256) void format_input_d4_kernel(const float* __restrict__ x, long HW, int T, float* __restrict__ out, long out_bstride) {
  const long img = blockIdx.y, q = (long)blockIdx.x * 256 + threadIdx.x;      // q: group of 4 pixels
  if (4 * q >= HW) return;
  const f32x4* src = (const f32x4*)(x + (img * HW + 4 * q) * 4);
  const f32x4 a = src[0], b = src[1], c = src[2], d = src[3];
  float* dst = out + (img / T) * out_bstride + (img % T) * 4 * HW + 4 * q;
#pragma unroll
  for (int ch = 0; ch < 4; ++ch)
    *(f32x4*)(dst + (long)ch * HW) = f32x4{nan_to_num_f(a[ch]), nan_to_num_f(b[ch]), nan_to_num_f(c[ch]), nan_to_num_f(d[ch])};
}

// torch.nan_to_num on a dense fp32 tensor (the formatter's pass over the reference frames, data/datamodule.py:187): 16 bytes per lane,
// four vectors in flight per lane (the torch elementwise kernel ran this pass at 55 us per rollout; a first version of this one with
// 8 MiB between a lane's vectors was SLOWER, 68 us)
__global__ __launch_bounds__(256) void nan_to_num_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, long n) {
  // a workgroup owns 4 consecutive KiB-rows of 256 float4 (16 KiB): four independent 16-byte loads per lane, all issued before the first use
  const long base = (long)blockIdx.x * 1024 + threadIdx.x;
  f32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + 256 * k < n4) v[k] = ((const f32x4*)x)[base + 256 * k];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (base + 256 * k < n4) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = nan_to_num_f(v[k][e]);
      ((f32x4*)y)[base + 256 * k] = o;
    }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) y[4 * n4 + threadIdx.x] = nan_to_num_f(x[4 * n4 + threadIdx.x]);
}

// ---- Taylor sum -------------------------------------------------------------------------------------
struct TaylorArgs {
  const float* d[8];
  float coef[8][8];  // [out frame i-1][order k-1] = (i*dt)^k / k!
};

template <int NO>
__global__ void taylor_kernel(const float* __restrict__ last, long last_bstride, const TaylorArgs ta, int n_out,
                              float* __restrict__ out, long out_bstride, long B, long frame4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index within (B, frame)
  if (idx >= B * frame4) return;
  const long b = idx / frame4, f = idx - b * frame4;
  const f32x4 base = *(const f32x4*)(last + b * last_bstride + 4 * f);
  f32x4 dv[NO];
#pragma unroll
  for (int k = 0; k < NO; ++k) dv[k] = ((const f32x4*)ta.d[k])[idx];
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // static indices only: a runtime index into the kernarg struct goes to scratch
    if (i < n_out) {
      f32x4 o = base;
#pragma unroll
      for (int k = 0; k < NO; ++k) o += dv[k] * ta.coef[i][k];
      *(f32x4*)(out + b * out_bstride + ((long)i * frame4 + f) * 4) = o;
    }
  }
}

// rt[b] = mean_l clamp(t[b][l], 0, out_T - 1) + ep ; one wave per sample
__global__ __launch_bounds__(64) void rt_reduce_kernel(const float* __restrict__ t, int L, float hi, float ep, float* __restrict__ rt) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float s = 0.0f;
  for (int l = lane; l < L; l += 64) {
    const float v = t[(long)b * L + l];
    // t + relu(-t) - relu(t - hi): forward value of the straight-through clamp (tante.py:196-198)
    s += v + fmaxf(-v, 0.0f) - fmaxf(v - hi, 0.0f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) rt[b] = s / (float)L + ep;
}

template <int N>
void launch_axis(float* x, long outer, int n, long inner, const float* w1, const float* b1, const float* w2, const float* b2,
                 hipStream_t s) {
  const long cols = outer * inner;
  hipLaunchKernelGGL(axis_mlp_kernel<N>, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, x, outer, n, inner, w1, b1, w2, b2);
}

}  // namespace

extern "C" int tante_axis_mlp_c(float* x, int64_t outer, int n, int64_t inner, const float* w1, const float* b1, const float* w2,
                                const float* b2, int compute, void* stream) {
  if (!x || !w1 || !b1 || !w2 || !b2) TANTE_FAIL(-1, "tante_axis_mlp_c: null pointer");
  if (outer <= 0 || n <= 0 || inner <= 0) TANTE_FAIL(-1, "tante_axis_mlp_c: bad shape");
  // long axes in the bf16 mode: the two n x n products on the matrix pipe (axis_mfma.hip); TANTE_AXIS_MFMA = 0: the fp32 vector kernel
  if (compute == TANTE_BF16 && tante_axis_mlp_mfma_supported(outer, n, inner, x, w1, w2) && ((uintptr_t)b1 % 16) == 0 && ((uintptr_t)b2 % 16) == 0 &&
      tante_opt("TANTE_AXIS_MFMA", 1)) {
    const int rc = tante_axis_mlp_mfma(x, outer, n, inner, w1, b1, w2, b2, (hipStream_t)stream);
    if (rc) TANTE_FAIL(rc, "tante_axis_mlp_c: matrix-pipe propagator launch failed");
    return 0;
  }
  if (n <= 8 && inner % 4 == 0 && ((uintptr_t)x % 16) == 0) {
    hipStream_t s = (hipStream_t)stream;
    const long cols = outer * (inner / 4);
    const dim3 grid((unsigned)((cols + 255) / 256));
#define TANTE_AV(NN, FF) hipLaunchKernelGGL((axis_mlp_vec_kernel<NN, FF>), grid, dim3(256), 0, s, x, (long)outer, n, (long)(inner / 4), w1, b1, w2, b2)
    if (n <= 4) { if (compute == TANTE_BF16) TANTE_AV(4, true); else TANTE_AV(4, false); }
    else { if (compute == TANTE_BF16) TANTE_AV(8, true); else TANTE_AV(8, false); }
#undef TANTE_AV
    TANTE_CHECK_LAUNCH();
    return 0;
  }
  return tante_axis_mlp(x, outer, n, inner, w1, b1, w2, b2, stream);
}

extern "C" int tante_axis_mlp_oop(const float* src, float* dst, int64_t outer, int n, int64_t inner, const float* w1, const float* b1,
                                  const float* w2, const float* b2, int compute, void* stream) {
  if (!src || !dst || !w1 || !b1 || !w2 || !b2) TANTE_FAIL(-1, "tante_axis_mlp_oop: null pointer");
  if (outer <= 0 || n <= 0 || inner <= 0) TANTE_FAIL(-1, "tante_axis_mlp_oop: bad shape");
  if (n > 8 || inner % 4 || ((uintptr_t)src % 16) || ((uintptr_t)dst % 16))
    TANTE_FAIL(-2, "tante_axis_mlp_oop: short axes (n <= 8) with 16-byte columns only (copy, then tante_axis_mlp_c)");
  hipStream_t s = (hipStream_t)stream;
  const long cols = outer * (inner / 4);
  const dim3 grid((unsigned)((cols + 255) / 256));
#define TANTE_AVO(NN, FF) hipLaunchKernelGGL((axis_mlp_vec_kernel<NN, FF>), grid, dim3(256), 0, s, dst, (long)outer, n, (long)(inner / 4), w1, b1, w2, b2, src)
  if (n <= 4) { if (compute == TANTE_BF16) TANTE_AVO(4, true); else TANTE_AVO(4, false); }
  else { if (compute == TANTE_BF16) TANTE_AVO(8, true); else TANTE_AVO(8, false); }
#undef TANTE_AVO
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_axis_mlp(float* x, int64_t outer, int n, int64_t inner, const float* w1, const float* b1,
                              const float* w2, const float* b2, void* stream) {
  if (!x || !w1 || !b1 || !w2 || !b2) TANTE_FAIL(-1, "tante_axis_mlp: null pointer");
  if (outer <= 0 || n <= 0 || inner <= 0) TANTE_FAIL(-1, "tante_axis_mlp: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (n <= 4) launch_axis<4>(x, outer, n, inner, w1, b1, w2, b2, s);
  else if (n <= 8) launch_axis<8>(x, outer, n, inner, w1, b1, w2, b2, s);
  else if (n <= 16) launch_axis<16>(x, outer, n, inner, w1, b1, w2, b2, s);
  else if (n <= 32) launch_axis<32>(x, outer, n, inner, w1, b1, w2, b2, s);
  else if (n <= 48) launch_axis<48>(x, outer, n, inner, w1, b1, w2, b2, s);
  else if (n <= 64) launch_axis<64>(x, outer, n, inner, w1, b1, w2, b2, s);
  else {
    const size_t lds = 2 * (size_t)n * 64 * sizeof(float);
    if (lds > 160 * 1024) TANTE_FAIL(-2, "tante_axis_mlp: axis length %d too long", n);
    if (lds > 64 * 1024)
      hipFuncSetAttribute((const void*)axis_mlp_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long cols = outer * inner;
    hipLaunchKernelGGL(axis_mlp_lds_kernel, dim3((unsigned)((cols + 63) / 64)), dim3(64), lds, s, x, outer, n, inner, w1, b1, w2, b2);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}

template <bool BF16, int MT>
static void launch_axis_hw(float* x, const AxisSrc& S, long BT, int nH, int nW, int C, const float* wh1, const float* bh1, const float* wh2,
                           const float* bh2, const float* ww1, const float* bw1, const float* ww2, const float* bw2, size_t lds,
                           hipStream_t s) {
  static size_t attr = 0;
  if (lds > attr) {
    hipFuncSetAttribute((const void*)axis_hw_kernel<BF16, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = lds;
  }
  hipLaunchKernelGGL((axis_hw_kernel<BF16, MT>), dim3((unsigned)(BT * (C / 16))), dim3(AXT), lds, s, x, S, nH, nW, C, wh1, bh1, wh2, bh2,
                     ww1, bw1, ww2, bw2, tante_ablate_env("TANTE_AXIS_DEBUG"));
}

static int axis_hw_impl(float* x, const AxisSrc& S, int64_t BT, int nH, int nW, int C, const float* wh1, const float* bh1, const float* wh2,
                        const float* bh2, const float* ww1, const float* bw1, const float* ww2, const float* bw2, int compute, void* stream,
                        const float* xin = nullptr, float* xmid = nullptr);

#ifdef TANTE_ABLATE
extern "C" void tante_axe_set_stamps(unsigned long long* p) { g_axe_stamps = p; }
#endif

extern "C" int tante_axis_hw_train(const float* xin, float* xout, float* xmid, int64_t BT, int nH, int nW, int C, const float* wh1, const float* bh1,
                                   const float* wh2, const float* bh2, const float* ww1, const float* bw1, const float* ww2, const float* bw2,
                                   int compute, void* stream) {
  if (!xin || !xmid || ((uintptr_t)xin % 16) || ((uintptr_t)xmid % 16)) TANTE_FAIL(-1, "tante_axis_hw_train: null or misaligned pointer");
  const AxisSrc none = {nullptr, 0, 0, nullptr, nullptr, nullptr, 1};
  return axis_hw_impl(xout, none, BT, nH, nW, C, wh1, bh1, wh2, bh2, ww1, bw1, ww2, bw2, compute, stream, xin, xmid);
}

extern "C" int tante_axis_hw_oop(const float* xin, float* xout, int64_t BT, int nH, int nW, int C, const float* wh1, const float* bh1,
                                 const float* wh2, const float* bh2, const float* ww1, const float* bw1, const float* ww2, const float* bw2,
                                 int compute, void* stream) {
  if (!xin || ((uintptr_t)xin % 16) || xin == xout) TANTE_FAIL(-1, "tante_axis_hw_oop: null or misaligned source (or source == destination: use tante_axis_hw)");
  const AxisSrc none = {nullptr, 0, 0, nullptr, nullptr, nullptr, 1};
  return axis_hw_impl(xout, none, BT, nH, nW, C, wh1, bh1, wh2, bh2, ww1, bw1, ww2, bw2, compute, stream, xin, nullptr);
}

extern "C" int tante_axis_hw(float* x, int64_t BT, int nH, int nW, int C, const float* wh1, const float* bh1, const float* wh2,
                             const float* bh2, const float* ww1, const float* bw1, const float* ww2, const float* bw2,
                             int compute, void* stream) {
  const AxisSrc none = {nullptr, 0, 0, nullptr, nullptr, nullptr, 1};
  return axis_hw_impl(x, none, BT, nH, nW, C, wh1, bh1, wh2, bh2, ww1, bw1, ww2, bw2, compute, stream);
}

extern "C" int tante_axis_hw_film(float* x, const float* src, int64_t src_t_stride, int64_t src_b_stride, const float* film_a,
                                  const float* film_b, const float* s_emb, int T, int64_t BT, int nH, int nW, int C, const float* wh1,
                                  const float* bh1, const float* wh2, const float* bh2, const float* ww1, const float* bw1, const float* ww2,
                                  const float* bw2, int compute, void* stream) {
  if (!src || !film_a || !film_b || !s_emb || T <= 0 || BT % T) TANTE_FAIL(-1, "tante_axis_hw_film: bad argument");
  if (((uintptr_t)src | (uintptr_t)film_a | (uintptr_t)film_b | (uintptr_t)s_emb) % 16 || src_t_stride % 4 || src_b_stride % 4)
    TANTE_FAIL(-1, "tante_axis_hw_film: 16-byte alignment");
  const AxisSrc S = {src, (long)src_t_stride, (long)src_b_stride, film_a, film_b, s_emb, T};
  return axis_hw_impl(x, S, BT, nH, nW, C, wh1, bh1, wh2, bh2, ww1, bw1, ww2, bw2, compute, stream);
}

static int axis_hw_impl(float* x, const AxisSrc& S, int64_t BT, int nH, int nW, int C, const float* wh1, const float* bh1, const float* wh2,
                        const float* bh2, const float* ww1, const float* bw1, const float* ww2, const float* bw2, int compute, void* stream,
                        const float* xin, float* xmid) {
  if (!x || !wh1 || !bh1 || !wh2 || !bh2 || !ww1 || !bw1 || !ww2 || !bw2) TANTE_FAIL(-1, "tante_axis_hw: null pointer");
  if (BT <= 0 || nH <= 0 || nW <= 0 || C <= 0) TANTE_FAIL(-1, "tante_axis_hw: bad shape");
  hipStream_t s = (hipStream_t)stream;
  auto wbytes = [](int m) { return (size_t)(2 * m * ((m + 1) / 2) * 1024 + 2 * 16 * m * 4); };   // = axe_wbytes<m>()
  const size_t wst_bytes = wbytes(nH / 16) + wbytes(nW / 16);
  if (compute == TANTE_BF16 && nH % 16 == 0 && nW % 16 == 0 && nH <= 64 && nW <= 64 && C % 16 == 0 && ((uintptr_t)x % 16) == 0 &&
      (size_t)nH * axe_rs(nW, 16) * sizeof(float) + wst_bytes <= 160 * 1024 && !tante_opt("TANTE_AXIS_GENERIC", 0)) {
    // 32-channel tiles (whole 128-byte lines per token) when the plane fits the LDS that way and still gives every CU a workgroup
    const int force_ct = tante_opt("TANTE_AXIS_CT", 0);
    const bool fits32 = C % 32 == 0 && (size_t)nH * axe_rs(nW, 32) * sizeof(float) + wst_bytes <= 160 * 1024;
    const int ct = force_ct == 16 ? 16 : ((fits32 && (force_ct == 32 || BT * (C / 32) >= 192)) ? 32 : 16);
    const size_t elds = (size_t)nH * axe_rs(nW, ct) * sizeof(float) + wst_bytes;
    const unsigned grid = (unsigned)(BT * (C / ct));
    // 16 waves for the 32-channel tiles when the fragments fit 128 registers (measured 16.2 against 17.2 us at 32 x 32)
    const int force_nt = tante_opt("TANTE_AXIS_NT", 0);
    const int nt32 = force_nt ? force_nt : ((nH <= 32 && nW <= 32) ? 1024 : 512);
#ifdef TANTE_ABLATE
    unsigned long long* axe_st = g_axe_stamps;
#else
    unsigned long long* axe_st = nullptr;
#endif
#define TANTE_AXE2(MH, MW, CTV, NTV)                                                                                                    \
  {                                                                                                                                     \
    static TantePerDevice attr;                                                                                                         \
    attr.once([&] { (void)hipFuncSetAttribute((const void*)axis_hw_exact_kernel<MH, MW, CTV, NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
    hipLaunchKernelGGL((axis_hw_exact_kernel<MH, MW, CTV, NTV>), dim3(grid), dim3(NTV), elds, s, x, S, C, wh1, bh1, wh2, bh2, ww1, bw1, ww2, bw2, axe_st, xin, xmid); \
  }
#define TANTE_AXE(MH, MW)                  \
  {                                        \
    if (ct == 32 && nt32 == 1024) TANTE_AXE2(MH, MW, 32, 1024) \
    else if (ct == 32) TANTE_AXE2(MH, MW, 32, 512) \
    else TANTE_AXE2(MH, MW, 16, 512)       \
  }
    switch ((nH / 16) * 8 + nW / 16) {
      case 1 * 8 + 1: TANTE_AXE(1, 1); break; case 1 * 8 + 2: TANTE_AXE(1, 2); break; case 1 * 8 + 3: TANTE_AXE(1, 3); break; case 1 * 8 + 4: TANTE_AXE(1, 4); break;
      case 2 * 8 + 1: TANTE_AXE(2, 1); break; case 2 * 8 + 2: TANTE_AXE(2, 2); break; case 2 * 8 + 3: TANTE_AXE(2, 3); break; case 2 * 8 + 4: TANTE_AXE(2, 4); break;
      case 3 * 8 + 1: TANTE_AXE(3, 1); break; case 3 * 8 + 2: TANTE_AXE(3, 2); break; case 3 * 8 + 3: TANTE_AXE(3, 3); break;
      case 4 * 8 + 1: TANTE_AXE(4, 1); break; case 4 * 8 + 2: TANTE_AXE(4, 2); break;
      default: TANTE_FAIL(-2, "tante_axis_hw: no whole-tile variant for %d x %d", nH, nW);
    }
#undef TANTE_AXE
#undef TANTE_AXE2
    TANTE_CHECK_LAUNCH();
    return 0;
  }
  if (xin || xmid) TANTE_FAIL(-2, "tante_axis_hw_train: needs the whole-tile bf16 form (nH, nW multiples of 16, the plane within the LDS)");
  const int nmax = nH > nW ? nH : nW;
  const size_t lds = (((size_t)nH * axis_row_stride(nW) + 3) & ~(size_t)3) * sizeof(float) + 2 * (size_t)nmax * sizeof(float) +
                     2 * (size_t)nmax * (compute == TANTE_BF16 ? (nmax + 8) * 2 : nmax * 4);
  if (nmax > 64 || C % 16 || lds > 160 * 1024 || ((uintptr_t)x % 16))
    TANTE_FAIL(-2, "tante_axis_hw: needs nH, nW <= 64, C %% 16 == 0 and the plane to fit LDS (use tante_axis_mlp)");
  const int mt = (nmax + 15) / 16;
#define TANTE_AHW(BF, MTV) launch_axis_hw<BF, MTV>(x, S, (long)BT, nH, nW, C, wh1, bh1, wh2, bh2, ww1, bw1, ww2, bw2, lds, s)
  if (compute == TANTE_BF16) {
    switch (mt) { case 1: TANTE_AHW(true, 1); break; case 2: TANTE_AHW(true, 2); break; case 3: TANTE_AHW(true, 3); break; default: TANTE_AHW(true, 4); }
  } else {
    switch (mt) { case 1: TANTE_AHW(false, 1); break; case 2: TANTE_AHW(false, 2); break; case 3: TANTE_AHW(false, 3); break; default: TANTE_AHW(false, 4); }
  }
#undef TANTE_AHW
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_film_table(const float* t, int rows, int C, const float* sc_w0, const float* sc_b0, const float* sc_w2,
                                const float* sc_b2, const float* sh_w0, const float* sh_b0, const float* sh_w2,
                                const float* sh_b2, const float* add, float* a_out, float* b_out, void* stream) {
  if (!t || !sc_w0 || !sc_b0 || !sc_w2 || !sc_b2 || !sh_w0 || !sh_b0 || !sh_w2 || !sh_b2 || !a_out || !b_out)
    TANTE_FAIL(-1, "tante_film_table: null pointer");
  if (rows <= 0 || C <= 0 || C % 2) TANTE_FAIL(-1, "tante_film_table: bad shape");
  hipLaunchKernelGGL(film_table_kernel, dim3((rows * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, rows, C, sc_w0, sc_b0,
                     sc_w2, sc_b2, sh_w0, sh_b0, sh_w2, sh_b2, add, a_out, b_out);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_film_table_bwd(const float* t, int rows, int C, const float* sc_w0, const float* sc_b0, const float* sc_w2, const float* sh_w0,
                                    const float* sh_b0, const float* sh_w2, const float* dA, const float* dB, float* g_sc_w0, float* g_sc_b0,
                                    float* g_sc_w2, float* g_sc_b2, float* g_sh_w0, float* g_sh_b0, float* g_sh_w2, float* g_sh_b2,
                                    int accumulate, void* stream) {
  if (!t || !sc_w0 || !sc_b0 || !sc_w2 || !sh_w0 || !sh_b0 || !sh_w2 || !dA || !dB || !g_sc_w0 || !g_sc_b0 || !g_sc_w2 || !g_sc_b2 || !g_sh_w0 ||
      !g_sh_b0 || !g_sh_w2 || !g_sh_b2)
    TANTE_FAIL(-1, "tante_film_table_bwd: null pointer");
  if (rows <= 0 || rows > 8 || C <= 0 || C % 16) TANTE_FAIL(-2, "tante_film_table_bwd: rows <= 8 and C %% 16 == 0 expected (got %d x %d)", rows, C);
  hipLaunchKernelGGL(film_table_bwd_kernel, dim3((unsigned)(C / 16), 2), dim3(256), 0, (hipStream_t)stream, t, rows, C, sc_w0, sc_b0, sc_w2, sh_w0,
                     sh_b0, sh_w2, dA, dB, g_sc_w0, g_sc_b0, g_sc_w2, g_sc_b2, g_sh_w0, g_sh_b0, g_sh_w2, g_sh_b2, accumulate);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_film_apply(const float* x, int64_t x_bstride, float* y, int64_t rows, int C, int64_t rows_per,
                                const float* a, const float* b, void* stream) {
  if (!x || !y || !a || !b) TANTE_FAIL(-1, "tante_film_apply: null pointer");
  if (rows <= 0 || C <= 0 || C % 4 || rows_per <= 0 || x_bstride % 4)
    TANTE_FAIL(-1, "tante_film_apply: bad shape (C and the batch stride must be multiples of 4)");
  const long n4 = rows * (C / 4);
  hipLaunchKernelGGL(film_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)x_bstride, y, (long)rows, C / 4, (long)rows_per, a, b);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_gather_last(const float* z, int64_t n, int E, float* out, void* stream) {
  if (!z || !out || n <= 0 || E <= 0) TANTE_FAIL(-1, "tante_gather_last: bad argument");
  hipLaunchKernelGGL(gather_last_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, (long)n, E, out);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_format_input(const float* x, int64_t n_img, int T, int64_t HW, int D, float* out, int64_t out_bstride, void* stream) {
  if (!x || !out || n_img <= 0 || T <= 0 || n_img % T || HW <= 0 || D <= 0 || D > 255) TANTE_FAIL(-1, "tante_format_input: bad argument");
  if (D == 4 && HW % 4 == 0 && out_bstride % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) % 16) == 0)
    hipLaunchKernelGGL(format_input_d4_kernel, dim3((unsigned)((HW / 4 + 255) / 256), (unsigned)n_img), dim3(256), 0, (hipStream_t)stream, x,
                       (long)HW, T, out, (long)out_bstride);
  else
    hipLaunchKernelGGL(format_input_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)n_img), dim3(256), 256 * (D + 1) * sizeof(float),
                       (hipStream_t)stream, x, (long)HW, D, T, out, (long)out_bstride);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_nan_to_num(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) TANTE_FAIL(-1, "tante_nan_to_num: bad argument");
  if (((uintptr_t)x % 16) || ((uintptr_t)y % 16)) TANTE_FAIL(-2, "tante_nan_to_num: 16-byte aligned tensors expected");
  const long n4 = n / 4;
  const long blocks = std::max<long>(1, (n4 + 1023) / 1024);
  hipLaunchKernelGGL(nan_to_num_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n4, (long)n);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_taylor(const float* last, int64_t last_bstride, const float* const* derivs, int n_order, double dt,
                            int n_out, float* out, int64_t out_bstride, int64_t B, int64_t frame, void* stream) {
  if (!last || !derivs || !out) TANTE_FAIL(-1, "tante_taylor: null pointer");
  if (n_order < 1 || n_order > 8 || n_out < 1 || n_out > 8) TANTE_FAIL(-2, "tante_taylor: order and n_out must be in 1..8");
  if (frame % 4 || last_bstride % 4 || out_bstride % 4 || ((uintptr_t)last % 16) || ((uintptr_t)out % 16))
    TANTE_FAIL(-2, "tante_taylor: frame size / stride must be multiples of 4 floats, 16-byte aligned");
  TaylorArgs ta;
  for (int k = 0; k < 8; ++k) ta.d[k] = (k < n_order) ? derivs[k] : nullptr;
  for (int i = 0; i < 8; ++i) {
    double fact = 1.0;
    for (int k = 0; k < 8; ++k) {
      fact *= (double)(k + 1);
      // (i*dt)^k / k!  evaluated like the reference: python float pow, then divide (tante.py:168)
      double p = 1.0;
      for (int e = 0; e <= k; ++e) p *= (double)(i + 1) * (double)dt;
      ta.coef[i][k] = (float)(p / fact);
    }
  }
  const long n4 = B * (frame / 4);
  const dim3 grid((unsigned)((n4 + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
#define TANTE_TAYLOR(NO) \
  case NO: hipLaunchKernelGGL(taylor_kernel<NO>, grid, dim3(256), 0, s, last, (long)last_bstride, ta, n_out, out, (long)out_bstride, (long)B, (long)(frame / 4)); break;
  switch (n_order) {
    TANTE_TAYLOR(1) TANTE_TAYLOR(2) TANTE_TAYLOR(3) TANTE_TAYLOR(4)
    TANTE_TAYLOR(5) TANTE_TAYLOR(6) TANTE_TAYLOR(7) TANTE_TAYLOR(8)
  }
#undef TANTE_TAYLOR
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_rt_reduce(const float* t, int B, int L, float out_T, float ep, float* rt, void* stream) {
  if (!t || !rt || B <= 0 || L <= 0) TANTE_FAIL(-1, "tante_rt_reduce: bad argument");
  hipLaunchKernelGGL(rt_reduce_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, t, L, out_T - 1.0f, ep, rt);
  TANTE_CHECK_LAUNCH();
  return 0;
}

// ---- error string / version -----------------------------------------------------------------------
static thread_local char g_err[512] = "";
void tante_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* tante_last_error(void) { return g_err; }

// ---- tuning options (common.hip.h: tante_opt) -------------------------------------------------------------------------------------
// A small fixed table under a mutex: set rarely (start-up, A/B scripts), read once per launch by the host-side dispatch code.
namespace {
struct TanteOptEntry { char name[48]; int value; };
constexpr int TANTE_OPT_MAX = 64;
TanteOptEntry g_opts[TANTE_OPT_MAX];
int g_nopts = 0;
std::mutex g_opt_mutex;
}  // namespace
int tante_opt(const char* name, int dflt) {
  std::lock_guard<std::mutex> lk(g_opt_mutex);
  for (int i = 0; i < g_nopts; ++i)
    if (!strcmp(g_opts[i].name, name)) return g_opts[i].value;
  return dflt;
}
extern "C" int tante_set_option(const char* name, int value) {
  if (!name || !*name || strlen(name) >= sizeof(g_opts[0].name)) TANTE_FAIL(-1, "tante_set_option: bad option name");
  if (strncmp(name, "TANTE_", 6)) TANTE_FAIL(-1, "tante_set_option: option names start with TANTE_ (got '%s')", name);
  std::lock_guard<std::mutex> lk(g_opt_mutex);
  for (int i = 0; i < g_nopts; ++i)
    if (!strcmp(g_opts[i].name, name)) { g_opts[i].value = value; return 0; }
  if (g_nopts == TANTE_OPT_MAX) TANTE_FAIL(-1, "tante_set_option: table full");
  strcpy(g_opts[g_nopts].name, name);
  g_opts[g_nopts++].value = value;
  return 0;
}
extern "C" int tante_get_option(const char* name, int dflt) { return name ? tante_opt(name, dflt) : dflt; }
extern "C" int tante_abi_version(void) { return 12; }      // = tante_amd/_lib.py ABI_VERSION; bumped with every added / changed entry point
