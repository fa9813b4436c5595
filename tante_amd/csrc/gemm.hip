// Token-stationary MFMA GEMM for the TANTE path:  out = epilogue(gather(A)[M,K] @ W[N,K]^T).
//
// Every dense contraction on the path has a huge M (tokens / patches), a small K (<= 512) and a
// small N (<= 768), with weights of a few hundred KB that every workgroup needs in full.  So the
// roles are the reverse of a classic tiled GEMM:
//   * each wave keeps its 16 or 32 token rows as MFMA fragments IN REGISTERS for the whole K
//     (loaded once from HBM, normalised there when a LayerNorm is fused, converted to bf16 there);
//   * the packed weight streams through LDS in [NT rows][K_pad] tiles, double buffered, shared by
//     the 4 waves; its global image is already the swizzled LDS image, so staging is a linear copy;
//   * the MFMA "A" operand is the weight tile (rows = output features) and the "B" operand the
//     tokens, so a lane's 4 accumulator registers are 4 CONSECUTIVE output features of ONE token:
//     the epilogue (bias, GELU, residual, FiLM, pixel-shuffle scatter) stores 16 B (fp32) or 8 B
//     (bf16) per lane with no cross-lane traffic.
//
// fp32 compute uses v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate); a lane's 16-byte
// chunk holds k = 16*blk + 4*(lane>>4) + s, s = 0..3, and MFMA step s consumes element s of both
// operands -- a fixed permutation of the k order inside each 16-wide block, which a dot product does
// not care about.  bf16 compute uses v_mfma_f32_16x16x32_bf16 whose natural operand layout is
// already one 16-byte chunk (8 consecutive k) per lane.
#include "common.hip.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int nt_for_cb(int CB) { return (512 / CB) < 64 ? (512 / CB) : 64; }

struct RowInfo {
  long a_base;  // element offset of the row's first gathered element
  bool ok;
};

__device__ __forceinline__ float ld_elem(const void* a, int dtype, long idx) {
  if (dtype == TANTE_BF16) return __uint_as_float(((unsigned)((const unsigned short*)a)[idx]) << 16);
  return ((const float*)a)[idx];
}

template <int E>
__device__ __forceinline__ void ld_vec(const void* a, int dtype, long idx, float (&v)[E]) {
  if (dtype == TANTE_BF16) {
    if constexpr (E == 8) {
      u32x4 u = *(const u32x4*)((const unsigned short*)a + idx);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[2 * i] = bf16_lo(u[i]);
        v[2 * i + 1] = bf16_hi(u[i]);
      }
    } else {
      u32x2 u = *(const u32x2*)((const unsigned short*)a + idx);
      v[0] = bf16_lo(u[0]); v[1] = bf16_hi(u[0]); v[2] = bf16_lo(u[1]); v[3] = bf16_hi(u[1]);
    }
  } else {
    const float* p = (const float*)a + idx;
#pragma unroll
    for (int i = 0; i < E / 4; ++i) {
      f32x4 f = *(const f32x4*)(p + 4 * i);
      v[4 * i] = f[0]; v[4 * i + 1] = f[1]; v[4 * i + 2] = f[2]; v[4 * i + 3] = f[3];
    }
  }
}

__device__ __forceinline__ RowInfo row_info(const TanteGemm& g, int row) {
  RowInfo ri;
  ri.ok = row < g.M;
  const int r = ri.ok ? row : 0;
  if (g.a_mode == TANTE_A_LINEAR) {
    ri.a_base = (long)(r / g.a_n0) * g.a_s1 + (long)(r % g.a_n0) * g.a_s0 + g.a_off;
  } else {
    const int Wo = g.Win / g.P, Ho = g.Hin / g.P;
    const int img = r / (Ho * Wo), rem = r % (Ho * Wo);
    const int ho = rem / Wo, wo = rem % Wo;
    // images may sit in a sliding window of a longer buffer: a_n0 images per batch item, a_s1 elements between items
    const long img_off = (long)(img / g.a_n0) * g.a_s1 + (long)(img % g.a_n0) * g.Cin * g.Hin * g.Win + g.a_off;
    if (g.a_mode == TANTE_A_PATCH_NHWC)
      ri.a_base = img_off + (((long)ho * g.P) * g.Win + (long)wo * g.P) * g.Cin;
    else
      ri.a_base = img_off + ((long)ho * g.P) * g.Win + (long)wo * g.P;
  }
  return ri;
}

// E consecutive k of one gathered row (k0 = first k of the chunk); zero beyond K / beyond M.
template <int E>
__device__ __forceinline__ void load_chunk(const TanteGemm& g, const RowInfo& ri, int k0, bool a_vec, float (&v)[E]) {
#pragma unroll
  for (int i = 0; i < E; ++i) v[i] = 0.0f;
  if (!ri.ok || k0 >= g.K) return;
  const bool full = (k0 + E <= g.K);
  if (g.a_mode == TANTE_A_LINEAR) {
    if (a_vec && full) {
      ld_vec<E>(g.a, g.a_dtype, ri.a_base + k0, v);
    } else {
#pragma unroll
      for (int i = 0; i < E; ++i)
        if (k0 + i < g.K) v[i] = ld_elem(g.a, g.a_dtype, ri.a_base + k0 + i);
    }
  } else if (g.a_mode == TANTE_A_PATCH_NHWC) {
    const int seg = g.P * g.Cin;  // one kernel row = P*Cin contiguous elements
    if (a_vec && full) {
      const int kh = k0 / seg, rem = k0 % seg;
      ld_vec<E>(g.a, g.a_dtype, ri.a_base + (long)kh * g.Win * g.Cin + rem, v);
    } else {
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const int k = k0 + i;
        if (k < g.K) v[i] = ld_elem(g.a, g.a_dtype, ri.a_base + (long)(k / seg) * g.Win * g.Cin + (k % seg));
      }
    }
  } else {  // PATCH_NCHW: k = (ci, kh, kw)
    const int pp = g.P * g.P;
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int k = k0 + i;
      if (k < g.K) {
        const int ci = k / pp, kh = (k % pp) / g.P, kw = k % g.P;
        v[i] = ld_elem(g.a, g.a_dtype, ri.a_base + ((long)ci * g.Hin + kh) * g.Win + kw);
      }
    }
  }
}

__device__ __forceinline__ void store4(void* out, int dtype, long idx, const float (&v)[4]) {
  if (dtype == TANTE_BF16) {
    u32x2 u;
    u[0] = pack_bf16x2(v[0], v[1]);
    u[1] = pack_bf16x2(v[2], v[3]);
    *(u32x2*)((unsigned short*)out + idx) = u;
  } else {
    f32x4 f = {v[0], v[1], v[2], v[3]};
    *(f32x4*)((float*)out + idx) = f;
  }
}
__device__ __forceinline__ void store1(void* out, int dtype, long idx, float v) {
  if (dtype == TANTE_BF16) {
    ((__bf16*)out)[idx] = (__bf16)v;
  } else {
    ((float*)out)[idx] = v;
  }
}

struct EpiRow {  // per token row, computed once
  long o_base, r_base;
  int t, hw;  // FILM
  bool ok;
};

__device__ __forceinline__ EpiRow epi_row(const TanteGemm& g, int row) {
  EpiRow e;
  e.ok = row < g.M;
  const int r = e.ok ? row : 0;
  e.o_base = 0; e.r_base = 0; e.t = 0; e.hw = 0;
  switch (g.e_mode) {
    case TANTE_E_LINEAR:
      e.o_base = (long)r * g.out_ld;
      e.r_base = (long)r * g.res_ld;
      break;
    case TANTE_E_FILM:
      e.o_base = (long)r * g.out_ld;
      e.hw = r % g.HW;
      e.t = (r / g.HW) % g.T;
      break;
    case TANTE_E_DECONV_NHWC: {
      const int img = r / (g.Hi * g.Wi), rem = r % (g.Hi * g.Wi);
      const int hi = rem / g.Wi, wi = rem % g.Wi;
      e.o_base = ((long)img * g.Hi * g.Po + (long)hi * g.Po) * (g.Wi * g.Po) + (long)wi * g.Po;  // pixel index
    } break;
    default: {  // DECONV_NCHW
      const int img = r / (g.Hi * g.Wi), rem = r % (g.Hi * g.Wi);
      const int hi = rem / g.Wi, wi = rem % g.Wi;
      const long Ho = (long)g.Hi * g.Po, Wo = (long)g.Wi * g.Po;
      e.o_base = (long)img * g.Cout * Ho * Wo + (long)hi * g.Po * Wo + (long)wi * g.Po;
    } break;
  }
  return e;
}

// ---- compile-time variants ------------------------------------------------------------------------
// The row gather (AM), the activation and the epilogue kind (EP) are template parameters so that the
// tile loop is one short straight-line path: a first version that switched on them at run time inlined
// every combination 8x per tile (170 KB of code per kernel, instruction-fetch bound at ~5 % of MFMA peak).
enum { AM_LIN = 0, AM_NHWC = 1, AM_GEN = 2, AM_NCHW2 = 3 };
enum { EP_LIN_NONE = 0, EP_LIN_RELU, EP_LIN_GELU_TANH, EP_LIN_GELU_ERF, EP_FILM, EP_DNHWC_GELU_ERF, EP_DNCHW_NONE, EP_DNHWC_NONE, EP_GEN };

template <bool FAST>
__device__ __forceinline__ float gelu_tanh_v(float x) {
  // 0.5 x (1 + tanh u) = x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3)
  const float u2 = 1.59576912160573071176f * (x + 0.044715f * x * x * x);
  if constexpr (FAST) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-u2));
  return x / (1.0f + expf(-u2));
}

template <int E>
__device__ __forceinline__ void load_chunk_fast_lin(const TanteGemm& g, const RowInfo& ri, int k0, float (&v)[E]) {
  if (ri.ok && k0 < g.K) {
    ld_vec<E>(g.a, g.a_dtype, ri.a_base + k0, v);
  } else {
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = 0.0f;
  }
}
template <int E>
__device__ __forceinline__ void load_chunk_fast_nhwc(const TanteGemm& g, const RowInfo& ri, int k0, float (&v)[E]) {
  if (ri.ok && k0 < g.K) {
    const int seg = g.P * g.Cin;
    const int kh = k0 / seg, rem = k0 - kh * seg;
    ld_vec<E>(g.a, g.a_dtype, ri.a_base + (long)kh * g.Win * g.Cin + rem, v);
  } else {
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = 0.0f;
  }
}

// channels-first fp32 image, 2x2 patches: k = 4*ci + 2*kh + kw, so a chunk is E/4 channels x 2 rows x one
// 8-byte (kw = 0,1) load; the 16 lanes of a token group read 128 contiguous bytes per (ci, kh)
template <int E>
__device__ __forceinline__ void load_chunk_fast_nchw2(const TanteGemm& g, const RowInfo& ri, int k0, float (&v)[E]) {
#pragma unroll
  for (int i = 0; i < E; ++i) v[i] = 0.0f;
  if (!ri.ok) return;
  const float* base = (const float*)g.a + ri.a_base;
#pragma unroll
  for (int e = 0; e < E / 4; ++e) {
    const int ci = k0 / 4 + e;
    if (ci < g.Cin) {
      const float2 r0 = *(const float2*)(base + (long)ci * g.Hin * g.Win);
      const float2 r1 = *(const float2*)(base + (long)ci * g.Hin * g.Win + g.Win);
      v[4 * e] = r0.x; v[4 * e + 1] = r0.y; v[4 * e + 2] = r1.x; v[4 * e + 3] = r1.y;
    }
  }
}

// generic epilogue: every mode, scalar stores allowed (slow variant for odd shapes)
__device__ __forceinline__ void epilogue4(const TanteGemm& g, const EpiRow& e, int n0, float (&v)[4], bool out_vec) {
  if (!e.ok || n0 >= g.N) return;
  {
    const f32x4 b = *(const f32x4*)(g.bias + n0);  // bias is padded to n_pad
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j] + b[j], g.act);
  }
  switch (g.e_mode) {
    case TANTE_E_LINEAR: {
      if (out_vec) {
        if (g.residual) {
          const f32x4 r = *(const f32x4*)(g.residual + e.r_base + n0);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += r[j];
        }
        store4(g.out, g.out_dtype, e.o_base + n0, v);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n0 + j < g.N) {
            float x = v[j];
            if (g.residual) x += g.residual[e.r_base + n0 + j];
            store1(g.out, g.out_dtype, e.o_base + n0 + j, x);
          }
      }
    } break;
    case TANTE_E_FILM: {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + j < g.N) {
          const int n = n0 + j;
          v[j] = v[j] * g.film_a[(long)e.t * g.N + n] + g.film_b[(long)e.t * g.N + n] + g.s_emb[(long)e.hw * g.N + n];
        }
      if (out_vec) {
        store4(g.out, g.out_dtype, e.o_base + n0, v);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n0 + j < g.N) store1(g.out, g.out_dtype, e.o_base + n0 + j, v[j]);
      }
    } break;
    case TANTE_E_DECONV_NHWC: {
      const int Wo = g.Wi * g.Po;
      if (out_vec) {  // Cout % 4 == 0: the 4 features share (kh, kw)
        const int khw = n0 / g.Cout, co = n0 % g.Cout;
        const int kh = khw / g.Po, kw = khw % g.Po;
        const long oi = (e.o_base + (long)kh * Wo + kw) * g.Cout + co;
        if (g.dact) {     // see epilogue4_fast: act'(pre) folded into the scatter
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= act_df(ld_elem(g.dact, g.dact_dtype, oi + j), g.dact_kind);
        }
        store4(g.out, g.out_dtype, oi, v);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n0 + j < g.N) {
            const int khw = (n0 + j) / g.Cout, co = (n0 + j) % g.Cout;
            store1(g.out, g.out_dtype, (e.o_base + (long)(khw / g.Po) * Wo + (khw % g.Po)) * g.Cout + co, v[j]);
          }
      }
    } break;
    default: {  // DECONV_NCHW, n = (co, kh, kw)
      const long Ho = (long)g.Hi * g.Po, Wo = (long)g.Wi * g.Po;
      const int pp = g.Po * g.Po;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + j < g.N) {
          const int n = n0 + j, co = n / pp, kh = (n % pp) / g.Po, kw = n % g.Po;
          store1(g.out, g.out_dtype, e.o_base + (long)co * Ho * Wo + (long)kh * Wo + kw, v[j]);
        }
    } break;
  }
}

// fast epilogues: vector stores only, one mode each
template <int EP, bool BF16>
__device__ __forceinline__ void epilogue4_fast(const TanteGemm& g, const EpiRow& e, int n0, float (&v)[4]) {
  if (!e.ok || n0 >= g.N) return;
  const f32x4 b = *(const f32x4*)(g.bias + n0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x = v[j] + b[j];
    if constexpr (EP == EP_LIN_RELU) x = fmaxf(x, 0.0f);
    if constexpr (EP == EP_LIN_GELU_TANH) x = gelu_tanh_v<BF16>(x);
    if constexpr (EP == EP_LIN_GELU_ERF || EP == EP_DNHWC_GELU_ERF) x = BF16 ? gelu_poly1<false>(x) : gelu_erf_f(x);
    v[j] = x;
  }
  if constexpr (EP <= EP_LIN_GELU_ERF) {
    if (g.residual) {
      const f32x4 r = *(const f32x4*)(g.residual + e.r_base + n0);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += r[j];
    }
    store4(g.out, g.out_dtype, e.o_base + n0, v);
  } else if constexpr (EP == EP_FILM) {
    const f32x4 fa = *(const f32x4*)(g.film_a + (long)e.t * g.N + n0), fb = *(const f32x4*)(g.film_b + (long)e.t * g.N + n0);
    const f32x4 se = *(const f32x4*)(g.s_emb + (long)e.hw * g.N + n0);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] * fa[j] + fb[j] + se[j];
    store4(g.out, g.out_dtype, e.o_base + n0, v);
  } else if constexpr (EP == EP_DNHWC_GELU_ERF || EP == EP_DNHWC_NONE) {
    const int khw = n0 / g.Cout, co = n0 - khw * g.Cout;
    const int kh = khw / g.Po, kw = khw - kh * g.Po;
    const long oi = (e.o_base + (long)kh * (g.Wi * g.Po) + kw) * g.Cout + co;
    if constexpr (EP == EP_DNHWC_NONE) {
      if (g.dact) {     // training: this scatter is the data gradient of a patch conv whose INPUT was act(pre): times act'(pre), same index
        float pre[4];
        if (g.dact_dtype == TANTE_BF16) {
          const u32x2 u = *(const u32x2*)((const unsigned short*)g.dact + oi);
          pre[0] = bf16_lo(u[0]); pre[1] = bf16_hi(u[0]); pre[2] = bf16_lo(u[1]); pre[3] = bf16_hi(u[1]);
        } else {
          const f32x4 f = *(const f32x4*)((const float*)g.dact + oi);
          pre[0] = f[0]; pre[1] = f[1]; pre[2] = f[2]; pre[3] = f[3];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= (BF16 && g.dact_kind == TANTE_ACT_GELU_ERF) ? gelu_erf_grad_fast(pre[j]) : act_df(pre[j], g.dact_kind);
      }
    }
    store4(g.out, g.out_dtype, oi, v);
  } else {  // EP_DNCHW_NONE: n = (co, kh, kw); P*P consecutive n are one output pixel block of one channel
    const long Ho = (long)g.Hi * g.Po, Wo = (long)g.Wi * g.Po;
    const int pp = g.Po * g.Po;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + j < g.N) {
        const int n = n0 + j, co = n / pp, rem = n - co * pp, kh = rem / g.Po, kw = rem - kh * g.Po;
        ((float*)g.out)[e.o_base + (long)co * Ho * Wo + (long)kh * Wo + kw] = v[j];
      }
  }
}

template <bool BF16>
__device__ __forceinline__ u32x4 to_frag(const float (&v)[BF16 ? 8 : 4]) {
  u32x4 f;
  if constexpr (BF16) {
    f[0] = pack_bf16x2(v[0], v[1]);
    f[1] = pack_bf16x2(v[2], v[3]);
    f[2] = pack_bf16x2(v[4], v[5]);
    f[3] = pack_bf16x2(v[6], v[7]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __float_as_uint(v[i]);
  }
  return f;
}

template <int AM, int E>
__device__ __forceinline__ void load_chunk_any(const TanteGemm& g, const RowInfo& ri, int k0, bool a_vec, float (&v)[E]) {
  if constexpr (AM == AM_LIN) load_chunk_fast_lin<E>(g, ri, k0, v);
  else if constexpr (AM == AM_NHWC) load_chunk_fast_nhwc<E>(g, ri, k0, v);
  else if constexpr (AM == AM_NCHW2) load_chunk_fast_nchw2<E>(g, ri, k0, v);
  else load_chunk<E>(g, ri, k0, a_vec, v);
}

template <bool BF16, int CB, int TT, bool LN, int AM, int EP>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const TanteGemm g, int n_tiles, int tiles_per_split, int flags) {
  constexpr int E = BF16 ? 8 : 4;       // elements per 16-byte chunk
  constexpr int CPR = CB * 4;           // chunks per packed weight row
  constexpr int NT = nt_for_cb(CB);     // weight rows (output features) per LDS tile
  constexpr int NSUB = NT / 16;
  constexpr int TILE_U = NT * CPR;      // 16-byte units per tile
  constexpr int UPT = TILE_U / 256;     // units staged per thread
  constexpr bool PREFETCH = (CB < 32);  // K = 512 fp32: no registers left to hold a tile across the MFMAs
  static_assert(TILE_U % 256 == 0, "tile must split evenly over 256 threads");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 tiles

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, l15 = lane & 15;
  const int row0 = (blockIdx.x * 4 + wave) * (TT * 16);
  const int t_begin = blockIdx.y * tiles_per_split;
  const int t_end = min(n_tiles, t_begin + tiles_per_split);
  const bool a_vec = flags & 1, out_vec = flags & 2;

  // ---- first weight tile: issue the loads before anything else -------------------------------
  const u32x4* wsrc = (const u32x4*)g.w;
  u32x4 st[UPT];
  {
    const u32x4* p = wsrc + (size_t)t_begin * TILE_U;
#pragma unroll
    for (int u = 0; u < UPT; ++u) st[u] = p[tid + u * 256];
  }

  // ---- token fragments: this wave's TT*16 rows, whole K, kept in registers --------------------
  u32x4 xf[TT][CB];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    const RowInfo ri = row_info(g, row0 + tt * 16 + l15);
    if constexpr (LN) {  // LayerNorm without affine (gamma/beta are folded into the packed weight / bias)
      float v[CB][E];
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) load_chunk_any<AM, E>(g, ri, (cb * 4 + kk) * E, a_vec, v[cb]);
      float s = 0.0f;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < E; ++i) s += v[cb][i];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mean = s / (float)g.K;
      float q = 0.0f;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < E; ++i) {
          const float d = ((cb * 4 + kk) * E + i < g.K) ? v[cb][i] - mean : 0.0f;
          v[cb][i] = d;
          q += d * d;
        }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      const float rstd = rsqrtf(q / (float)g.K + g.ln_eps);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
        for (int i = 0; i < E; ++i) v[cb][i] *= rstd;
        xf[tt][cb] = to_frag<BF16>(v[cb]);
      }
    } else {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        float v[E];
        load_chunk_any<AM, E>(g, ri, (cb * 4 + kk) * E, a_vec, v);
        xf[tt][cb] = to_frag<BF16>(v);
      }
    }
  }

  EpiRow er[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) er[tt] = epi_row(g, row0 + tt * 16 + l15);

  {
    u32x4* d = (u32x4*)smem;
#pragma unroll
    for (int u = 0; u < UPT; ++u) d[tid + u * 256] = st[u];
  }
  __syncthreads();

  int cur = 0;
  for (int t = t_begin; t < t_end; ++t) {
    const bool more = (t + 1 < t_end);
    if (PREFETCH && more) {  // next tile's global loads fly under this tile's MFMAs
      const u32x4* p = wsrc + (size_t)(t + 1) * TILE_U;
#pragma unroll
      for (int u = 0; u < UPT; ++u) st[u] = p[tid + u * 256];
    }
    const char* wt = smem + cur * (TILE_U * 16);
    f32x4 acc[NSUB][TT];
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[ns][tt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      const int r = ns * 16 + l15;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const int phys = swz_chunk(r, cb * 4 + kk, CPR);
        const u32x4 wf = *(const u32x4*)(wt + ((r * CPR + phys) << 4));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          if constexpr (BF16) {
            acc[ns][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf),
                                                                  __builtin_bit_cast(bf16x8, xf[tt][cb]), acc[ns][tt], 0, 0, 0);
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc[ns][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wf[s]), __uint_as_float(xf[tt][cb][s]),
                                                                 acc[ns][tt], 0, 0, 0);
          }
        }
      }
    }

#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        float v[4] = {acc[ns][tt][0], acc[ns][tt][1], acc[ns][tt][2], acc[ns][tt][3]};
        if constexpr (EP == EP_GEN) epilogue4(g, er[tt], t * NT + ns * 16 + kk * 4, v, out_vec);
        else epilogue4_fast<EP, BF16>(g, er[tt], t * NT + ns * 16 + kk * 4, v);
      }

    if (more) {
      if constexpr (!PREFETCH) {
        const u32x4* p = wsrc + (size_t)(t + 1) * TILE_U;
#pragma unroll
        for (int u = 0; u < UPT; ++u) st[u] = p[tid + u * 256];
      }
      u32x4* d = (u32x4*)(smem + (cur ^ 1) * (TILE_U * 16));
#pragma unroll
      for (int u = 0; u < UPT; ++u) d[tid + u * 256] = st[u];
    }
    __syncthreads();
    cur ^= 1;
  }
}

// ---- a few hundred rows (CViT's encoder at B = 1: 256 tokens x 512) ------------------------------------------------------------------
// gemm_kernel gives such a GEMM 4 workgroups' worth of rows; split over N (launch_variant) every workgroup re-loads and re-normalises
// its 64 rows (128 KB of fp32) for ONE 32-row weight tile and then walks load -> LDS -> barrier -> MFMA -> store: 10.8 us per launch for
// 0.4 GFLOP, and the one-launch block tail (cvit_fused.hip, 16 workgroups) pulls 1.5 MB of weights through each of 16 CUs: 26 us.
// Here ONE WAVE owns a (16-row tile, 32-feature weight tile) pair and nothing is shared, staged or waited for twice: its 32 weight
// fragments (straight from the packed tile image in L2: the swizzle is an address function) and its 16 rows are all requested up front,
// the rows are normalised in registers, 32 MFMAs, the epilogue.  One memory round trip deep; M / 16 x N / 32 waves (256 - 768 at
// cfg4) spread the weight stream over every CU.  bf16 compute, K = 512, linear rows / linear epilogue only.
template <bool LN, int EP>
__global__ __launch_bounds__(64) void gemm_small_kernel(const TanteGemm g) {
  constexpr int CB = 16, CPR = 64, NT = 32, E = 8;
  const int lane = threadIdx.x, kk = lane >> 4, l15 = lane & 15;
  const int row0 = blockIdx.x * 16, t = blockIdx.y;
  // (the same fragments from a fragment-ordered copy of the weight -- whole-KiB loads instead of 16 rows x 64 bytes per instruction --
  // measured: 6.94 against 7.14 us per launch, nothing on the forward; not kept: the launch is latency, not address work)
  const char* wt = (const char*)g.w + (size_t)t * (NT * CPR * 16);
  u32x4 wf[2][CB];
#pragma unroll
  for (int ns = 0; ns < 2; ++ns) {
    const int r = ns * 16 + l15;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) wf[ns][cb] = *(const u32x4*)(wt + ((r * CPR + swz_chunk(r, cb * 4 + kk, CPR)) << 4));
  }
  const RowInfo ri = row_info(g, row0 + l15);
  const EpiRow er = epi_row(g, row0 + l15);
  // the epilogue's operands ride the same round trip: bias, and the residual rows (behind the MFMAs they were a second, dependent one)
  f32x4 bq[2], rq[2];
#pragma unroll
  for (int ns = 0; ns < 2; ++ns) {
    const int n0 = t * NT + ns * 16 + kk * 4;
    bq[ns] = n0 < g.N ? *(const f32x4*)(g.bias + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
    rq[ns] = (g.residual && er.ok && n0 < g.N) ? *(const f32x4*)(g.residual + er.r_base + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  u32x4 xf[CB];
  if constexpr (LN) {
    float v[CB][E];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) load_chunk_fast_lin<E>(g, ri, (cb * 4 + kk) * E, v[cb]);
    float s = 0.0f;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < E; ++i) s += v[cb][i];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s / (float)g.K;
    float q = 0.0f;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const float d = v[cb][i] - mean;
        v[cb][i] = d;
        q += d * d;
      }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    const float rstd = rsqrtf(q / (float)g.K + g.ln_eps);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
      for (int i = 0; i < E; ++i) v[cb][i] *= rstd;
      xf[cb] = to_frag<true>(v[cb]);
    }
  } else {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      float v[E];
      load_chunk_fast_lin<E>(g, ri, (cb * 4 + kk) * E, v);
      xf[cb] = to_frag<true>(v);
    }
  }
  f32x4 acc[2][2];      // [row tile][even / odd k-step]: four independent MFMA chains
#pragma unroll
  for (int ns = 0; ns < 2; ++ns) acc[ns][0] = acc[ns][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
      acc[ns][cb & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[ns][cb]), __builtin_bit_cast(bf16x8, xf[cb]),
                                                                acc[ns][cb & 1], 0, 0, 0);
#pragma unroll
  for (int ns = 0; ns < 2; ++ns) {
    const int n0 = t * NT + ns * 16 + kk * 4;
    if (!er.ok || n0 >= g.N) continue;
    const f32x4 a = acc[ns][0] + acc[ns][1] + bq[ns];
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // = epilogue4_fast<EP, true>: the same activation forms, the same order (act, then the residual)
      float x = a[j];
      if constexpr (EP == EP_LIN_GELU_ERF) x = gelu_poly1<false>(x);
      v[j] = x + rq[ns][j];
    }
    store4(g.out, g.out_dtype, er.o_base + n0, v);
  }
}

// -> launched?  (bf16 compute, K = 512 dense rows, vector epilogue, at most TANTE_GEMM_SMALLM rows; 0 turns the path off)
inline bool try_small(const TanteGemm& g, int n_tiles, int flags, int am, int ep, hipStream_t s) {
  const int lim = tante_opt("TANTE_GEMM_SMALLM", 1024);
  if (g.M > lim || g.K != 512 || am != AM_LIN || (flags & 3) != 3 || g.e_mode != TANTE_E_LINEAR || g.drop_p > 0.0f || g.dact) return false;
  const dim3 grid((unsigned)((g.M + 15) / 16), (unsigned)n_tiles);
  const bool ln = g.ln != 0;
#define TANTE_SM(LNV, EPV) do { hipLaunchKernelGGL((gemm_small_kernel<LNV, EPV>), grid, dim3(64), 0, s, g); return true; } while (0)
  if (ln && ep == EP_LIN_NONE) TANTE_SM(true, EP_LIN_NONE);
  if (ln && ep == EP_LIN_GELU_ERF) TANTE_SM(true, EP_LIN_GELU_ERF);
  if (!ln && ep == EP_LIN_NONE) TANTE_SM(false, EP_LIN_NONE);
  if (!ln && ep == EP_LIN_GELU_ERF) TANTE_SM(false, EP_LIN_GELU_ERF);
#undef TANTE_SM
  return false;
}

// ---- dense bf16 GEMM for the training path (M = every token of the batch) ---------------------------------------------------------
// At M ~ 16-32 k tokens and K, N of a few hundred the job is a 25 MB stream with < 2 us of MFMA work, and gemm_kernel's 186 VGPRs +
// 64 KB of LDS hold two workgroups per CU: a grid of several hundred workgroups then runs as 1.5 serial load -> MFMA -> store rounds.
// This variant keeps everything that costs residency out of the workgroup - the weight tile arrives by LDS-DMA (no staging registers,
// one 32 KB buffer), the token fragments load as raw bf16 - so four workgroups share a CU and the whole grid is resident at once
// (train step 48.0 -> 45.7 ms).  What is left is lockstep: with one resident round every workgroup loads, then every workgroup
// stores, so HBM reads never overlap HBM writes (24576 x 256 x 256, rocprofv3: 4.9 us empty kernel + 3.5 us loads + 1.4 us MFMA +
// 4 us stores = 13.6 us).  A persistent variant (one resident tile per workgroup, row groups prefetched under the previous group's
// stores) was measured slower (14.3 / 30-39 us at N = 256 / 768): every tile's workgroup re-reads the token rows through L2.
// training epilogues of the lite kernel (TanteGemm.drop_p / .dact): 0 none, 1 dropout before the residual add, 2 times act'(dact)
template <int TR>
__device__ __forceinline__ void epilogue4_train(const TanteGemm& g, const EpiRow& e, long row, int n0, float (&v)[4]) {
  if (!e.ok || n0 >= g.N) return;
  const f32x4 b = *(const f32x4*)(g.bias + n0);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] += b[j];
  if constexpr (TR == 1) {
    const float ks = 1.0f / (1.0f - g.drop_p);
    const unsigned long long i0 = (unsigned long long)row * g.N + n0;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = dropout_keep(g.drop_seed, i0 + j, g.drop_p) ? v[j] * ks : 0.0f;
    if (g.residual) {
      const f32x4 r = *(const f32x4*)(g.residual + e.r_base + n0);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += r[j];
    }
  } else {
    float pre[4];
    if (g.dact_dtype == TANTE_BF16) {
      const u32x2 u = *(const u32x2*)((const unsigned short*)g.dact + row * g.N + n0);
      pre[0] = bf16_lo(u[0]); pre[1] = bf16_hi(u[0]); pre[2] = bf16_lo(u[1]); pre[3] = bf16_hi(u[1]);
    } else {
      const f32x4 f = *(const f32x4*)((const float*)g.dact + row * g.N + n0);
      pre[0] = f[0]; pre[1] = f[1]; pre[2] = f[2]; pre[3] = f[3];
    }
    if (g.dact_kind == TANTE_ACT_GELU_TANH) {   // bf16 path: tanh through one exp and one reciprocal (libm's tanhf is ~40 instructions)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x = pre[j], c = 0.79788456080286535588f, u = c * (x + 0.044715f * x * x * x);
        const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));
        v[j] *= 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * c * (1.0f + 3.0f * 0.044715f * x * x);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= act_df(pre[j], g.dact_kind);
    }
  }
  store4(g.out, g.out_dtype, e.o_base + n0, v);
}

// Token fragments straight from a channels-FIRST image (TANTE_A_PATCH_NCHW, kernel P, stride P, padding a_pad = 0 or -- P = 4 only -- the
// 'same' padding 1 of enc_dec_cnn.py:66-81, under which the conv still has H / P x W / P outputs and only the top row / left column of
// the patch grid reach outside the image): lane (token l15, k-chunk q) needs 8 consecutive k = (ci, kh, kw) of its patch -- at P = 4 two
// image rows of four pixels of channel q / 2, at P = 2 two rows of two pixels of channels 2 q and 2 q + 1 -- and the 16 tokens of a wave
// are 16 neighbouring patches, so every load instruction covers whole 128-byte (bf16, P = 4) runs of four image rows.  With the padding
// the four pixels start one to the left of an aligned quad: the lane loads its own quad and the quad before it (the neighbour lane's,
// an L1 hit) and funnel-shifts.  Values are rounded to bf16 exactly as tante_im2col's gather rounds them, and the k order is the packed
// weight's, so the product is bit-identical to im2col + the dense kernel -- without the (M, K) matrix, which at cfg5's first conv stage
// (8 images x 32 ch x 512 x 512 -> 131072 x 512) was a 134 MB write and a 134 MB read around a 118 us gather.
// AM: 1 / 2 = P 4 from bf16 / fp32, 3 / 4 = P 2 from bf16 / fp32.
struct PatchRow {
  long org;          // element offset of pixel (ho P, wo P) of channel 0
  bool ok, top, left;
};

__device__ __forceinline__ PatchRow patch_row(const TanteGemm& g, int row) {
  PatchRow pr;
  pr.ok = row < g.M;
  const int r = pr.ok ? row : 0;
  const int Wo = g.Win / g.P, Ho = g.Hin / g.P;
  const int img = r / (Ho * Wo), rem = r % (Ho * Wo);
  const int ho = rem / Wo, wo = rem % Wo;
  pr.org = (long)(img / g.a_n0) * g.a_s1 + (long)(img % g.a_n0) * g.Cin * g.Hin * g.Win + g.a_off + ((long)ho * g.P) * g.Win + (long)wo * g.P;
  pr.top = g.a_pad != 0 && ho == 0;
  pr.left = g.a_pad != 0 && wo == 0;
  return pr;
}

template <int AM>
__device__ __forceinline__ u32x4 patch_frag(const TanteGemm& g, const PatchRow& pr, int q) {
  const long plane = (long)g.Hin * g.Win;
  if constexpr (AM == 1 || AM == 2) {
    const int kh0 = (q & 1) * 2;
    const long o = pr.org + (long)(q >> 1) * plane + (long)(kh0 - g.a_pad) * g.Win;
    const bool skip0 = pr.top && kh0 == 0;      // image row -1
    if constexpr (AM == 1) {
      const unsigned short* p = (const unsigned short*)g.a + o;
      u32x2 r0 = {0u, 0u}, r1;
      if (!skip0) r0 = *(const u32x2*)p;
      r1 = *(const u32x2*)(p + g.Win);
      if (g.a_pad) {
        u32x2 l0 = {0u, 0u}, l1 = {0u, 0u};
        if (!pr.left) {
          if (!skip0) l0 = *(const u32x2*)(p - 4);
          l1 = *(const u32x2*)(p + g.Win - 4);
        }
        return u32x4{(l0[1] >> 16) | (r0[0] << 16), (r0[0] >> 16) | (r0[1] << 16), (l1[1] >> 16) | (r1[0] << 16), (r1[0] >> 16) | (r1[1] << 16)};
      }
      return u32x4{r0[0], r0[1], r1[0], r1[1]};
    } else {
      const float* p = (const float*)g.a + o;
      f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1;
      if (!skip0) r0 = *(const f32x4*)p;
      r1 = *(const f32x4*)(p + g.Win);
      if (g.a_pad) {
        float l0 = 0.0f, l1 = 0.0f;
        if (!pr.left) {
          if (!skip0) l0 = p[-1];
          l1 = p[g.Win - 1];
        }
        return u32x4{pack_bf16x2(l0, r0[0]), pack_bf16x2(r0[1], r0[2]), pack_bf16x2(l1, r1[0]), pack_bf16x2(r1[1], r1[2])};
      }
      return u32x4{pack_bf16x2(r0[0], r0[1]), pack_bf16x2(r0[2], r0[3]), pack_bf16x2(r1[0], r1[1]), pack_bf16x2(r1[2], r1[3])};
    }
  } else {
    const long o = pr.org + (long)(q * 2) * plane;
    if constexpr (AM == 3) {
      const unsigned short* p = (const unsigned short*)g.a + o;
      return u32x4{*(const unsigned*)p, *(const unsigned*)(p + g.Win), *(const unsigned*)(p + plane), *(const unsigned*)(p + plane + g.Win)};
    } else {
      const float* p = (const float*)g.a + o;
      const f32x2 a = *(const f32x2*)p, b = *(const f32x2*)(p + g.Win), c = *(const f32x2*)(p + plane), d = *(const f32x2*)(p + plane + g.Win);
      return u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(b[0], b[1]), pack_bf16x2(c[0], c[1]), pack_bf16x2(d[0], d[1])};
    }
  }
}

// what the fragment-load form needs (tante_gemm refuses a_pad != 0 elsewhere: no other path implements the padding)
inline bool patch_lite_ok(const TanteGemm& g, int flags, int cb) {
  const bool e_lin = (flags & 2) && g.e_mode == TANTE_E_LINEAR;
  const bool e_cf = g.e_mode == TANTE_E_DECONV_NCHW && g.Po == 1 && g.out_dtype == TANTE_F32 && g.N % 4 == 0 && ((uintptr_t)g.out % 4) == 0;
  return g.a_mode == TANTE_A_PATCH_NCHW && g.compute == TANTE_BF16 && !g.ln && g.drop_p <= 0.0f && !g.dact && (e_lin || e_cf) &&
         (cb == 8 || cb == 16) && g.K == cb * 32 && g.M >= 4096 && (g.P == 2 || g.P == 4) && ((uintptr_t)g.a % 16) == 0 &&
         (g.a_dtype == TANTE_BF16 || g.a_dtype == TANTE_F32) && (g.a_pad == 0 || (g.a_pad == 1 && g.P == 4)) &&
         (g.act == TANTE_ACT_NONE || g.act == TANTE_ACT_GELU_ERF);
}

// channels-FIRST output of a patch stage (e_mode DECONV_NCHW with Po = 1: out[img][n][ho][wo], fp32): a lane holds 4 channels of one token and
// the 16 lanes of a group hold 16 neighbouring tokens, so every store instruction writes four 64-byte runs; the four waves of the workgroup
// own the next 16 tokens each and L2 merges their runs before they leave for HBM.  (TR = 3: no activation, TR = 4: exact GELU.)
template <int TR>
__device__ __forceinline__ void epilogue4_nchw(const TanteGemm& g, const EpiRow& e, int n0, float (&v)[4]) {
  if (!e.ok || n0 >= g.N) return;
  const f32x4 b = *(const f32x4*)(g.bias + n0);
  const long hw = (long)g.Hi * g.Wi;
  float* o = (float*)g.out + e.o_base + (long)n0 * hw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x = v[j] + b[j];
    if constexpr (TR == 4) x = gelu_poly1<false>(x);
    o[j * hw] = x;
  }
}

// pixel-shuffle output of a kernel-2 stride-2 transposed conv, channels first (e_mode DECONV_NCHW, Po = 2; fp32): the lane's four
// consecutive n are one channel's 2 x 2 output block of its token -- two 8-byte stores, and the 16 tokens of a lane group are 16
// neighbouring input pixels, so each store instruction writes 128-byte runs.  (TR = 5: no activation, TR = 6: exact GELU.)  The generic
// kernel's scalar scatter took 44.6 us for cfg5's first decoder stage (32768 x 256 -> 512, 67 MB out).
template <int TR>
__device__ __forceinline__ void epilogue4_dnchw2(const TanteGemm& g, const EpiRow& e, int n0, float (&v)[4]) {
  if (!e.ok || n0 >= g.N) return;
  const f32x4 b = *(const f32x4*)(g.bias + n0);
  const long Wo = 2L * g.Wi, plane = 2L * g.Hi * Wo;
  float* o = (float*)g.out + e.o_base + (long)(n0 >> 2) * plane;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] += b[j];
    if constexpr (TR == 6) v[j] = gelu_erf_fast(v[j]);      // |error| <= 1.5e-7 against erff (the generic epilogue's): the output is an fp32 image
  }
  *(f32x2*)o = f32x2{v[0], v[1]};
  *(f32x2*)(o + Wo) = f32x2{v[2], v[3]};
}

template <int CB, int TT, int EP, int TR = 0, int AM = 0>
__global__ __launch_bounds__(256, 4) void gemm_lite_kernel(const TanteGemm g, int n_tiles, int tiles_per_split) {
  constexpr int CPR = CB * 4, NT = nt_for_cb(CB), NSUB = NT / 16, TILE_U = NT * CPR, UPT = TILE_U / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // one tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 4, l15 = lane & 15;
  const int row0 = (blockIdx.x * 4 + wave) * (TT * 16);
  const int t_begin = blockIdx.y * tiles_per_split;
  const int t_end = min(n_tiles, t_begin + tiles_per_split);
  const u32x4* wsrc = (const u32x4*)g.w;
  auto stage = [&](int t) {   // packed tiles are stored in their LDS image: a lane-linear copy
    const u32x4* p = wsrc + (size_t)t * TILE_U + wave * 64 + lane;
#pragma unroll
    for (int u = 0; u < UPT; ++u)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + u * 256),
                                       (__attribute__((address_space(3))) void*)(smem + (u * 256 + wave * 64) * 16), 16, 0, 0);
  };
  stage(t_begin);

  u32x4 xf[TT][CB];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    if constexpr (AM == 0 || AM == 5) {
      const RowInfo ri = row_info(g, row0 + tt * 16 + l15);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        xf[tt][cb] = u32x4{0u, 0u, 0u, 0u};
        if (ri.ok) {
          if constexpr (AM == 0) {
            xf[tt][cb] = *(const u32x4*)((const unsigned short*)g.a + ri.a_base + (cb * 4 + kk) * 8);
          } else {      // fp32 rows (the residual stream), rounded to bf16 as the staged kernels round them
            const float* p = (const float*)g.a + ri.a_base + (cb * 4 + kk) * 8;
            const f32x4 lo = *(const f32x4*)p, hi = *(const f32x4*)(p + 4);
            xf[tt][cb] = u32x4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
          }
        }
      }
    } else {
      const PatchRow pr = patch_row(g, row0 + tt * 16 + l15);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        xf[tt][cb] = u32x4{0u, 0u, 0u, 0u};
        if (pr.ok) xf[tt][cb] = patch_frag<AM>(g, pr, cb * 4 + kk);
      }
    }
  }
  EpiRow er[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) er[tt] = epi_row(g, row0 + tt * 16 + l15);

  for (int t = t_begin; t < t_end; ++t) {
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): this wave's slice of the tile (and its token rows) have landed
    __syncthreads();
    f32x4 acc[NSUB][TT];
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) acc[ns][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      const int r = ns * 16 + l15;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        const u32x4 wf = *(const u32x4*)(smem + ((r * CPR + swz_chunk(r, cb * 4 + kk, CPR)) << 4));
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
          acc[ns][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf[tt][cb]),
                                                                acc[ns][tt], 0, 0, 0);
      }
    }
    if (t + 1 < t_end) {
      __syncthreads();     // every wave has read the tile
      stage(t + 1);        // the next tile flies under this tile's epilogue
    }
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        float v[4] = {acc[ns][tt][0], acc[ns][tt][1], acc[ns][tt][2], acc[ns][tt][3]};
        if constexpr (TR == 0) epilogue4_fast<EP, true>(g, er[tt], t * NT + ns * 16 + kk * 4, v);
        else if constexpr (TR >= 5) epilogue4_dnchw2<TR>(g, er[tt], t * NT + ns * 16 + kk * 4, v);
        else if constexpr (TR >= 3) epilogue4_nchw<TR>(g, er[tt], t * NT + ns * 16 + kk * 4, v);
        else epilogue4_train<TR>(g, er[tt], row0 + tt * 16 + l15, t * NT + ns * 16 + kk * 4, v);
      }
  }
}

template <int CB, int EP, int TR = 0, int AM = 0>
void launch_lite(const TanteGemm& g, int n_tiles, hipStream_t s) {
  constexpr int TT = (CB <= 8) ? 2 : 1;
  constexpr int NT = nt_for_cb(CB);
  const int gx = (g.M + 4 * TT * 16 - 1) / (4 * TT * 16);
  // one resident round: 4 workgroups x 256 CUs; every N-split re-reads the token rows (through L2), so stop at the first split that fills it
  int nsplit = 1;
  for (int d = 1; d <= n_tiles; ++d)
    if (n_tiles % d == 0) { nsplit = d; if ((long)gx * d >= 768) break; }
  const size_t lds = (size_t)NT * CB * 4 * 16;
  hipLaunchKernelGGL((gemm_lite_kernel<CB, TT, EP, TR, AM>), dim3(gx, nsplit), dim3(256), lds, s, g, n_tiles, n_tiles / nsplit);
}

template <int CB, int AM>
bool lite_patch(const TanteGemm& g, int n_tiles, hipStream_t s) {
  if (g.e_mode == TANTE_E_DECONV_NCHW) {      // channels-first output (Po = 1)
    if (g.act == TANTE_ACT_NONE) launch_lite<CB, EP_LIN_NONE, 3, AM>(g, n_tiles, s);
    else launch_lite<CB, EP_LIN_NONE, 4, AM>(g, n_tiles, s);
    return true;
  }
  switch (g.act) {
    case TANTE_ACT_NONE: launch_lite<CB, EP_LIN_NONE, 0, AM>(g, n_tiles, s); return true;
    case TANTE_ACT_GELU_ERF: launch_lite<CB, EP_LIN_GELU_ERF, 0, AM>(g, n_tiles, s); return true;
    default: return false;
  }
}

template <int CB>
bool try_lite(const TanteGemm& g, int n_tiles, int flags, hipStream_t s) {
  const bool off = tante_opt("TANTE_GEMM_NO_LITE", 0) != 0;
  if (g.a_mode == TANTE_A_PATCH_NCHW) {     // the patch gather of a channels-first image as the fragment load (patch_frag)
    if ((off && !g.a_pad) || !patch_lite_ok(g, flags, CB)) return false;
    if constexpr (CB >= 8) {
      const bool b16 = g.a_dtype == TANTE_BF16;
      if (g.P == 4) return b16 ? lite_patch<CB, 1>(g, n_tiles, s) : lite_patch<CB, 2>(g, n_tiles, s);
      return b16 ? lite_patch<CB, 3>(g, n_tiles, s) : lite_patch<CB, 4>(g, n_tiles, s);
    }
    return false;
  }
  if (g.a_mode == TANTE_A_LINEAR && g.e_mode == TANTE_E_DECONV_NCHW) {      // kernel-2 transposed conv, channels-first pixel shuffle (epilogue4_dnchw2)
    if (off || g.ln || g.drop_p > 0.0f || g.dact || g.Po != 2 || g.out_dtype != TANTE_F32 || !(flags & 1) || g.K != CB * 32 || g.M < 4096 ||
        ((uintptr_t)g.out % 8) != 0 || (g.act != TANTE_ACT_NONE && g.act != TANTE_ACT_GELU_ERF) || (g.a_dtype != TANTE_BF16 && g.a_dtype != TANTE_F32))
      return false;
    const bool gelu = g.act == TANTE_ACT_GELU_ERF;
    if (g.a_dtype == TANTE_BF16) { if (gelu) launch_lite<CB, EP_LIN_NONE, 6, 0>(g, n_tiles, s); else launch_lite<CB, EP_LIN_NONE, 5, 0>(g, n_tiles, s); }
    else { if (gelu) launch_lite<CB, EP_LIN_NONE, 6, 5>(g, n_tiles, s); else launch_lite<CB, EP_LIN_NONE, 5, 5>(g, n_tiles, s); }
    return true;
  }
  if ((off && g.drop_p <= 0.0f && !g.dact) || g.ln || g.a_mode != TANTE_A_LINEAR || g.a_dtype != TANTE_BF16 || (flags & 3) != 3 || g.e_mode != TANTE_E_LINEAR) return false;
  if (g.K != CB * 32 || g.M < 4096) return false;   // whole 32-wide k blocks: the raw fragment loads have no K tail
  if (g.drop_p > 0.0f || g.dact) {
    if (g.act != TANTE_ACT_NONE || (g.drop_p > 0.0f && g.dact)) return false;
    if (g.drop_p > 0.0f) launch_lite<CB, EP_LIN_NONE, 1>(g, n_tiles, s);
    else launch_lite<CB, EP_LIN_NONE, 2>(g, n_tiles, s);
    return true;
  }
  switch (g.act) {
    case TANTE_ACT_NONE: launch_lite<CB, EP_LIN_NONE>(g, n_tiles, s); return true;
    case TANTE_ACT_RELU: launch_lite<CB, EP_LIN_RELU>(g, n_tiles, s); return true;
    case TANTE_ACT_GELU_TANH: launch_lite<CB, EP_LIN_GELU_TANH>(g, n_tiles, s); return true;
    case TANTE_ACT_GELU_ERF: launch_lite<CB, EP_LIN_GELU_ERF>(g, n_tiles, s); return true;
    default: return false;
  }
}

// ---- weight packing ----------------------------------------------------------------------------
__device__ __forceinline__ long w_src_index(int layout, int n, int k, int N, int K, int P, int Co) {
  switch (layout) {
    case TANTE_W_CONV_NHWC: {  // (Cout, Cin, P, P); k = (kh, kw, ci), Co = Cin
      const int kh = k / (P * Co), kw = (k / Co) % P, ci = k % Co;
      return (((long)n * Co + ci) * P + kh) * P + kw;
    }
    case TANTE_W_DECONV_NHWC: {  // (Cin, Cout, P, P); n = (kh, kw, co), Co = Cout
      const int kh = n / (P * Co), kw = (n / Co) % P, co = n % Co;
      return (((long)k * Co + co) * P + kh) * P + kw;
    }
    case TANTE_W_DECONV_NCHW:  // (Cin, Cout, P, P); n = (co, kh, kw)
      return (long)k * N + n;
    case TANTE_W_LINEAR_T:  // dgrad of a linear / of a conv read as (ci,kh,kw): packed[n'][k'] = W[k'][n'], W is (K, N)
      return (long)k * N + n;
    case TANTE_W_CONV_NHWC_T: {  // dgrad of a CONV_NHWC stage: packed[n' = (kh,kw,ci)][k' = co] = W[co][ci][kh][kw], Co = Cin
      const int kh = n / (P * Co), kw = (n / Co) % P, ci = n % Co;
      return (((long)k * Co + ci) * P + kh) * P + kw;
    }
    case TANTE_W_DECONV_NHWC_T: {  // dgrad of a DECONV_NHWC stage: packed[n' = ci][k' = (kh,kw,co)] = W[ci][co][kh][kw], Co = Cout
      const int kh = k / (P * Co), kw = (k / Co) % P, co = k % Co;
      return (((long)n * Co + co) * P + kh) * P + kw;
    }
    case TANTE_W_DECONV_NCHW_T:  // dgrad of the DECONV_NCHW stage: packed[n' = ci][k' = (co,kh,kw)] = W[ci][k']  (native order)
      return (long)n * K + k;
    default:
      return (long)n * K + k;
  }
}

__device__ __forceinline__ void pack_bias_row(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ beta,
                                              int layout, int N, int K, int P, int Co, int n, float* __restrict__ out);

// blocks [0, w_blocks) pack the weight; the blocks after them the bias (+ W beta of a folded LayerNorm): one launch per packing --
// a train step re-packs ~90 weights, and every launch costs 4-5 us of GPU time however little it does
template <bool BF16>
__global__ void pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma, int layout, int N, int K, int P,
                            int Co, int n_pad, int cpr, int nt, u32x4* __restrict__ out, int w_blocks, const float* __restrict__ bias,
                            const float* __restrict__ beta, float* __restrict__ bias_out) {
  constexpr int E = BF16 ? 8 : 4;
  if ((int)blockIdx.x >= w_blocks) {
    const int n = ((int)blockIdx.x - w_blocks) * blockDim.x + threadIdx.x;
    if (n < n_pad) pack_bias_row(w, bias, beta, layout, N, K, P, Co, n, bias_out);
    return;
  }
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_pad * cpr) return;
  const int n = (int)(idx / cpr), c = (int)(idx % cpr);
  float v[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const int k = c * E + i;
    float x = 0.0f;
    if (n < N && k < K) {
      x = w[w_src_index(layout, n, k, N, K, P, Co)];
      if (gamma) x *= gamma[k];
    }
    v[i] = x;
  }
  u32x4 o;
  if constexpr (BF16) {
    o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]);
    o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
  } else {
    o[0] = __float_as_uint(v[0]); o[1] = __float_as_uint(v[1]); o[2] = __float_as_uint(v[2]); o[3] = __float_as_uint(v[3]);
  }
  const int tile = n / nt, r = n % nt;
  out[((long)tile * nt + r) * cpr + swz_chunk(r, c, cpr)] = o;
}

__device__ __forceinline__ void pack_bias_row(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ beta,
                                              int layout, int N, int K, int P, int Co, int n, float* __restrict__ out) {
  float b = 0.0f;
  if (n < N) {
    if (bias) {
      int bi = n;
      if (layout == TANTE_W_DECONV_NHWC) bi = n % Co;
      else if (layout == TANTE_W_DECONV_NCHW) bi = n / (P * P);
      b = bias[bi];
    }
    if (beta) {
      float s = 0.0f;
      for (int k = 0; k < K; ++k) s += w[w_src_index(layout, n, k, N, K, P, Co)] * beta[k];
      b += s;
    }
  }
  out[n] = b;
}

template <bool BF16, int CB, bool LN, int AM, int EP>
void launch_variant(const TanteGemm& g, int n_tiles, int flags, hipStream_t s) {
  constexpr int TT = (CB <= 8) ? 2 : 1;
  constexpr int NT = nt_for_cb(CB);
  const int rows_per_wg = 4 * TT * 16;
  const int gx = (g.M + rows_per_wg - 1) / rows_per_wg;
  int nsplit = 1;
  const int target = tante_opt("TANTE_GEMM_WGS", 512);
  // every N-split re-loads (and re-normalises) the token rows, so split only while the grid is short of WGs
  while (nsplit < n_tiles && (long)gx * nsplit < target && (n_tiles % (nsplit * 2) == 0)) nsplit *= 2;
  // a few hundred rows (CViT's encoder at B = 1: 256 tokens, gx = 4, 24 tiles): doubling stops at 8 splits = 32 workgroups with three
  // tiles each in a row; any divisor will do -- every output tile is one workgroup's either way
  if ((long)gx * nsplit * 2 <= target)
    for (int d = nsplit + 1; d <= n_tiles; ++d)
      if (n_tiles % d == 0) { nsplit = d; if ((long)gx * d >= target) break; }
  const int per = (n_tiles + nsplit - 1) / nsplit;
  const size_t lds = 2 * (size_t)NT * CB * 4 * 16;
  auto kern = gemm_kernel<BF16, CB, TT, LN, AM, EP>;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(gx, nsplit), dim3(256), lds, s, g, n_tiles, per, flags);
}

// the variants the TANTE path uses get a dedicated straight-line kernel; anything else runs the generic one
template <bool BF16, int CB>
void launch_gemm(const TanteGemm& g, int n_tiles, int flags, hipStream_t s) {
  const bool a_vec = flags & 1, out_vec = flags & 2;
  const int E = BF16 ? 8 : 4;
  const bool k_ok = (g.K % E) == 0;
  int am = AM_GEN;
  if (a_vec && k_ok && g.a_mode == TANTE_A_LINEAR) am = AM_LIN;
  if (a_vec && k_ok && g.a_mode == TANTE_A_PATCH_NHWC) am = AM_NHWC;
  if (g.a_mode == TANTE_A_PATCH_NCHW && g.P == 2 && g.a_dtype == TANTE_F32 && g.Win % 2 == 0 && ((uintptr_t)g.a % 8) == 0) am = AM_NCHW2;
  int ep = EP_GEN;
  if (out_vec && g.e_mode == TANTE_E_LINEAR) {
    ep = g.act == TANTE_ACT_NONE ? EP_LIN_NONE : g.act == TANTE_ACT_RELU ? EP_LIN_RELU
       : g.act == TANTE_ACT_GELU_TANH ? EP_LIN_GELU_TANH : EP_LIN_GELU_ERF;
  } else if (out_vec && g.e_mode == TANTE_E_FILM && g.act == TANTE_ACT_NONE) {
    ep = EP_FILM;
  } else if (out_vec && g.e_mode == TANTE_E_DECONV_NHWC && g.act == TANTE_ACT_GELU_ERF) {
    ep = EP_DNHWC_GELU_ERF;
  } else if (out_vec && g.e_mode == TANTE_E_DECONV_NHWC && g.act == TANTE_ACT_NONE) {
    ep = EP_DNHWC_NONE;
  } else if (g.e_mode == TANTE_E_DECONV_NCHW && g.act == TANTE_ACT_NONE) {
    ep = EP_DNCHW_NONE;
  }
  const bool ln = g.ln != 0;
  if constexpr (BF16 && CB == 16) {
    if (try_small(g, n_tiles, flags, am, ep, s)) return;
  }
  if constexpr (BF16 && CB >= 4 && CB <= 16) {
    if (try_lite<CB>(g, n_tiles, flags, s)) return;
  }
#define TANTE_V(LNV, AMV, EPV) launch_variant<BF16, CB, LNV, AMV, EPV>(g, n_tiles, flags, s)
  if (ln && am == AM_LIN && ep == EP_LIN_NONE) return TANTE_V(true, AM_LIN, EP_LIN_NONE);            // LN + QKV
  if (ln && am == AM_LIN && ep == EP_LIN_GELU_TANH) return TANTE_V(true, AM_LIN, EP_LIN_GELU_TANH);  // LN + fc1 + GELU
  if (ln && am == AM_LIN && ep == EP_LIN_GELU_ERF) return TANTE_V(true, AM_LIN, EP_LIN_GELU_ERF);    // CViT: LN + fc1 + exact GELU (cvit.py:49-58)
  if (!ln && am == AM_LIN && ep == EP_LIN_NONE) return TANTE_V(false, AM_LIN, EP_LIN_NONE);          // out-proj / fc2 (+res)
  if (!ln && am == AM_LIN && ep == EP_LIN_RELU) return TANTE_V(false, AM_LIN, EP_LIN_RELU);          // interprator
  if (!ln && am == AM_LIN && ep == EP_LIN_GELU_ERF) return TANTE_V(false, AM_LIN, EP_LIN_GELU_ERF);  // CViT output MLP: x + gelu(dense(x)) (cvit.py Mlp)
  if (!ln && am == AM_LIN && ep == EP_LIN_GELU_TANH) return TANTE_V(false, AM_LIN, EP_LIN_GELU_TANH);
  if (!ln && am == AM_NCHW2 && ep == EP_LIN_GELU_ERF) return TANTE_V(false, AM_NCHW2, EP_LIN_GELU_ERF);  // patch embed 1 (NCHW)
  if (!ln && am == AM_NCHW2 && ep == EP_LIN_NONE) return TANTE_V(false, AM_NCHW2, EP_LIN_NONE);        // train path: stage 1 pre-activation
  if (am == AM_NCHW2) am = AM_GEN;
  if (!ln && am == AM_NHWC && ep == EP_LIN_GELU_ERF) return TANTE_V(false, AM_NHWC, EP_LIN_GELU_ERF);  // patch embed 2
  if (!ln && am == AM_NHWC && ep == EP_FILM) return TANTE_V(false, AM_NHWC, EP_FILM);                // patch embed 3 + FiLM
  if (!ln && am == AM_LIN && ep == EP_DNHWC_GELU_ERF) return TANTE_V(false, AM_LIN, EP_DNHWC_GELU_ERF);  // heads 1, 2
  if (!ln && am == AM_LIN && ep == EP_DNCHW_NONE) return TANTE_V(false, AM_LIN, EP_DNCHW_NONE);      // head 3
  // the train path's pre-activation stages and their data gradients (PatchEmbedFn / DeconvFn forward + backward)
  if (!ln && am == AM_LIN && ep == EP_DNHWC_NONE) return TANTE_V(false, AM_LIN, EP_DNHWC_NONE);
  if (!ln && am == AM_NHWC && ep == EP_LIN_NONE) return TANTE_V(false, AM_NHWC, EP_LIN_NONE);
  if (ln) return TANTE_V(true, AM_GEN, EP_GEN);
  return TANTE_V(false, AM_GEN, EP_GEN);
#undef TANTE_V
}

}  // namespace

extern "C" int tante_pack_geom(int N, int K, int compute, TantePackGeom* out) {
  if (!out || N <= 0 || K <= 0) TANTE_FAIL(-1, "tante_pack_geom: bad argument");
  const int kb = (compute == TANTE_BF16) ? 32 : 16;
  const int need = (K + kb - 1) / kb;
  int cb = 2;
  while (cb < need) cb *= 2;
  const int cb_max = (compute == TANTE_BF16) ? 16 : 32;
  if (cb > cb_max) TANTE_FAIL(-2, "tante_pack_geom: K=%d exceeds the register-stationary limit (512)", K);
  out->cb = cb;
  out->k_pad = cb * kb;
  out->nt = nt_for_cb(cb);
  out->n_pad = (N + out->nt - 1) / out->nt * out->nt;
  out->bytes = (int64_t)out->n_pad * out->k_pad * ((compute == TANTE_BF16) ? 2 : 4);
  return 0;
}

extern "C" int tante_pack_weight(const float* w, const float* bias, const float* gamma, const float* beta, int layout,
                                 int N, int K, int P, int C_other, int compute, void* w_out, float* bias_out,
                                 void* stream) {
  TantePackGeom geo;
  int rc = tante_pack_geom(N, K, compute, &geo);
  if (rc) return rc;
  if (!w || !w_out || !bias_out) TANTE_FAIL(-1, "tante_pack_weight: null pointer");
  const bool needs_geom = layout == TANTE_W_CONV_NHWC || layout == TANTE_W_DECONV_NHWC || layout == TANTE_W_DECONV_NCHW ||
                          layout == TANTE_W_CONV_NHWC_T || layout == TANTE_W_DECONV_NHWC_T;
  if (layout < 0 || layout > TANTE_W_DECONV_NCHW_T) TANTE_FAIL(-1, "tante_pack_weight: bad layout %d", layout);
  if (needs_geom && (P <= 0 || C_other <= 0)) TANTE_FAIL(-1, "tante_pack_weight: conv layout needs P, C_other");
  if (layout >= TANTE_W_LINEAR_T && bias) TANTE_FAIL(-1, "tante_pack_weight: a data-gradient packing has no bias");
  if ((gamma || beta) && layout != TANTE_W_LINEAR) TANTE_FAIL(-1, "tante_pack_weight: LayerNorm fold only for linear weights");
  hipStream_t s = (hipStream_t)stream;
  const int cpr = geo.cb * 4;
  const long units = (long)geo.n_pad * cpr;
  const int blocks = (int)((units + 255) / 256);
  const int b_blocks = (geo.n_pad + 255) / 256;
  if (compute == TANTE_BF16)
    hipLaunchKernelGGL(pack_kernel<true>, dim3(blocks + b_blocks), dim3(256), 0, s, w, gamma, layout, N, K, P, C_other, geo.n_pad, cpr,
                       geo.nt, (u32x4*)w_out, blocks, bias, beta, bias_out);
  else
    hipLaunchKernelGGL(pack_kernel<false>, dim3(blocks + b_blocks), dim3(256), 0, s, w, gamma, layout, N, K, P, C_other, geo.n_pad, cpr,
                       geo.nt, (u32x4*)w_out, blocks, bias, beta, bias_out);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_gemm(const TanteGemm* gp, void* stream) {
  if (!gp) TANTE_FAIL(-1, "tante_gemm: null descriptor");
  const TanteGemm& g = *gp;
  if (!g.a || !g.w || !g.bias || !g.out) TANTE_FAIL(-1, "tante_gemm: null pointer");
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) TANTE_FAIL(-1, "tante_gemm: bad shape M=%d N=%d K=%d", g.M, g.N, g.K);
  if (g.compute != TANTE_F32 && g.compute != TANTE_BF16) TANTE_FAIL(-1, "tante_gemm: bad compute dtype");
  TantePackGeom geo;
  int rc = tante_pack_geom(g.N, g.K, g.compute, &geo);
  if (rc) return rc;
  const int E = (g.compute == TANTE_BF16) ? 8 : 4;
  const int al = (g.a_dtype == TANTE_BF16) ? E : 4;  // elements per aligned vector load
  int flags = 0;
  const bool a_ptr_ok = ((uintptr_t)g.a % 16) == 0;
  switch (g.a_mode) {
    case TANTE_A_LINEAR:
      if (g.a_n0 <= 0) TANTE_FAIL(-1, "tante_gemm: a_n0 must be > 0");
      if (a_ptr_ok && g.a_s1 % al == 0 && g.a_s0 % al == 0 && g.a_off % al == 0) flags |= 1;
      break;
    case TANTE_A_PATCH_NHWC:
    case TANTE_A_PATCH_NCHW:
      if (g.P <= 0 || g.Hin % g.P || g.Win % g.P || g.Cin <= 0) TANTE_FAIL(-1, "tante_gemm: bad patch geometry");
      if (g.K != g.Cin * g.P * g.P) TANTE_FAIL(-1, "tante_gemm: K != Cin*P*P");
      if (g.M % ((g.Hin / g.P) * (g.Win / g.P))) TANTE_FAIL(-1, "tante_gemm: M is not a whole number of images");
      if (g.a_n0 <= 0) TANTE_FAIL(-1, "tante_gemm: a_n0 (images per batch item) must be > 0");
      if (g.a_s1 % 4 || g.a_off % 4) TANTE_FAIL(-1, "tante_gemm: image strides must be multiples of 4 elements");
      if (g.a_mode == TANTE_A_PATCH_NHWC && a_ptr_ok && (g.P * g.Cin) % E == 0 && g.Cin % al == 0) flags |= 1;
      break;
    default:
      TANTE_FAIL(-1, "tante_gemm: bad a_mode %d", g.a_mode);
  }
  const int oal = 4;
  const bool o_ptr_ok = ((uintptr_t)g.out % 16) == 0;
  switch (g.e_mode) {
    case TANTE_E_LINEAR:
      if (o_ptr_ok && g.N % 4 == 0 && g.out_ld % oal == 0 &&
          (!g.residual || (g.res_ld % 4 == 0 && ((uintptr_t)g.residual % 16) == 0)))
        flags |= 2;
      break;
    case TANTE_E_FILM:
      if (!g.film_a || !g.film_b || !g.s_emb || g.T <= 0 || g.HW <= 0) TANTE_FAIL(-1, "tante_gemm: FILM epilogue needs tables");
      if (o_ptr_ok && g.N % 4 == 0 && g.out_ld % oal == 0) flags |= 2;
      break;
    case TANTE_E_DECONV_NHWC:
    case TANTE_E_DECONV_NCHW:
      if (g.Hi <= 0 || g.Wi <= 0 || g.Po <= 0 || g.Cout <= 0) TANTE_FAIL(-1, "tante_gemm: bad deconv geometry");
      if (g.N != g.Cout * g.Po * g.Po) TANTE_FAIL(-1, "tante_gemm: N != Cout*P*P");
      if (g.M % (g.Hi * g.Wi)) TANTE_FAIL(-1, "tante_gemm: M is not a whole number of images");
      if (g.e_mode == TANTE_E_DECONV_NCHW && g.out_dtype != TANTE_F32) TANTE_FAIL(-1, "tante_gemm: NCHW output is fp32");
      if (g.e_mode == TANTE_E_DECONV_NHWC && o_ptr_ok && g.Cout % 4 == 0) flags |= 2;
      break;
    default:
      TANTE_FAIL(-1, "tante_gemm: bad e_mode %d", g.e_mode);
  }
  if (((uintptr_t)g.w % 16) || ((uintptr_t)g.bias % 16)) TANTE_FAIL(-1, "tante_gemm: packed weight/bias must be 16-byte aligned");
  if (g.drop_p < 0.0f || g.drop_p >= 1.0f) TANTE_FAIL(-1, "tante_gemm: dropout probability must be in [0, 1)");
  const bool train_epi = g.drop_p > 0.0f || g.dact != nullptr;
  const bool dact_scatter = g.dact != nullptr && g.drop_p <= 0.0f && g.e_mode == TANTE_E_DECONV_NHWC && (flags & 2) && g.act == TANTE_ACT_NONE &&
                            ((uintptr_t)g.dact % 16) == 0;      // act'(pre) folded into the channels-last pixel-shuffle store
  if (train_epi && !dact_scatter && (g.compute != TANTE_BF16 || g.e_mode != TANTE_E_LINEAR || g.a_mode != TANTE_A_LINEAR || g.a_dtype != TANTE_BF16 || g.ln ||
                    g.act != TANTE_ACT_NONE || (flags & 3) != 3 || g.M < 4096 || (g.K != 128 && g.K != 256 && g.K != 512) || (g.drop_p > 0.0f && g.dact) ||
                    (g.dact && (((uintptr_t)g.dact % 16) || g.N % 4))))
    TANTE_FAIL(-2, "tante_gemm: the dropout / activation-gradient epilogues need dense 16-byte aligned bf16 rows, M >= 4096, K of 128, 256 or 512, no LayerNorm, no activation");
  if (g.a_pad < 0 || (g.a_pad != 0 && !patch_lite_ok(g, flags, geo.cb)))
    TANTE_FAIL(-2, "tante_gemm: a_pad needs a channels-first bf16-compute patch stage with P = 4, a_pad = 1, K of 256 or 512, M >= 4096, a linear epilogue");
  const int n_tiles = geo.n_pad / geo.nt;
  hipStream_t s = (hipStream_t)stream;
  const bool bf = g.compute == TANTE_BF16;
#define TANTE_DISPATCH(CBV)                                          \
  case CBV:                                                          \
    if (bf) launch_gemm<true, CBV>(g, n_tiles, flags, s);            \
    else launch_gemm<false, CBV>(g, n_tiles, flags, s);              \
    break;
  switch (geo.cb) {
    TANTE_DISPATCH(2)
    TANTE_DISPATCH(4)
    TANTE_DISPATCH(8)
    TANTE_DISPATCH(16)
    case 32:
      if (bf) TANTE_FAIL(-2, "tante_gemm: K too large for bf16 path");
      launch_gemm<false, 32>(g, n_tiles, flags, s);
      break;
    default:
      TANTE_FAIL(-2, "tante_gemm: unsupported K");
  }
#undef TANTE_DISPATCH
  TANTE_CHECK_LAUNCH();
  return 0;
}
