// Per-(sequence, head) scaled-dot-product attention over a strided regrouping of the token grid.
//
// Only this kernel ever sees the axis letter: LayerNorm, projections and the MLP are per-token, so
// the reference's rearrange copies (attn_backbone.py:150-182) are replaced by index arithmetic here
// (TanteSeq: token(s, l) = base(s) + off(l)).
//
// v1 (this file): one lane per query row, K/V of the workgroup's sequences staged in LDS as fp32,
// online softmax in registers.  Two shapes of the same kernel:
//   SMALL (L <= 256): a workgroup owns floor(256 / L) whole sequences of one head; a lane's keys are
//                     the rows of its own sequence (lanes of one sequence read LDS in broadcast).
//   LONG  (L  > 256): a workgroup owns 256 consecutive queries of one (sequence, head) and walks the
//                     keys in chunks of KC through LDS (flash-style), causal chunks skipped.
#include "common.cuh"

namespace {

__device__ __forceinline__ long seq_token(const TanteSeq& q, int s, int l) {
  return (long)(s / q.n_s0) * q.S1 + (long)(s % q.n_s0) * q.S0 + (long)(l / q.n_l0) * q.P1 + (long)(l % q.n_l0) * q.P0;
}

template <int D>
__device__ __forceinline__ void load_row(const void* base, int dtype, long elem, float (&v)[D]) {
  if (dtype == TANTE_BF16) {
    const unsigned short* p = (const unsigned short*)base + elem;
    if constexpr (D == 4) {
      const u32x2 u = *(const u32x2*)p;
      v[0] = bf16_lo(u[0]); v[1] = bf16_hi(u[0]); v[2] = bf16_lo(u[1]); v[3] = bf16_hi(u[1]);
    }
#pragma unroll
    for (int i = 0; i < D / 8; ++i) {
      const u32x4 u = *(const u32x4*)(p + 8 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[8 * i + 2 * j] = bf16_lo(u[j]);
        v[8 * i + 2 * j + 1] = bf16_hi(u[j]);
      }
    }
  } else {
    const float* p = (const float*)base + elem;
#pragma unroll
    for (int i = 0; i < D / 4; ++i) {
      const f32x4 f = *(const f32x4*)(p + 4 * i);
      v[4 * i] = f[0]; v[4 * i + 1] = f[1]; v[4 * i + 2] = f[2]; v[4 * i + 3] = f[3];
    }
  }
}

template <int D>
__device__ __forceinline__ void store_row(void* base, int dtype, long elem, const float (&v)[D]) {
  if (dtype == TANTE_BF16) {
    unsigned short* p = (unsigned short*)base + elem;
    if constexpr (D == 4) {
      u32x2 u;
      u[0] = pack_bf16x2(v[0], v[1]);
      u[1] = pack_bf16x2(v[2], v[3]);
      *(u32x2*)p = u;
    }
#pragma unroll
    for (int i = 0; i < D / 8; ++i) {
      u32x4 u;
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = pack_bf16x2(v[8 * i + 2 * j], v[8 * i + 2 * j + 1]);
      *(u32x4*)(p + 8 * i) = u;
    }
  } else {
    float* p = (float*)base + elem;
#pragma unroll
    for (int i = 0; i < D / 4; ++i) *(f32x4*)(p + 4 * i) = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
  }
}

// LDS row: D floats + 4 pad floats so that rows of different sequences land on different 16-byte slots
template <int D>
struct Lds {
  static constexpr int STRIDE = D + 4;
};

template <int D>
__device__ __forceinline__ void online_step(const float* krow, const float* vrow, const float (&q)[D], float& m, float& ssum,
                                            float (&o)[D], float keep_scale = 1.0f) {
  float sc = 0.0f;
#pragma unroll
  for (int i = 0; i < D / 4; ++i) {
    const f32x4 k4 = *(const f32x4*)(krow + 4 * i);
    sc += q[4 * i] * k4[0] + q[4 * i + 1] * k4[1] + q[4 * i + 2] * k4[2] + q[4 * i + 3] * k4[3];
  }
  const float mn = fmaxf(m, sc);
  const float corr = expf(m - mn);   // m = -inf on the first key -> 0
  const float p = expf(sc - mn);
  ssum = ssum * corr + p;                 // the normaliser sees the un-dropped probabilities (dropout acts on softmax's output)
  const float pd = p * keep_scale;        // keep_scale = 0 (dropped) or 1 / (1 - p_drop)
#pragma unroll
  for (int i = 0; i < D / 4; ++i) {
    const f32x4 v4 = *(const f32x4*)(vrow + 4 * i);
    o[4 * i] = o[4 * i] * corr + pd * v4[0];
    o[4 * i + 1] = o[4 * i + 1] * corr + pd * v4[1];
    o[4 * i + 2] = o[4 * i + 2] * corr + pd * v4[2];
    o[4 * i + 3] = o[4 * i + 3] * corr + pd * v4[3];
  }
  m = mn;
}

// SMALL: grid = (ceil(nseq / G), n_head); G = 256 / L sequences per workgroup
template <int D>
__global__ __launch_bounds__(256) void attn_small_kernel(const void* __restrict__ qkv, void* __restrict__ o, int dtype, int C,
                                                         TanteSeq sq, int G, int causal, float scale, float p_drop,
                                                         unsigned long long seed) {
  constexpr int ST = Lds<D>::STRIDE;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // K rows [256][ST] then V rows [256][ST]
  float* Ks = sm;
  float* Vs = sm + 256 * ST;
  const int tid = threadIdx.x, h = blockIdx.y;
  const int L = sq.L;
  const int g = tid / L, l = tid - g * L;
  const int s = blockIdx.x * G + g;
  const bool live = (g < G) && (s < sq.nseq);
  long tok = 0;
  float q[D];
  if (live) {
    tok = seq_token(sq, s, l);
    const long e = tok * 3L * C + (long)h * D;
    float kv[D];
    load_row<D>(qkv, dtype, e + C, kv);
#pragma unroll
    for (int i = 0; i < D / 4; ++i) *(f32x4*)(Ks + tid * ST + 4 * i) = f32x4{kv[4 * i], kv[4 * i + 1], kv[4 * i + 2], kv[4 * i + 3]};
    load_row<D>(qkv, dtype, e + 2L * C, kv);
#pragma unroll
    for (int i = 0; i < D / 4; ++i) *(f32x4*)(Vs + tid * ST + 4 * i) = f32x4{kv[4 * i], kv[4 * i + 1], kv[4 * i + 2], kv[4 * i + 3]};
    load_row<D>(qkv, dtype, e, q);
#pragma unroll
    for (int i = 0; i < D; ++i) q[i] *= scale;
  }
  __syncthreads();
  if (!live) return;
  float m = -INFINITY, ssum = 0.0f, acc[D];
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = 0.0f;
  const int nk = causal ? (l + 1) : L;
  const int r0 = g * L;
  if (p_drop > 0.0f) {
    const float ks = 1.0f / (1.0f - p_drop);
    const unsigned long long base = (((unsigned long long)s * gridDim.y + h) * L + l) * L;   // (sequence, head, query) row of the mask
    for (int j = 0; j < nk; ++j)
      online_step<D>(Ks + (r0 + j) * ST, Vs + (r0 + j) * ST, q, m, ssum, acc, dropout_keep(seed, base + j, p_drop) ? ks : 0.0f);
  } else {
    for (int j = 0; j < nk; ++j) online_step<D>(Ks + (r0 + j) * ST, Vs + (r0 + j) * ST, q, m, ssum, acc);
  }
  const float inv = 1.0f / ssum;
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] *= inv;
  store_row<D>(o, dtype, tok * (long)C + (long)h * D, acc);
}

// LONG: grid = (ceil(L / 256), nseq, n_head); keys stream through LDS in chunks of KC
template <int D>
__global__ __launch_bounds__(256) void attn_long_kernel(const void* __restrict__ qkv, void* __restrict__ o, int dtype, int C,
                                                        TanteSeq sq, int causal, float scale) {
  constexpr int ST = Lds<D>::STRIDE;
  constexpr int KC = 128;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // K [KC][ST], V [KC][ST]
  float* Ks = sm;
  float* Vs = sm + KC * ST;
  const int tid = threadIdx.x, s = blockIdx.y, h = blockIdx.z;
  const int L = sq.L;
  const int l = blockIdx.x * 256 + tid;
  const bool live = l < L;
  long tok = 0;
  float q[D], acc[D];
#pragma unroll
  for (int i = 0; i < D; ++i) { q[i] = 0.0f; acc[i] = 0.0f; }
  if (live) {
    tok = seq_token(sq, s, l);
    load_row<D>(qkv, dtype, tok * 3L * C + (long)h * D, q);
#pragma unroll
    for (int i = 0; i < D; ++i) q[i] *= scale;
  }
  float m = -INFINITY, ssum = 0.0f;
  const int k_end = causal ? min(L, (int)(blockIdx.x + 1) * 256) : L;  // block-uniform
  for (int k0 = 0; k0 < k_end; k0 += KC) {
    __syncthreads();
    if (tid < KC && k0 + tid < L) {  // 128 loader lanes: one key row each (K then V)
      const long e = seq_token(sq, s, k0 + tid) * 3L * C + (long)h * D;
      float kv[D];
      load_row<D>(qkv, dtype, e + C, kv);
#pragma unroll
      for (int i = 0; i < D / 4; ++i) *(f32x4*)(Ks + tid * ST + 4 * i) = f32x4{kv[4 * i], kv[4 * i + 1], kv[4 * i + 2], kv[4 * i + 3]};
      load_row<D>(qkv, dtype, e + 2L * C, kv);
#pragma unroll
      for (int i = 0; i < D / 4; ++i) *(f32x4*)(Vs + tid * ST + 4 * i) = f32x4{kv[4 * i], kv[4 * i + 1], kv[4 * i + 2], kv[4 * i + 3]};
    }
    __syncthreads();
    if (live) {
      int nk = min(KC, L - k0);
      if (causal) nk = min(nk, l + 1 - k0);
      for (int j = 0; j < nk; ++j) online_step<D>(Ks + j * ST, Vs + j * ST, q, m, ssum, acc);
    }
  }
  if (!live) return;
  const float inv = 1.0f / ssum;
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] *= inv;
  store_row<D>(o, dtype, tok * (long)C + (long)h * D, acc);
}

template <int D>
int launch_attn(const void* qkv, void* o, int dtype, int C, int n_head, const TanteSeq& sq, int causal, float p_drop,
                unsigned long long seed, hipStream_t s) {
  constexpr int ST = Lds<D>::STRIDE;
  const float scale = 1.0f / sqrtf((float)D);
  if (sq.L <= 256) {
    const int G = 256 / sq.L;
    const size_t lds = 2 * 256 * ST * sizeof(float);
    if (lds > 64 * 1024)
      hipFuncSetAttribute((const void*)attn_small_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attn_small_kernel<D>, dim3((sq.nseq + G - 1) / G, n_head), dim3(256), lds, s, qkv, o, dtype, C, sq, G,
                       causal, scale, p_drop, seed);
  } else {
    const size_t lds = 2 * 128 * ST * sizeof(float);
    hipLaunchKernelGGL(attn_long_kernel<D>, dim3((sq.L + 255) / 256, sq.nseq, n_head), dim3(256), lds, s, qkv, o, dtype, C, sq,
                       causal, scale);
  }
  return 0;
}

}  // namespace

static int attention_impl(const void* qkv, void* o, int dtype, int C, int n_head, const TanteSeq* seq, int causal, float p_drop,
                          unsigned long long seed, void* stream) {
  if (!qkv || !o || !seq) TANTE_FAIL(-1, "tante_attention: null pointer");
  if (n_head <= 0 || C % n_head) TANTE_FAIL(-1, "tante_attention: C=%d not divisible by n_head=%d", C, n_head);
  if (seq->nseq <= 0 || seq->L <= 0 || seq->n_s0 <= 0 || seq->n_l0 <= 0) TANTE_FAIL(-1, "tante_attention: bad sequence descriptor");
  if (((uintptr_t)qkv % 16) || ((uintptr_t)o % 16)) TANTE_FAIL(-1, "tante_attention: buffers must be 16-byte aligned");
  const int d = C / n_head;
  if (seq->nseq > 65535 && seq->L > 256) TANTE_FAIL(-2, "tante_attention: too many long sequences for one launch");
  if (p_drop > 0.0f && seq->L > 256) TANTE_FAIL(-2, "tante_attention: attention dropout is implemented for sequences up to 256 tokens");
  if (p_drop < 0.0f || p_drop >= 1.0f) TANTE_FAIL(-1, "tante_attention: dropout probability must be in [0, 1)");
  hipStream_t s = (hipStream_t)stream;
  switch (d) {
    case 4: launch_attn<4>(qkv, o, dtype, C, n_head, *seq, causal, p_drop, seed, s); break;
    case 8: launch_attn<8>(qkv, o, dtype, C, n_head, *seq, causal, p_drop, seed, s); break;
    case 16: launch_attn<16>(qkv, o, dtype, C, n_head, *seq, causal, p_drop, seed, s); break;
    case 32: launch_attn<32>(qkv, o, dtype, C, n_head, *seq, causal, p_drop, seed, s); break;
    case 64: launch_attn<64>(qkv, o, dtype, C, n_head, *seq, causal, p_drop, seed, s); break;
    default: TANTE_FAIL(-2, "tante_attention: head dim %d unsupported (4, 8, 16, 32, 64)", d);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_attention(const void* qkv, void* o, int dtype, int C, int n_head, const TanteSeq* seq, int causal,
                               void* stream) {
  return attention_impl(qkv, o, dtype, C, n_head, seq, causal, 0.0f, 0ull, stream);
}

extern "C" int tante_attention_dropout(const void* qkv, void* o, int dtype, int C, int n_head, const TanteSeq* seq, int causal,
                                       float p_drop, uint64_t seed, void* stream) {
  return attention_impl(qkv, o, dtype, C, n_head, seq, causal, p_drop, (unsigned long long)seed, stream);
}
