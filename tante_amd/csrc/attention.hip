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
#include "common.hip.h"
#include "fused_common.hip.h"

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

// ---- small-sequence attention on the matrix cores: bf16, head dim 32, L <= 64 (the train path's forward) ---------------------------
// Same unit / tile scheme as attn_bwd_mfma_kernel (backward.hip): one wave per (unit, head), a unit being one sequence in NT 16-slot
// tiles or the 16 / L sequences of one tile behind a block-diagonal mask.  S^T = K Q^T puts a query in each lane's column, so the row
// statistics are two cross-group shuffles; the (dropped) probabilities of two key tiles pack straight into the B operand of
// O^T = V^T P^T, whose A operand is two transposing LDS reads of the row-major V image.
__device__ __forceinline__ u32x2 afm_tr(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
template <int NT>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ o, int C, int n_head,
                                                            TanteSeq sq, int SPT, int causal, float scale, float p_drop, unsigned long long seed) {
  constexpr int ROWS = NT * 16, NP = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char sm_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15, qq = l15 >> 2, pp = l15 & 3;
  const int h = blockIdx.y * 4 + wave;
  if (h >= n_head) return;                       // no barriers: each wave owns its slice of LDS
  constexpr int RS = 48;   // 96-byte LDS rows (64 of data): keeps the transposing reads of the four 16-lane groups off each other's banks
  unsigned short* Vs = (unsigned short*)(sm_raw + wave * ROWS * RS * 2);
  const int L = sq.L, unit = blockIdx.x;
  const bool big = L >= 16;
  const unsigned rcpL = (65536u + L - 1) / L;    // floor(slot / L) for slot < 16
  auto decode = [&](int slot, int& sl, int& pos, bool& live) {
    sl = big ? 0 : (int)(((unsigned)slot * rcpL) >> 16);
    pos = slot - sl * L;
    const int seq = big ? unit : unit * SPT + sl;
    live = (big ? pos < L : sl < SPT) && seq < sq.nseq;
  };
  long ctok[NT];
  int csl[NT], cpos[NT];
  bool clive[NT];
  u32x4 qf[NT], kf[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    decode(t * 16 + l15, csl[t], cpos[t], clive[t]);
    const int seq = big ? unit : unit * SPT + csl[t];
    ctok[t] = clive[t] ? (long)(seq / sq.n_s0) * sq.S1 + (long)(seq % sq.n_s0) * sq.S0 + (long)(cpos[t] / sq.n_l0) * sq.P1 +
                             (long)(cpos[t] % sq.n_l0) * sq.P0 : 0;
    u32x4 vf = u32x4{0u, 0u, 0u, 0u};
    qf[t] = kf[t] = vf;
    if (clive[t]) {
      const unsigned short* r = qkv + ctok[t] * 3L * C + h * 32 + kk * 8;
      qf[t] = *(const u32x4*)r;
      kf[t] = *(const u32x4*)(r + C);
      vf = *(const u32x4*)(r + 2 * C);
    }
    *(u32x4*)(Vs + (t * 16 + l15) * RS + kk * 8) = vf;   // row-major image (64-byte rows) for the transposing reads
  }
  const unsigned troff = lds_addr((const char*)Vs) + (4 * kk + qq) * (RS * 2) + pp * 8;
  const float c2 = scale * 1.4426950408889634f;
  const float ksc = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const bool ilive = clive[it];
    const int ipos = cpos[it], isl = csl[it];
    const unsigned long long mrow = (((unsigned long long)(big ? unit : unit * SPT + isl) * n_head + h) * L + ipos) * L;
    f32x4 st[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) st[jt] = mfma_bf16(kf[jt], qf[it], zero4);
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
    if (p_drop > 0.f) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        if ((L & 3) == 0) {      // the lane's four keys are one aligned run of mask indices of one sequence
          int jsl, jpos; bool jlive;
          decode(jt * 16 + 4 * kk, jsl, jpos, jlive);
          const unsigned m4 = dropout_keep4(seed, mrow + jpos, p_drop);
#pragma unroll
          for (int r = 0; r < 4; ++r) st[jt][r] = ((m4 >> r) & 1u) ? st[jt][r] * ksc : 0.f;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int jsl, jpos; bool jlive;
            decode(jt * 16 + 4 * kk + r, jsl, jpos, jlive);
            st[jt][r] = dropout_keep(seed, mrow + jpos, p_drop) ? st[jt][r] * ksc : 0.f;
          }
        }
      }
    }
    f32x4 oa[2] = {zero4, zero4};
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      const int j0 = 2 * jp, j1 = (2 * jp + 1 < NT) ? 2 * jp + 1 : 2 * jp;
      const u32x4 pf = pack8(st[j0], (2 * jp + 1 < NT) ? st[j1] : zero4);   // un-normalised: 1 / l scales the 8 outputs instead
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x2 lo = afm_tr(troff + dt * 32 + j0 * (16 * RS * 2)), hi = afm_tr(troff + dt * 32 + j1 * (16 * RS * 2));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory");
        oa[dt] = mfma_bf16(u32x4{lo[0], lo[1], hi[0], hi[1]}, pf, oa[dt]);
      }
    }
    if (ilive) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x2 u;
        u[0] = pack_bf16x2(oa[dt][0] * inv_l, oa[dt][1] * inv_l);
        u[1] = pack_bf16x2(oa[dt][2] * inv_l, oa[dt][3] * inv_l);
        *(u32x2*)(o + ctok[it] * (long)C + h * 32 + dt * 16 + 4 * kk) = u;
      }
    }
  }
}

template <int NT>
void launch_attn_fwd_mfma(const void* qkv, void* o, int C, int n_head, const TanteSeq& sq, int SPT, int units, int causal, float p_drop,
                          unsigned long long seed, hipStream_t s) {
  hipLaunchKernelGGL((attn_fwd_mfma_kernel<NT>), dim3(units, (n_head + 3) / 4), dim3(256), 4 * (size_t)NT * 16 * 96, s, (const unsigned short*)qkv,
                     (unsigned short*)o, C, n_head, sq, SPT, causal, 1.0f / sqrtf(32.0f), p_drop, seed);
}
bool try_attn_fwd_mfma(const void* qkv, void* o, int dtype, int C, int n_head, const TanteSeq& sq, int causal, float p_drop, unsigned long long seed,
                       hipStream_t s) {
  const bool off = tante_opt("TANTE_ATTN_FWD_VALU", 0) != 0;
  if (off || dtype != TANTE_BF16 || C != n_head * 32 || sq.L > 64 || sq.L < 1) return false;
  const int L = sq.L;
  const int SPT = L >= 16 ? 1 : 16 / L;
  const int units = L >= 16 ? sq.nseq : (sq.nseq + SPT - 1) / SPT;
  switch (L >= 16 ? (L + 15) / 16 : 1) {
    case 1: launch_attn_fwd_mfma<1>(qkv, o, C, n_head, sq, SPT, units, causal, p_drop, seed, s); break;
    case 2: launch_attn_fwd_mfma<2>(qkv, o, C, n_head, sq, SPT, units, causal, p_drop, seed, s); break;
    case 3: launch_attn_fwd_mfma<3>(qkv, o, C, n_head, sq, SPT, units, causal, p_drop, seed, s); break;
    default: launch_attn_fwd_mfma<4>(qkv, o, C, n_head, sq, SPT, units, causal, p_drop, seed, s); break;
  }
  return true;
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
  if (try_attn_fwd_mfma(qkv, o, dtype, C, n_head, *seq, causal, p_drop, seed, s)) {
    TANTE_CHECK_LAUNCH();
    return 0;
  }
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

// ---- masked attention over dense (batch, L) sequences: nn.MultiheadAttention's attn_mask / key_padding_mask --------------------------
// TransformerBlock.forward(x, key_padding_mask, attn_mask, causal) is the reference's documented operator signature
// (attn_backbone.py:59-72); the TANTE path itself only ever passes `causal`, so this is the boundary's completeness, not a hot kernel:
// one lane per (batch, head, query) walks the keys from global memory with an online softmax; the masks arrive as ADDITIVE fp32 tensors
// (the host turns bool masks into 0 / -inf): attn_mask (L, L) shared or (B n_head, L, L) by mask_bstride, key_padding_mask (B, L).
// A key whose score is -inf contributes nothing; a row with every key blocked yields NaN, as torch's softmax does.
namespace {
template <int D>
__global__ __launch_bounds__(256) void attn_masked_kernel(const void* __restrict__ qkv, void* __restrict__ o, int dtype, int C, int n_head, int Bp,
                                                          int L, int causal, float scale, const float* __restrict__ amask, long mask_bstride,
                                                          const float* __restrict__ kpm) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (b, h, l)
  if (idx >= (long)Bp * n_head * L) return;
  const int l = (int)(idx % L), h = (int)((idx / L) % n_head), b = (int)(idx / ((long)L * n_head));
  float q[D], acc[D];
  load_row<D>(qkv, dtype, ((long)b * L + l) * 3 * C + h * D, q);
#pragma unroll
  for (int i = 0; i < D; ++i) { q[i] *= scale; acc[i] = 0.f; }
  float m = -INFINITY, ssum = 0.f;
  const float* am = amask ? amask + ((long)b * n_head + h) * mask_bstride + (long)l * L : nullptr;
  const float* kp = kpm ? kpm + (long)b * L : nullptr;
  const int jmax = causal ? l + 1 : L;
  for (int j = 0; j < jmax; ++j) {
    float k[D], v[D];
    load_row<D>(qkv, dtype, ((long)b * L + j) * 3 * C + C + h * D, k);
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) sc += q[i] * k[i];
    if (am) sc += am[j];
    if (kp) sc += kp[j];
    if (sc == -INFINITY) continue;
    load_row<D>(qkv, dtype, ((long)b * L + j) * 3 * C + 2 * C + h * D, v);
    const float mn = fmaxf(m, sc);
    const float corr = expf(m - mn), p = expf(sc - mn);      // m = -inf before the first live key: corr = 0
    ssum = ssum * corr + p;
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = acc[i] * corr + p * v[i];
    m = mn;
  }
  const float inv = 1.0f / ssum;      // ssum = 0 (every key blocked): inf * 0 = NaN, torch's result for such a row
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] *= inv;
  store_row<D>(o, dtype, ((long)b * L + l) * C + h * D, acc);
}
}  // namespace

extern "C" int tante_attention_masked(const void* qkv, void* o, int dtype, int C, int n_head, int Bp, int L, int causal, const float* attn_mask,
                                      int64_t mask_bstride, const float* key_padding_mask, void* stream) {
  if (!qkv || !o) TANTE_FAIL(-1, "tante_attention_masked: null pointer");
  if (dtype != TANTE_F32 && dtype != TANTE_BF16) TANTE_FAIL(-2, "tante_attention_masked: dtype");
  if (C <= 0 || n_head <= 0 || C % n_head || Bp <= 0 || L <= 0) TANTE_FAIL(-1, "tante_attention_masked: bad shape");
  if (attn_mask && mask_bstride != 0 && mask_bstride != (int64_t)L * L) TANTE_FAIL(-1, "tante_attention_masked: mask stride must be 0 (shared) or L * L");
  const int D = C / n_head;
  const float scale = 1.0f / sqrtf((float)D);
  const long n = (long)Bp * n_head * L;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define TANTE_AM(DD) hipLaunchKernelGGL(attn_masked_kernel<DD>, grid, block, 0, s, qkv, o, dtype, C, n_head, Bp, L, causal, scale, attn_mask, (long)mask_bstride, key_padding_mask)
  switch (D) {
    case 4: TANTE_AM(4); break;
    case 8: TANTE_AM(8); break;
    case 16: TANTE_AM(16); break;
    case 32: TANTE_AM(32); break;
    case 64: TANTE_AM(64); break;
    default: TANTE_FAIL(-2, "tante_attention_masked: head dim %d (4, 8, 16, 32, 64)", D);
  }
#undef TANTE_AM
  TANTE_CHECK_LAUNCH();
  return 0;
}

// The masked attention's BACKWARD, and the train path's fallback for sequences the MFMA backward does not take (L > 128: the channel
// letter 'C' over 256 channels).  Two lane-per-row passes over the same additive masks, recomputing the probabilities from q, k:
//   pass Q (one lane per (b, h, query)):  m, 1/sum and delta = dO . O by the forward's online recurrence, then
//                                          dq = scale * sum_j p_j (dO . v_j - delta) k_j;  (m, 1/sum, delta) go to `stats`
//   pass K (one lane per (b, h, key)):    dv = sum_i p_ij dO_i,  dk = scale * sum_i p_ij (dO_i . v_j - delta_i) q_i
// Plain sums in a fixed order: deterministic, no atomics.  A completeness kernel like the forward (VALU, global-memory operands).
namespace {
template <int D>
__global__ __launch_bounds__(256) void attn_masked_bwd_q_kernel(const void* __restrict__ qkv, const void* __restrict__ dO, void* __restrict__ dqkv,
                                                                int dtype, int C, int n_head, int Bp, int L, int causal, float scale,
                                                                const float* __restrict__ amask, long mask_bstride,
                                                                const float* __restrict__ kpm, float* __restrict__ stats) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (b, h, l)
  if (idx >= (long)Bp * n_head * L) return;
  const int l = (int)(idx % L), h = (int)((idx / L) % n_head), b = (int)(idx / ((long)L * n_head));
  float q[D], g[D], acc[D];
  load_row<D>(qkv, dtype, ((long)b * L + l) * 3 * C + h * D, q);
  load_row<D>(dO, dtype, ((long)b * L + l) * C + h * D, g);
#pragma unroll
  for (int i = 0; i < D; ++i) { q[i] *= scale; acc[i] = 0.f; }
  float m = -INFINITY, ssum = 0.f;
  const float* am = amask ? amask + ((long)b * n_head + h) * mask_bstride + (long)l * L : nullptr;
  const float* kp = kpm ? kpm + (long)b * L : nullptr;
  const int jmax = causal ? l + 1 : L;
  for (int j = 0; j < jmax; ++j) {
    float k[D], v[D];
    load_row<D>(qkv, dtype, ((long)b * L + j) * 3 * C + C + h * D, k);
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) sc += q[i] * k[i];
    if (am) sc += am[j];
    if (kp) sc += kp[j];
    if (sc == -INFINITY) continue;
    load_row<D>(qkv, dtype, ((long)b * L + j) * 3 * C + 2 * C + h * D, v);
    const float mn = fmaxf(m, sc);
    const float corr = expf(m - mn), p = expf(sc - mn);
    ssum = ssum * corr + p;
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = acc[i] * corr + p * v[i];
    m = mn;
  }
  const float inv = 1.0f / ssum;
  float delta = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) delta += g[i] * acc[i];
  delta *= inv;
  stats[idx * 3 + 0] = m;
  stats[idx * 3 + 1] = inv;
  stats[idx * 3 + 2] = delta;
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = 0.f;      // now dq
  for (int j = 0; j < jmax; ++j) {
    float k[D], v[D];
    load_row<D>(qkv, dtype, ((long)b * L + j) * 3 * C + C + h * D, k);
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) sc += q[i] * k[i];
    if (am) sc += am[j];
    if (kp) sc += kp[j];
    if (sc == -INFINITY) continue;
    load_row<D>(qkv, dtype, ((long)b * L + j) * 3 * C + 2 * C + h * D, v);
    float dp = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) dp += g[i] * v[i];
    const float ds = expf(sc - m) * inv * (dp - delta) * scale;
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] += ds * k[i];
  }
  store_row<D>(dqkv, dtype, ((long)b * L + l) * 3 * C + h * D, acc);
}

template <int D>
__global__ __launch_bounds__(256) void attn_masked_bwd_kv_kernel(const void* __restrict__ qkv, const void* __restrict__ dO, void* __restrict__ dqkv,
                                                                 int dtype, int C, int n_head, int Bp, int L, int causal, float scale,
                                                                 const float* __restrict__ amask, long mask_bstride,
                                                                 const float* __restrict__ kpm, const float* __restrict__ stats) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (b, h, key j)
  if (idx >= (long)Bp * n_head * L) return;
  const int j = (int)(idx % L), h = (int)((idx / L) % n_head), b = (int)(idx / ((long)L * n_head));
  float k[D], v[D], dk[D], dv[D];
  load_row<D>(qkv, dtype, ((long)b * L + j) * 3 * C + C + h * D, k);
  load_row<D>(qkv, dtype, ((long)b * L + j) * 3 * C + 2 * C + h * D, v);
#pragma unroll
  for (int i = 0; i < D; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
  const float kpj = kpm ? kpm[(long)b * L + j] : 0.f;
  const float* am = amask ? amask + ((long)b * n_head + h) * mask_bstride + j : nullptr;
  const float* st = stats + ((long)b * n_head + h) * L * 3;
  if (kpj != -INFINITY) {
    for (int l = causal ? j : 0; l < L; ++l) {
      float q[D], g[D];
      load_row<D>(qkv, dtype, ((long)b * L + l) * 3 * C + h * D, q);
      float sc = 0.f;
#pragma unroll
      for (int i = 0; i < D; ++i) { q[i] *= scale; sc += q[i] * k[i]; }
      if (am) sc += am[(long)l * L];
      sc += kpj;
      if (sc == -INFINITY) continue;
      load_row<D>(dO, dtype, ((long)b * L + l) * C + h * D, g);
      const float p = expf(sc - st[l * 3 + 0]) * st[l * 3 + 1];
      float dp = 0.f;
#pragma unroll
      for (int i = 0; i < D; ++i) { dv[i] += p * g[i]; dp += g[i] * v[i]; }
      const float ds = p * (dp - st[l * 3 + 2]);
#pragma unroll
      for (int i = 0; i < D; ++i) dk[i] += ds * q[i];
    }
  }
  store_row<D>(dqkv, dtype, ((long)b * L + j) * 3 * C + C + h * D, dk);
  store_row<D>(dqkv, dtype, ((long)b * L + j) * 3 * C + 2 * C + h * D, dv);
}
}  // namespace

extern "C" int tante_attention_masked_bwd(const void* qkv, const void* dO, void* dqkv, int dtype, int C, int n_head, int Bp, int L, int causal,
                                          const float* attn_mask, int64_t mask_bstride, const float* key_padding_mask, float* stats,
                                          void* stream) {
  if (!qkv || !dO || !dqkv || !stats) TANTE_FAIL(-1, "tante_attention_masked_bwd: null pointer");
  if (dtype != TANTE_F32 && dtype != TANTE_BF16) TANTE_FAIL(-2, "tante_attention_masked_bwd: dtype");
  if (C <= 0 || n_head <= 0 || C % n_head || Bp <= 0 || L <= 0) TANTE_FAIL(-1, "tante_attention_masked_bwd: bad shape");
  if (attn_mask && mask_bstride != 0 && mask_bstride != (int64_t)L * L)
    TANTE_FAIL(-1, "tante_attention_masked_bwd: mask stride must be 0 (shared) or L * L");
  const int D = C / n_head;
  const float scale = 1.0f / sqrtf((float)D);
  const long n = (long)Bp * n_head * L;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define TANTE_AMB(DD)                                                                                                                          \
  do {                                                                                                                                          \
    hipLaunchKernelGGL(attn_masked_bwd_q_kernel<DD>, grid, block, 0, s, qkv, dO, dqkv, dtype, C, n_head, Bp, L, causal, scale, attn_mask,       \
                       (long)mask_bstride, key_padding_mask, stats);                                                                          \
    hipLaunchKernelGGL(attn_masked_bwd_kv_kernel<DD>, grid, block, 0, s, qkv, dO, dqkv, dtype, C, n_head, Bp, L, causal, scale, attn_mask,      \
                       (long)mask_bstride, key_padding_mask, (const float*)stats);                                                            \
  } while (0)
  switch (D) {
    case 4: TANTE_AMB(4); break;
    case 8: TANTE_AMB(8); break;
    case 16: TANTE_AMB(16); break;
    case 32: TANTE_AMB(32); break;
    case 64: TANTE_AMB(64); break;
    default: TANTE_FAIL(-2, "tante_attention_masked_bwd: head dim %d (4, 8, 16, 32, 64)", D);
  }
#undef TANTE_AMB
  TANTE_CHECK_LAUNCH();
  return 0;
}
