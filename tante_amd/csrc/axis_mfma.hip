// One axis propagator on the matrix pipe, bf16 compute:  x[o, :, i] += W2 gelu(W1 x[o, :, i] + b1) + b2  along an axis of n = 16 .. 64
// positions (attn_backbone.py:140-143: the vertical / horizontal nn.Sequential(Linear(n, n), GELU, Linear(n, n)) applied through a
// rearrange), for planes too large for the whole-plane H + W kernel (axis_hw_exact_kernel keeps a 32-channel tile of a WHOLE plane in LDS:
// a 64 x 64 plane -- cfg5's 512 x 512 fields at patch scale 8 -- does not fit).  Before, those shapes ran the lane-per-column fp32 kernel
// (axis_mlp_kernel<64>: two 64 x 64 mat-vecs per column on the vector units, 75 us per axis at 2 x 4 x 64 x 64 x 256).
//
// A workgroup owns one line group: all n positions x 256 consecutive inner elements.  The rows arrive as whole contiguous 1 KiB runs
// (16 bytes per lane) into an fp32 LDS tile [n][258] (stride 258: an operand fragment's four 8-row groups sit 16 banks apart); a wave
// takes 16 columns at a time: B fragments = 8 positions of a column from the tile, packed to bf16; pre = W1 x + b1 with W1 as A fragments
// held in registers; GELU (the bf16 mode's polynomial) on the accumulators; the accumulator tiles, packed in pairs, ARE the B operand of
// the second product (W2's fragments are gathered in the matching k order); the result is added to the fp32 tile in place and the tile
// leaves as whole rows.  x is read and written once.
#include "common.hip.h"
#include "fused_common.hip.h"

namespace {

constexpr int AMB = 256;        // columns (inner elements) per workgroup
constexpr int AMP = AMB + 2;    // tile row stride in floats

// FILM (the first propagator of a call whose window comes from the rollout's frame cache, C = AMB = 256 channels: a workgroup's 1 KiB
// run is one token): x is not read but PRODUCED -- row p is token (p, ib) of frame t = o % T of item o / T in the frame-major cache,
// times a[t] plus b[t] plus the positional term, tante_film_pos_fwd_frames_kernel's expression (tante.py:136-141) -- while the tile loads.
struct AxmFilm {
  const float* f[8];       // frame t: (B, HW, C) fp32, item stride bstride[t]
  long bstride[8];
  const float *a, *b, *s;  // (T, C), (T, C), (HW, C)
  int T;
};

template <int MT, bool FILM = false>               // n = 16 MT
__global__ __launch_bounds__(256) void axis_mlp_mfma_kernel(float* __restrict__ x, long inner, const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2, AxmFilm F = AxmFilm{}) {
  constexpr int N = 16 * MT, KS = (MT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float tile[];     // [N][AMP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kk = lane >> 4;
  const long nib = inner / AMB, o = blockIdx.x / nib, ib = blockIdx.x - o * nib;
  float* gx = x + o * N * inner + ib * AMB + 4 * lane;               // row p: gx + p * inner
  if constexpr (FILM) {
    const int t = (int)(o % F.T);
    const long bi = o / F.T;
    const float* src = F.f[t] + bi * F.bstride[t] + ib * AMB + 4 * lane;      // token (p, ib) = row p * nib + ib of the (HW, C) frame
    const f32x4 av = *(const f32x4*)(F.a + t * AMB + 4 * lane), bv = *(const f32x4*)(F.b + t * AMB + 4 * lane);
    for (int p = wave; p < N; p += 4) {
      const f32x4 vv = *(const f32x4*)(src + (long)p * inner);
      const f32x4 sv = *(const f32x4*)(F.s + ((long)p * nib + ib) * AMB + 4 * lane);
      const f32x4 v = vv * av + bv + sv;
      float2* d = (float2*)(tile + p * AMP + 4 * lane);
      d[0] = float2{v[0], v[1]};
      d[1] = float2{v[2], v[3]};
    }
  } else {
    for (int p = wave; p < N; p += 4) {
      const f32x4 v = *(const f32x4*)(gx + (long)p * inner);
      float2* d = (float2*)(tile + p * AMP + 4 * lane);               // (row stride 1 032 bytes: 8-byte aligned)
      d[0] = float2{v[0], v[1]};
      d[1] = float2{v[2], v[3]};
    }
  }
  // weights as A fragments: lane (row l15 of tile mt, kk) holds 8 k values.  W1 in the natural k order (k = axis position); W2 in the
  // order the packed accumulator pairs present the hidden positions: k = 8 kk + e  <->  position 32 s + 16 (e >> 2) + 4 kk + (e & 3)
  u32x4 Wa[MT][KS], Wb[MT][KS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float* r1 = w1 + (long)(16 * mt + l15) * N;
      const float* r2 = w2 + (long)(16 * mt + l15) * N;
      const int k0 = 32 * s + 8 * kk, p0 = 32 * s + 4 * kk;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, c0 = a0, c1 = a0;
      if (k0 < N) { a0 = *(const f32x4*)(r1 + k0); a1 = *(const f32x4*)(r1 + k0 + 4); }
      if (p0 < N) c0 = *(const f32x4*)(r2 + p0);
      if (p0 + 16 < N) c1 = *(const f32x4*)(r2 + p0 + 16);
      Wa[mt][s] = pack8(a0, a1);
      Wb[mt][s] = pack8(c0, c1);
    }
  f32x4 b1v[MT], b2v[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    b1v[mt] = *(const f32x4*)(b1 + 16 * mt + 4 * kk);
    b2v[mt] = *(const f32x4*)(b2 + 16 * mt + 4 * kk);
  }
  __syncthreads();
#pragma unroll 1
  for (int ct = wave; ct < AMB / 16; ct += 4) {
    const float* col = tile + 16 * ct + l15;
    u32x4 Xf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int p = 32 * s + 8 * kk + q;
        v[q] = p < N ? col[p * AMP] : 0.0f;
      }
      Xf[s][0] = pack_bf16x2(v[0], v[1]); Xf[s][1] = pack_bf16x2(v[2], v[3]);
      Xf[s][2] = pack_bf16x2(v[4], v[5]); Xf[s][3] = pack_bf16x2(v[6], v[7]);
    }
    f32x4 acc[2 * KS];                                  // hidden positions 16 mt + 4 kk + r of column l15 (tiles >= MT stay zero)
#pragma unroll
    for (int mt = 0; mt < 2 * KS; ++mt) acc[mt] = mt < MT ? b1v[mt < MT ? mt : 0] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < KS; ++s) acc[mt] = mfma_bf16(Wa[mt][s], Xf[s], acc[mt]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = gelu_poly4<false>(acc[mt]);
    f32x4 out[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) out[mt] = b2v[mt];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 gf = pack8(acc[2 * s], acc[2 * s + 1]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) out[mt] = mfma_bf16(Wb[mt][s], gf, out[mt]);
    }
    float* dcol = tile + 16 * ct + l15;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dcol[(16 * mt + 4 * kk + r) * AMP] += out[mt][r];
  }
  __syncthreads();
  for (int p = wave; p < N; p += 4) {
    const float2* d = (const float2*)(tile + p * AMP + 4 * lane);
    const float2 lo = d[0], hi = d[1];
    *(f32x4*)(gx + (long)p * inner) = f32x4{lo.x, lo.y, hi.x, hi.y};
  }
}

TantePerDevice g_axm_attr[4];

template <int MT>
void axm_launch(float* x, long outer, long inner, const float* w1, const float* b1, const float* w2, const float* b2, hipStream_t s) {
  constexpr int lds = 16 * MT * AMP * 4;
  g_axm_attr[MT - 1].once([] { (void)hipFuncSetAttribute((const void*)axis_mlp_mfma_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
  hipLaunchKernelGGL((axis_mlp_mfma_kernel<MT>), dim3((unsigned)(outer * (inner / AMB))), dim3(256), lds, s, x, inner, w1, b1, w2, b2);
}

template <int MT>
void axm_launch_film(float* x, long outer, long inner, const float* w1, const float* b1, const float* w2, const float* b2, const AxmFilm& F, hipStream_t s) {
  constexpr int lds = 16 * MT * AMP * 4;
  static TantePerDevice attr;
  attr.once([] { (void)hipFuncSetAttribute((const void*)axis_mlp_mfma_kernel<MT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); });
  hipLaunchKernelGGL((axis_mlp_mfma_kernel<MT, true>), dim3((unsigned)(outer * (inner / AMB))), dim3(256), lds, s, x, inner, w1, b1, w2, b2, F);
}

}  // namespace

// The vertical propagator of a call whose window is T cached frame encodings: FiLM + positional terms applied while the tile loads
// (tante_film_pos_fwd_frames + tante_axis_mlp_c in one launch).  x (B T, n, inner) is written, not read; C = 256 (inner = W * 256).
extern "C" int tante_axis_mlp_film_supported(int64_t B, int T, int n, int64_t inner, int C) {
  return C == AMB && T >= 1 && T <= 8 && B > 0 && n >= 16 && n <= 64 && n % 16 == 0 && inner % AMB == 0 && B * T * (inner / AMB) <= 2147483647L &&
         tante_opt("TANTE_AXIS_MFMA", 1) && tante_opt("TANTE_AXIS_FILM", 1);
}

extern "C" int tante_axis_mlp_film(float* x, const TanteFrames* frames, const float* a, const float* b, const float* s_emb, int64_t B, int T, int n,
                                   int64_t inner, int C, const float* w1, const float* b1, const float* w2, const float* b2, void* stream) {
  if (!x || !frames || !a || !b || !s_emb || !w1 || !b1 || !w2 || !b2) TANTE_FAIL(-1, "tante_axis_mlp_film: null pointer");
  if (!tante_axis_mlp_film_supported(B, T, n, inner, C)) TANTE_FAIL(-2, "tante_axis_mlp_film: bf16 matrix-pipe propagator shapes with C = 256 only");
  if (((uintptr_t)x | (uintptr_t)a | (uintptr_t)b | (uintptr_t)s_emb | (uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)b1 | (uintptr_t)b2) % 16)
    TANTE_FAIL(-1, "tante_axis_mlp_film: 16-byte alignment");
  AxmFilm F;
  for (int t = 0; t < 8; ++t) {
    F.f[t] = t < T ? (const float*)frames->f[t] : nullptr;
    F.bstride[t] = t < T ? (long)frames->bstride[t] : 0;
    if (t < T && (!F.f[t] || ((uintptr_t)F.f[t] % 16) || F.bstride[t] % 4)) TANTE_FAIL(-1, "tante_axis_mlp_film: frame %d: null / unaligned", t);
  }
  F.a = a; F.b = b; F.s = s_emb; F.T = T;
  hipStream_t s = (hipStream_t)stream;
  switch (n / 16) {
    case 1: axm_launch_film<1>(x, (long)(B * T), (long)inner, w1, b1, w2, b2, F, s); break;
    case 2: axm_launch_film<2>(x, (long)(B * T), (long)inner, w1, b1, w2, b2, F, s); break;
    case 3: axm_launch_film<3>(x, (long)(B * T), (long)inner, w1, b1, w2, b2, F, s); break;
    default: axm_launch_film<4>(x, (long)(B * T), (long)inner, w1, b1, w2, b2, F, s); break;
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}

// 1: the matrix-pipe form applies (tante_axis_mlp_c, bf16 compute)
int tante_axis_mlp_mfma_supported(int64_t outer, int n, int64_t inner, const void* x, const void* w1, const void* w2) {
  return n >= 16 && n <= 64 && n % 16 == 0 && inner % AMB == 0 && outer > 0 && outer * (inner / AMB) <= 2147483647L && ((uintptr_t)x % 16) == 0 &&
         ((uintptr_t)w1 % 16) == 0 && ((uintptr_t)w2 % 16) == 0;
}

int tante_axis_mlp_mfma(float* x, int64_t outer, int n, int64_t inner, const float* w1, const float* b1, const float* w2, const float* b2, hipStream_t s) {
  switch (n / 16) {
    case 1: axm_launch<1>(x, (long)outer, (long)inner, w1, b1, w2, b2, s); break;
    case 2: axm_launch<2>(x, (long)outer, (long)inner, w1, b1, w2, b2, s); break;
    case 3: axm_launch<3>(x, (long)outer, (long)inner, w1, b1, w2, b2, s); break;
    default: axm_launch<4>(x, (long)outer, (long)inner, w1, b1, w2, b2, s); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
