// Fused TransformerBlock halves for the bf16 path (attn_backbone.py:59-83), one launch each:
//
//   fused_attn_kernel:  x <- x + Wo . Attention(LayerNorm1(x)) + bo      (l.68-81)
//   fused_mlp_kernel:   x <- x + W2 . gelu_tanh(W1 . LayerNorm2(x) + b1) + b2   (l.82)
//
// Unfused, one block moves ~356 MB (B=8, C=256) of activations through HBM/MALL for 21.5 GFLOP; fused it
// reads and writes the fp32 residual stream once per half (~128 MB) and nothing else.
//
// Structure (both kernels): a wave owns 32 tokens for the whole kernel.  Their LayerNorm-ed bf16 MFMA
// fragments live in registers; the weights stream through LDS as pre-swizzled tiles by LDS-DMA
// (global_load_lds_dwordx4, no staging registers), double buffered, one barrier per tile.  Every
// intermediate (q, k, v, scores, probabilities, per-head outputs, MLP hidden) stays in registers: the
// accumulator layout of v_mfma_f32_16x16x32_bf16 -- output column on lane&15, output rows 4*(lane>>4)+r --
// is, after a bf16 pack of two 16-row tiles, exactly a B (or A) operand fragment of the next MFMA with the
// k order inside each 32-block permuted to  p = 8*kk + 4*dt + r  <->  c = 16*dt + 4*kk + r.  A dot product
// does not care about the order as long as both operands agree, so the weights that consume such a
// fragment (Wo, W2) are packed with that permutation and no cross-lane movement is ever needed:
//     q,k   = W x              (A = W tile, B = x)         -> Q/K fragments (token on lane&15)
//     v^T   = x^T W^T          (A = x, B = W tile: roles swapped) -> V^T fragments (feature on lane&15)
//     S^T   = K Q^T            (A = K, B = Q)              -> key j on rows, query i on lane&15
//     P^T   = softmax over rows (4 regs x 2 tiles in-lane, 2 shuffles across lane groups)
//     O^T   = V^T P^T          (A = V^T, B = P^T)          -> per-head output, token on lane&15
//     y     = Wo O             (A = Wo tile [k-permuted], B = O)
// A sequence (L | 32) lives entirely inside one wave, so attention needs no LDS exchange at all; shorter
// sequences are packed 32/L per wave and separated by the mask (block diagonal, causal inside for 'T').
#include "common.hip.h"

#include <stdlib.h>
#include "fused_common.hip.h"
#include "block_sliced.h"

namespace {

constexpr int BIAS_BYTES = 4096;  // per-tile bias block: one full LDS-DMA pass (4 waves x 1 KiB), so every wave issues the same count

__device__ __forceinline__ long seq_token(const TanteSeq& q, int s, int l) {
  return (long)(s / q.n_s0) * q.S1 + (long)(s % q.n_s0) * q.S0 + (long)(l / q.n_l0) * q.P1 + (long)(l % q.n_l0) * q.P0;
}

// LDS-DMA copy of BYTES (multiple of 4096) from global to LDS by the whole 256-thread workgroup: every
// wave-instruction moves 1 KiB to  wave-uniform base + lane * 16; each wave issues exactly BYTES / 4096
// instructions, which is what the counted s_waitcnt vmcnt(N) of the tile pipeline relies on.
template <int BYTES>
__device__ __forceinline__ void glds_copy(const char* __restrict__ g, char* l, int tid) {
  static_assert(BYTES % 4096 == 0, "tiles are whole LDS-DMA passes");
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int i = 0; i < BYTES / 4096; ++i) {
    const int off = i * 4096 + wave * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off + lane * 16),
                                     (__attribute__((address_space(3))) void*)(l + off), 16, 0, 0);
  }
}

__device__ __forceinline__ float gelu_tanh_fast(float x) {
  const float u2 = 1.59576912160573071176f * (x + 0.044715f * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __expf(-u2));   // v_rcp_f32 (1 ulp), not the IEEE division sequence
}

// LayerNorm (no affine) of the wave's 32 residual rows, held in registers in ACCUMULATOR order:
// res[tt][g] = features 16*g + 4*kk .. +3 of the token of slot (tt, lane&15).  The bf16 fragment of k-block b
// is pack8(res[2b], res[2b+1]), i.e. the k-permuted order every weight tile is packed in.
template <int CB>
__device__ __forceinline__ void norm_frags(const f32x4 (&res)[2][2 * CB], const bool (&live)[2], int C, float eps,
                                           u32x4 (&xn)[2][CB]) {
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    float s = 0.0f;
#pragma unroll
    for (int g = 0; g < 2 * CB; ++g) s += res[tt][g][0] + res[tt][g][1] + res[tt][g][2] + res[tt][g][3];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int g = 0; g < 2 * CB; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = res[tt][g][j] - mean;
        q += d * d;
      }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    const float rstd = live[tt] ? rsqrtf(q / (float)C + eps) : 0.0f;
#pragma unroll
    for (int b = 0; b < CB; ++b) xn[tt][b] = pack8((res[tt][2 * b] - mean) * rstd, (res[tt][2 * b + 1] - mean) * rstd);
  }
}

// residual epilogue of a 64-row output tile: x[token][n0..n0+3] += acc + bias  (fp32, in place)
__device__ __forceinline__ void store_residual(float* __restrict__ x, long tok, int n0, const f32x4& acc, const f32x4& bias) {
  if (tok < 0) return;
  float* p = x + tok + n0;
  const f32x4 r = *(const f32x4*)p;
  *(f32x4*)p = r + acc + bias;
}

// ------------------------------------------------------------------------------------------------------
// Whole block in one launch.  Stream: NH head tiles [q_h/sqrt(d) | k_h | v_h] (96 rows x C, LN1 folded),
// C/64 tiles of Wo (64 rows x C), HID/64 tiles of W1 (64 rows x C, LN2 folded), C/64 tiles of W2
// (64 rows x HID); all rows k-permuted; each tile is followed by its 1 KiB bias block.
// The fp32 residual rows are read once, live in registers for the whole kernel and are written once.
// ------------------------------------------------------------------------------------------------------
template <int CB, int HB>
__global__ __launch_bounds__(256, 1) void fused_block_kernel(float* __restrict__ x, const char* __restrict__ stream, TanteSeq sq,
                                                            int causal, float eps) {
  constexpr int C = CB * 32, HID = HB * 32, NH = CB;
  constexpr int CPR = CB * 4, CPRH = HB * 4;
  constexpr int TO = C / 64, T1 = HID / 64;
  constexpr int TILEH = 96 * CPR * 16 + BIAS_BYTES, TILEO = 64 * CPR * 16 + BIAS_BYTES, TILE2 = 64 * CPRH * 16 + BIAS_BYTES;
  constexpr int SLOT = (TILEH > TILE2 ? TILEH : TILE2);
  constexpr int NT = NH + TO + T1 + TO;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;

  // weight tiles stream through a ring of 3 LDS slots, two tiles ahead of the MFMAs
  auto tile_bytes = [](int u) constexpr { return u < NH ? TILEH : (u < NH + TO + T1 ? TILEO : TILE2); };
  auto tile_off = [](int u) constexpr {
    long o = 0;
    for (int i = 0; i < u; ++i) o += (i < NH ? TILEH : (i < NH + TO + T1 ? TILEO : TILE2));
    return o;
  };
  glds_copy<TILEH>(stream, smem, tid);                              // tile 0 -> slot 0
  glds_copy<tile_bytes(1)>(stream + tile_off(1), smem + SLOT, tid);  // tile 1 -> slot 1

  const int L = sq.L, spw = 32 / L;  // sequences per wave
  const int seq0 = (blockIdx.x * 4 + wave) * spw;
  long tok[2];
  bool live[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int slot = tt * 16 + l15, sl = slot / L, pos = slot - sl * L;
    live[tt] = seq0 + sl < sq.nseq;
    tok[tt] = live[tt] ? seq_token(sq, seq0 + sl, pos) * C : 0;
  }
  f32x4 res[2][2 * CB];  // the residual stream of this wave's tokens (fp32), accumulator order
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int g = 0; g < 2 * CB; ++g) res[tt][g] = *(const f32x4*)(x + tok[tt] + 16 * g + 4 * kk);

  // key j may be seen by query i (slots inside this wave): same sequence, causal order, key slot in use
  unsigned allow[2] = {0u, 0u};  // bit (jt*4 + r) for query tile it
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = it * 16 + l15, si = i / L, pi = i - si * L;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = jt * 16 + kk * 4 + r, sj = j / L, pj = j - sj * L;
        const bool ok = (sj == si) && (!causal || pj <= pi) && (seq0 + sj < sq.nseq);
        allow[it] |= (ok ? 1u : 0u) << (jt * 4 + r);
      }
  }

  // byte offset of this lane's operand chunk of k-block kb inside a weight row (XOR swizzle resolved once)
  int xo_c[CB], xo_h[HB];
#pragma unroll
  for (int kb = 0; kb < CB; ++kb) xo_c[kb] = swz_chunk(l15, kb * 4 + kk, CPR) << 4;
#pragma unroll
  for (int kb = 0; kb < HB; ++kb) xo_h[kb] = swz_chunk(l15, kb * 4 + kk, CPRH) << 4;
  const int row_c = l15 * CPR * 16, row_h = l15 * CPRH * 16;

  u32x4 xn[2][CB];   // LayerNorm-ed tokens (LN1, later LN2) as B-operand fragments
  norm_frags<CB>(res, live, C, eps, xn);
  u32x4 of[2][HB > CB ? HB : CB];  // per-head attention outputs, later the MLP hidden activations

  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();

  static_for<NT>([&](auto t_c) {
    constexpr int t = decltype(t_c)::value;
    // slot (t+2)%3 was last read during tile t-1, which every wave left through the barrier below
    if constexpr (t + 2 < NT) glds_copy<tile_bytes(t + 2)>(stream + tile_off(t + 2), smem + ((t + 2) % 3) * SLOT, tid);
    const char* wt = smem + (t % 3) * SLOT;
    if constexpr (t < NH) {
      // =============================== one attention head ==============================================
      const float* bias = (const float*)(wt + 96 * CPR * 16);
      f32x4 aqk[4][2];  // q (ns 0,1) and k (ns 2,3):  D[feature][token]
#pragma unroll
      for (int ns = 0; ns < 4; ++ns)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) aqk[ns][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const u32x4 wf = *(const u32x4*)(wt + ns * 16 * CPR * 16 + row_c + xo_c[cb]);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) aqk[ns][tt] = mfma_bf16(wf, xn[tt][cb], aqk[ns][tt]);
        }
      }
      f32x4 av[2][2];  // v with the operand roles swapped:  D[token][feature]
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) av[tt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const u32x4 wf = *(const u32x4*)(wt + (64 + dt * 16) * CPR * 16 + row_c + xo_c[cb]);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) av[tt][dt] = mfma_bf16(xn[tt][cb], wf, av[tt][dt]);
        }
      }
      u32x4 qf[2], kf[2], vtf[2];
      {
        const f32x4 bq0 = *(const f32x4*)(bias + kk * 4), bq1 = *(const f32x4*)(bias + 16 + kk * 4);
        const f32x4 bk0 = *(const f32x4*)(bias + 32 + kk * 4), bk1 = *(const f32x4*)(bias + 48 + kk * 4);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          qf[tt] = pack8(aqk[0][tt] + bq0, aqk[1][tt] + bq1);
          kf[tt] = pack8(aqk[2][tt] + bk0, aqk[3][tt] + bk1);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const float bv = bias[64 + dt * 16 + l15];
          vtf[dt] = pack8(av[0][dt] + bv, av[1][dt] + bv);
        }
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {  // scores transposed: rows = keys, column (lane&15) = query
        f32x4 sc[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) sc[jt] = mfma_bf16(kf[jt], qf[it], f32x4{0.f, 0.f, 0.f, 0.f});
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((allow[it] >> (jt * 4 + r)) & 1u) m = fmaxf(m, sc[jt][r]);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.0f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = ((allow[it] >> (jt * 4 + r)) & 1u) ? __builtin_amdgcn_exp2f(sc[jt][r] - m) : 0.0f;
            sc[jt][r] = p;
            sum += p;
          }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = sum > 0.0f ? __builtin_amdgcn_rcpf(sum) : 0.0f;
        const u32x4 pf = pack8(sc[0] * inv, sc[1] * inv);
        const f32x4 o0 = mfma_bf16(vtf[0], pf, f32x4{0.f, 0.f, 0.f, 0.f});  // O^T = V^T P^T
        const f32x4 o1 = mfma_bf16(vtf[1], pf, f32x4{0.f, 0.f, 0.f, 0.f});
        of[it][t] = pack8(o0, o1);
      }
    } else {
      // =============================== 64 output features of a dense layer ==============================
      constexpr bool is_out = t < NH + TO, is_fc1 = !is_out && t < NH + TO + T1;
      constexpr int KB = is_fc1 ? CB : (is_out ? CB : HB);   // k-blocks of this layer
      constexpr int cpr = KB * 4;
      const float* bias = (const float*)(wt + 64 * cpr * 16);
      constexpr int tb = is_fc1 ? t - NH - TO : (is_out ? t - NH : t - NH - TO - T1);  // which 64 features
      f32x4 acc[4][2];
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) {
        const f32x4 b = *(const f32x4*)(bias + ns * 16 + kk * 4);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          // residual layers start the MFMA accumulation from (residual + bias): the updated residual is the
          // accumulator itself, so it can stay in the accumulator register file
          if constexpr (is_fc1) acc[ns][tt] = b;
          else acc[ns][tt] = res[tt][4 * tb + ns] + b;
        }
      }
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const u32x4 wf = *(const u32x4*)(wt + ns * 16 * cpr * 16 + ((KB == CB) ? row_c + xo_c[kb < CB ? kb : 0] : row_h + xo_h[kb < HB ? kb : 0]));
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            if constexpr (is_fc1) acc[ns][tt] = mfma_bf16(wf, xn[tt][kb], acc[ns][tt]);
            else acc[ns][tt] = mfma_bf16(wf, of[tt][kb], acc[ns][tt]);
          }
        }
      }
      if constexpr (is_fc1) {
#pragma unroll
        for (int ns = 0; ns < 4; ++ns)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            acc[ns][tt] = gelu_poly4<true>(acc[ns][tt]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          of[tt][2 * tb] = pack8(acc[0][tt], acc[1][tt]);
          of[tt][2 * tb + 1] = pack8(acc[2][tt], acc[3][tt]);
        }
      } else {
#pragma unroll
        for (int ns = 0; ns < 4; ++ns)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) res[tt][4 * tb + ns] = acc[ns][tt];
        if constexpr (is_out && tb == TO - 1) norm_frags<CB>(res, live, C, eps, xn);  // x1 complete: LayerNorm2
      }
    }
    // tile t+1 must have landed (every wave's share) before anyone reads it; tile t+2 stays in flight:
    // it is exactly the tile_bytes(t+2)/4096 youngest vector-memory operations of this wave
    if constexpr (t + 2 < NT) wait_vmcnt<tile_bytes(t + 2) / 4096>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  });
  // the block's output rows, written once (16 B per lane per 16-feature group)
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
    if (live[tt]) {
#pragma unroll
      for (int g = 0; g < 2 * CB; ++g) *(f32x4*)(x + tok[tt] + 16 * g + 4 * kk) = res[tt][g];
    }
}

// ------------------------------------------------------------------------------------------------------
// Same block, 16 tokens per wave, 8 waves per workgroup (two waves per SIMD).
//
// The 32-token-per-wave kernel above needs the whole 512-entry register file per wave, i.e. one wave per SIMD:
// with nothing to switch to, every LDS / LDS-DMA latency and every non-MFMA instruction is exposed (measured:
// MFMA busy 18 % of wave cycles, 44 % parked in waits).  Here a wave owns 16 tokens (residual 64 regs, fragments
// 32 + 32), fits 256 registers, and two waves share each SIMD.  A 32-long sequence then spans a PAIR of waves
// (2p, 2p+1: positions 0-15 / 16-31); per head the pair swaps its K fragments and packed V^T halves through a
// 2 KiB-per-wave LDS mailbox.  The attention of head h-1 runs after the projections of head h, so the mailbox
// written in iteration h-1 is covered by the tile barrier that is there anyway (no extra barrier per head).
// Sequences with L | 16 stay inside one wave (16/L per wave) and use no mailbox.
// ------------------------------------------------------------------------------------------------------
template <int CB>
__device__ __forceinline__ void norm_frags16(const f32x4 (&res)[2 * CB], bool live, int C, float eps, u32x4 (&xn)[CB]) {
  // one pass: sum and sum of squares (E[x^2] - mean^2 in fp32 is ample for a stream that is rounded to bf16 next), then one fma per
  // element -- this kernel is bound by VALU issue, every instruction per element counts
  float s = 0.0f, q = 0.0f;
#pragma unroll
  for (int g = 0; g < 2 * CB; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s += res[g][j];
      q = fmaf(res[g][j], res[g][j], q);
    }
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  q += __shfl_xor(q, 16);
  q += __shfl_xor(q, 32);
  const float mean = s / (float)C;
  const float var = fmaxf(q / (float)C - mean * mean, 0.0f);
  const float rstd = live ? rsqrtf(var + eps) : 0.0f;
  const float sh = -mean * rstd;
#pragma unroll
  for (int b = 0; b < CB; ++b) {
    f32x4 lo, hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) { lo[j] = fmaf(res[2 * b][j], rstd, sh); hi[j] = fmaf(res[2 * b + 1][j], rstd, sh); }
    xn[b] = pack8(lo, hi);
  }
}

// LDS-DMA copy by a 512-thread workgroup (8 waves x 1 KiB per pass)
__device__ __forceinline__ void glds_copy8(const char* __restrict__ g, char* l, int bytes, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  for (int off = wave * 1024; off < bytes; off += 8192) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off + lane * 16),
                                     (__attribute__((address_space(3))) void*)(l + off), 16, 0, 0);
  }
}

template <int CB, int HB>
__global__ __launch_bounds__(512, 2) void fused_block16_kernel(float* __restrict__ x, const char* __restrict__ stream, TanteSeq sq,
                                                              int causal, float eps, int dbg) {
  constexpr int C = CB * 32, HID = HB * 32, NH = CB;
  constexpr int CPR = CB * 4, CPRH = HB * 4;
  constexpr int TO = C / 64, T1 = HID / 64;
  constexpr int TILEH = 96 * CPR * 16 + BIAS_BYTES, TILEO = 64 * CPR * 16 + BIAS_BYTES, TILE2 = 64 * CPRH * 16 + BIAS_BYTES;
  constexpr int SLOT = (TILEH > TILE2 ? TILEH : TILE2);
  constexpr int NT = NH + TO + T1 + TO;
  constexpr int MAXB = (HB > CB ? HB : CB);
#ifndef TANTE_RING
#define TANTE_RING 4
#endif
  constexpr int RING = TANTE_RING;  // weight fragments in flight per wave (mfma_stream); 8 measured no faster
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 weight slots, then 2 mailbox sets of 8 x 2 KiB
  char* mbox = smem + 2 * SLOT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  auto tile_bytes = [](int u) constexpr { return u < NH ? TILEH : (u < NH + TO + T1 ? TILEO : TILE2); };
  auto tile_off = [](int u) constexpr {
    long o = 0;
    for (int i = 0; i < u; ++i) o += (i < NH ? TILEH : (i < NH + TO + T1 ? TILEO : TILE2));
    return o;
  };
  glds_copy8(stream, smem, TILEH, tid);  // tile 0 in flight while the tokens load

  const int L = sq.L;
  const bool paired = (L == 32);
  const int half = wave & 1;  // paired: which 16 positions of the sequence this wave holds
  int seq, pos;
  if (paired) {
    seq = blockIdx.x * 4 + (wave >> 1);
    pos = half * 16 + l15;
  } else {
    const int spw = 16 / L, sl = l15 / L;
    seq = (blockIdx.x * 8 + wave) * spw + sl;
    pos = l15 - sl * L;
  }
  const bool live = seq < sq.nseq;
  const long tok = live ? seq_token(sq, seq, pos) * C : 0;
  f32x4 res[2 * CB];
#pragma unroll
  for (int g = 0; g < 2 * CB; ++g) res[g] = *(const f32x4*)(x + tok + 16 * g + 4 * kk);

  // which keys (own wave: slot 4*kk+r; partner wave: its slot 4*kk+r) the query of this lane may see
  unsigned allow_own = 0u, allow_par = 0u;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = kk * 4 + r;
    bool ok_o, ok_p = false;
    if (paired) {
      ok_o = live && (!causal || j <= l15);
      ok_p = live && (!causal || half == 1);  // the partner holds the other half of the same sequence
    } else {
      const int sj = j / L, pj = j - sj * L, si = l15 / L;
      const bool live_j = ((int)(blockIdx.x * 8 + wave) * (16 / L) + sj) < sq.nseq;
      ok_o = (sj == si) && (!causal || pj <= pos) && live_j;
    }
    allow_own |= (ok_o ? 1u : 0u) << r;
    allow_par |= (ok_p ? 1u : 0u) << r;
  }

  int xo_c[CB], xo_h[HB];
#pragma unroll
  for (int kb = 0; kb < CB; ++kb) xo_c[kb] = swz_chunk(l15, kb * 4 + kk, CPR) << 4;
#pragma unroll
  for (int kb = 0; kb < HB; ++kb) xo_h[kb] = swz_chunk(l15, kb * 4 + kk, CPRH) << 4;
  const int row_c = l15 * CPR * 16, row_h = l15 * CPRH * 16;

  u32x4 xn[CB];
  norm_frags16<CB>(res, live, C, eps, xn);
  u32x4 of[MAXB];
  u32x4 qf_p, kf_p, vpk_p;  // head h-1: this wave's q / k fragments and packed V^T halves ({dt0.xy, dt1.xy})
  qf_p = kf_p = vpk_p = u32x4{0u, 0u, 0u, 0u};

  // attention of head h for this wave's 16 queries; keys/values: own 16 tokens (+ the partner's 16 when paired).  It is cut into
  // three stages so that the caller can drop them between the MFMAs of the NEXT weight tile: the softmax is VALU work (exp,
  // cross-lane max / sum) that otherwise alternates with, instead of hiding under, the matrix pipe.
  f32x4 at_so, at_sp;
  u32x4 at_vo, at_vq, at_pf;
  auto attend_qk = [&](int h) {
    u32x4 kf_q = u32x4{0u, 0u, 0u, 0u};
    at_vo = vpk_p;
    at_vq = u32x4{0u, 0u, 0u, 0u};
    if (paired) {
      const char* pb = mbox + (h & 1) * 16384 + (wave ^ 1) * 2048 + lane * 16;
      kf_q = *(const u32x4*)pb;
      at_vq = *(const u32x4*)(pb + 1024);
    }
    at_so = mfma_bf16(kf_p, qf_p, f32x4{0.f, 0.f, 0.f, 0.f});  // rows = own keys, column = query
    at_sp = f32x4{0.f, 0.f, 0.f, 0.f};
    if (paired) at_sp = mfma_bf16(kf_q, qf_p, at_sp);
  };
  const bool all_visible = paired ? (__all(allow_own == 0xfu) && __all(allow_par == 0xfu)) : false;   // wave-uniform: no mask work at all
  auto attend_softmax = [&]() {
    float m, sum;
    if (all_visible) {
      m = fmaxf(fmaxf(fmaxf(at_so[0], at_so[1]), fmaxf(at_so[2], at_so[3])), fmaxf(fmaxf(at_sp[0], at_sp[1]), fmaxf(at_sp[2], at_sp[3])));
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      sum = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        at_so[r] = __builtin_amdgcn_exp2f(at_so[r] - m);
        at_sp[r] = __builtin_amdgcn_exp2f(at_sp[r] - m);
        sum += at_so[r] + at_sp[r];
      }
    } else {
      m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if ((allow_own >> r) & 1u) m = fmaxf(m, at_so[r]);
        if ((allow_par >> r) & 1u) m = fmaxf(m, at_sp[r]);
      }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      sum = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        at_so[r] = ((allow_own >> r) & 1u) ? __builtin_amdgcn_exp2f(at_so[r] - m) : 0.0f;
        at_sp[r] = ((allow_par >> r) & 1u) ? __builtin_amdgcn_exp2f(at_sp[r] - m) : 0.0f;
        sum += at_so[r] + at_sp[r];
      }
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = sum > 0.0f ? __builtin_amdgcn_rcpf(sum) : 0.0f;
    at_pf = pack8(at_so * inv, at_sp * inv);  // k order: own keys then partner keys -- the same order as vt below
  };
  auto attend_pv = [&]() {
    const u32x4 vt0 = u32x4{at_vo[0], at_vo[1], at_vq[0], at_vq[1]}, vt1 = u32x4{at_vo[2], at_vo[3], at_vq[2], at_vq[3]};
    const f32x4 o0 = mfma_bf16(vt0, at_pf, f32x4{0.f, 0.f, 0.f, 0.f});  // O^T = V^T P^T
    const f32x4 o1 = mfma_bf16(vt1, at_pf, f32x4{0.f, 0.f, 0.f, 0.f});
    return pack8(o0, o1);
  };
  f32x4 pend[4];   // fc1 accumulators whose GELU + pack is deferred into the next tile's MFMA stream

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  static_for<NT>([&](auto t_c) {
    constexpr int t = decltype(t_c)::value;
    if constexpr (t + 1 < NT) { if (dbg != 2) glds_copy8(stream + tile_off(t + 1), smem + ((t + 1) & 1) * SLOT, tile_bytes(t + 1), tid); }
    const char* wt = smem + (t & 1) * SLOT;
    if (dbg == 1) {  // ablation: weight stream only
    } else if constexpr (t < NH) {
      // =============================== projections of head t, attention of head t-1 ======================
      const float* bias = (const float*)(wt + 96 * CPR * 16);
      f32x4 aqk[4], av[2];
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) aqk[ns] = *(const f32x4*)(bias + ns * 16 + kk * 4);  // start from the bias
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const float bv = bias[64 + dt * 16 + l15];
        av[dt] = f32x4{bv, bv, bv, bv};
      }
      // 6 row tiles (q: 0-1, k: 2-3, v: 4-5) x CB k-blocks, k-block outer so consecutive MFMAs hit different accumulators
      unsigned bt[CB];   // per-k-block LDS base of this tile: row + swizzled chunk; the row-tile offset is an immediate
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) bt[cb] = lds_addr(wt + row_c + xo_c[cb]);
      mfma_stream<6 * CB, RING>(
          [&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<(i % 6) * 16 * CPR * 16>{bt[i / 6]}; },
          [&](auto ic, const u32x4& wf) {
            constexpr int i = decltype(ic)::value, rt = i % 6, cb = i / 6;
            if constexpr (rt < 4) aqk[rt] = mfma_bf16(wf, xn[cb], aqk[rt]);
            else av[rt - 4] = mfma_bf16(xn[cb], wf, av[rt - 4]);  // roles swapped: D[token][feature]
            if constexpr (t > 0) {   // head t-1's attention rides this tile's MFMA stream
              constexpr int N6 = 6 * CB;
              if constexpr (i == N6 / 8) attend_qk(t - 1);
              if constexpr (i == (3 * N6) / 8) attend_softmax();
              if constexpr (i == (6 * N6) / 8) of[t - 1] = attend_pv();
            }
          });
      const u32x4 qf_n = pack8(aqk[0], aqk[1]), kf_n = pack8(aqk[2], aqk[3]);
      u32x4 vpk_n;
      vpk_n[0] = pack_bf16x2(av[0][0], av[0][1]); vpk_n[1] = pack_bf16x2(av[0][2], av[0][3]);
      vpk_n[2] = pack_bf16x2(av[1][0], av[1][1]); vpk_n[3] = pack_bf16x2(av[1][2], av[1][3]);
      if (paired) {  // publish this head's keys / values for the partner wave (read after this iteration's barrier)
        char* mb = mbox + (t & 1) * 16384 + wave * 2048 + lane * 16;
        *(u32x4*)mb = kf_n;
        *(u32x4*)(mb + 1024) = vpk_n;
      }
      qf_p = qf_n; kf_p = kf_n; vpk_p = vpk_n;
    } else {
      // =============================== 64 output features of a dense layer ==============================
      constexpr bool is_out = t < NH + TO, is_fc1 = !is_out && t < NH + TO + T1;
      constexpr int KB = is_fc1 ? CB : (is_out ? CB : HB);
      constexpr int cpr = KB * 4;
      const float* bias = (const float*)(wt + 64 * cpr * 16);
      constexpr int tb = is_fc1 ? t - NH - TO : (is_out ? t - NH : t - NH - TO - T1);
      // deferral needs room: the stream must be long enough, and an fc2 tile first reads of[2 (T1-1)] at MFMA 8 (T1-1)
      constexpr bool DEFER = (CB >= 8 && T1 >= 3);
      constexpr bool pend_gelu = DEFER && ((is_fc1 && tb > 0) || (!is_out && !is_fc1 && tb == 0));   // the previous tile was an fc1 tile
      constexpr int pend_tb = is_fc1 ? tb - 1 : T1 - 1;
      if constexpr (t == NH && CB < 8) {   // short stream: the last head's attention runs up front
        attend_qk(NH - 1);
        attend_softmax();
        of[NH - 1] = attend_pv();
      }
      f32x4 acc[4];
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) {
        const f32x4 b = *(const f32x4*)(bias + ns * 16 + kk * 4);
        if constexpr (is_fc1) acc[ns] = b;
        else acc[ns] = res[4 * tb + ns] + b;  // the residual rides the accumulator
      }
      unsigned bt[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) bt[kb] = lds_addr(wt + ((KB == CB) ? row_c + xo_c[kb < CB ? kb : 0] : row_h + xo_h[kb < HB ? kb : 0]));
      mfma_stream<4 * KB, RING>(
          [&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<(i % 4) * 16 * cpr * 16>{bt[i / 4]}; },
          [&](auto ic, const u32x4& wf) {
            constexpr int i = decltype(ic)::value, ns = i % 4, kb = i / 4;
            if constexpr (is_fc1) acc[ns] = mfma_bf16(wf, xn[kb < CB ? kb : 0], acc[ns]);
            else acc[ns] = mfma_bf16(wf, of[kb], acc[ns]);
            if constexpr (t == NH && CB >= 8) {   // the last head's attention rides the first out-proj tile (of[NH-1] is consumed from i = 4 (NH-1))
              if constexpr (i == 1) attend_qk(NH - 1);
              if constexpr (i == 6) attend_softmax();
              if constexpr (i == 12) of[NH - 1] = attend_pv();
            }
            if constexpr (pend_gelu) {  // GELU + pack of the previous fc1 tile under this tile's MFMAs; an fc2 tile reads of[2 pend_tb]
              if constexpr (i == 2 || i == 5 || i == 8 || i == 11) pend[(i - 2) / 3] = gelu_poly4<true>(pend[(i - 2) / 3]);  // from i = 8 pend_tb
              if constexpr (i == 13) { of[2 * pend_tb] = pack8(pend[0], pend[1]); of[2 * pend_tb + 1] = pack8(pend[2], pend[3]); }
            }
          });
      if constexpr (is_fc1) {
        if constexpr (DEFER) {
#pragma unroll
          for (int ns = 0; ns < 4; ++ns) pend[ns] = acc[ns];
        } else {
#pragma unroll
          for (int ns = 0; ns < 4; ++ns) acc[ns] = gelu_poly4<true>(acc[ns]);
          of[2 * tb] = pack8(acc[0], acc[1]);
          of[2 * tb + 1] = pack8(acc[2], acc[3]);
        }
      } else {
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) res[4 * tb + ns] = acc[ns];
        if constexpr (is_out && tb == TO - 1) norm_frags16<CB>(res, live, C, eps, xn);  // x1 complete: LayerNorm2
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  });
  if (live) {
#pragma unroll
    for (int g = 0; g < 2 * CB; ++g) *(f32x4*)(x + tok + 16 * g + 4 * kk) = res[g];
  }
}

// ------------------------------------------------------------------------------------------------------
// Stream packing.  One thread per 16-byte chunk (8 bf16) of one tile row, or per bias float.
// ------------------------------------------------------------------------------------------------------
// position p inside a 32-block of a k-permuted row holds source feature c:  p = 8*kk + 4*dt + r
__device__ __forceinline__ int kperm_src(int p) {
  const int blk = p >> 5, q = p & 31, kk = q >> 3, dt = (q >> 2) & 1, r = q & 3;
  return blk * 32 + dt * 16 + kk * 4 + r;
}

// rows of `w` (N_src x K, row-major fp32) selected by row_of(tile_row) -> one tile of `rows` rows
__device__ __forceinline__ void pack_row_chunk(const float* __restrict__ w, int src_row, int K, int c, bool perm,
                                               const float* __restrict__ gamma, float scale, u32x4* dst) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = c * 8 + i;
    const int k = perm ? kperm_src(p) : p;
    float xv = w[(long)src_row * K + k];
    if (gamma) xv *= gamma[k];
    v[i] = xv * scale;
  }
  u32x4 o;
  o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
  *dst = o;
}

__device__ __forceinline__ float folded_bias(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ beta,
                                             int row, int K) {
  float s = b ? b[row] : 0.0f;
  if (beta)
    for (int k = 0; k < K; ++k) s += w[(long)row * K + k] * beta[k];
  return s;
}

// one stream per block: NH head tiles, C/64 Wo tiles, HID/64 W1 tiles, C/64 W2 tiles; one thread block per tile
__global__ void pack_block_stream_kernel(const float* __restrict__ w_in, const float* __restrict__ b_in, const float* __restrict__ g1,
                                         const float* __restrict__ be1, const float* __restrict__ w_out,
                                         const float* __restrict__ b_out, const float* __restrict__ w1, const float* __restrict__ b1,
                                         const float* __restrict__ g2, const float* __restrict__ be2, const float* __restrict__ w2,
                                         const float* __restrict__ b2, int C, int HID, char* __restrict__ dst) {
  const int NH = C / 32, CPR = C / 8, CPRH = HID / 8, TO = C / 64, T1 = HID / 64;
  const long tileh = 96 * CPR * 16 + BIAS_BYTES, tileo = 64 * CPR * 16 + BIAS_BYTES, tile2 = 64 * CPRH * 16 + BIAS_BYTES;
  const float qscale = 0.17677669529663687f * 1.44269504088896340736f;  // log2(e) / sqrt(32): the kernels' softmax is exp2 of the scores
  const int t = blockIdx.x;
  int kind, tb;  // 0 head, 1 Wo, 2 W1, 3 W2
  long off;
  if (t < NH) { kind = 0; tb = t; off = t * tileh; }
  else if (t < NH + TO) { kind = 1; tb = t - NH; off = NH * tileh + tb * tileo; }
  else if (t < NH + TO + T1) { kind = 2; tb = t - NH - TO; off = NH * tileh + TO * tileo + tb * tileo; }
  else { kind = 3; tb = t - NH - TO - T1; off = NH * tileh + TO * tileo + T1 * tileo + tb * tile2; }
  char* base = dst + off;
  const int rows = kind == 0 ? 96 : 64;
  const int cpr = kind == 3 ? CPRH : CPR;
  const int K = kind == 3 ? HID : C;
  for (int idx = threadIdx.x; idx < rows * cpr; idx += blockDim.x) {
    const int r = idx / cpr, c = idx % cpr;
    u32x4* d = (u32x4*)base + (long)r * cpr + swz_chunk(r, c, cpr);
    if (kind == 0) {
      const int part = r / 32, src = part * C + tb * 32 + (r % 32);  // q | k | v rows of head tb
      pack_row_chunk(w_in, src, K, c, true, g1, part == 0 ? qscale : 1.0f, d);
    } else if (kind == 1) {
      pack_row_chunk(w_out, tb * 64 + r, K, c, true, nullptr, 1.0f, d);
    } else if (kind == 2) {
      pack_row_chunk(w1, tb * 64 + r, K, c, true, g2, 1.0f, d);
    } else {
      pack_row_chunk(w2, tb * 64 + r, K, c, true, nullptr, 1.0f, d);
    }
  }
  float* bias = (float*)(base + (long)rows * cpr * 16);
  for (int r = threadIdx.x; r < 256; r += blockDim.x) {
    float v = 0.0f;
    if (r < rows) {
      if (kind == 0) {
        const int part = r / 32, src = part * C + tb * 32 + (r % 32);
        v = folded_bias(w_in, b_in, be1, src, C) * (part == 0 ? qscale : 1.0f);
      } else if (kind == 1) {
        v = b_out[tb * 64 + r];
      } else if (kind == 2) {
        v = folded_bias(w1, b1, be2, tb * 64 + r, C);
      } else {
        v = b2[tb * 64 + r];
      }
    }
    bias[r] = v;
  }
}

template <int CB, int HB>
void launch_block16(float* x, const char* stream, const TanteSeq& sq, int causal, float eps, hipStream_t s) {
  constexpr int TH = 96 * CB * 4 * 16 + BIAS_BYTES, T2 = 64 * HB * 4 * 16 + BIAS_BYTES;
  constexpr int SLOT = TH > T2 ? TH : T2;
  constexpr int LDS = 2 * SLOT + 2 * 16384;
  const int per_wg = sq.L == 32 ? 4 : 8 * (16 / sq.L);  // sequences per workgroup
  static TantePerDevice attr;
  attr.once([&] {
    hipFuncSetAttribute((const void*)fused_block16_kernel<CB, HB>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  static const int dbg = tante_ablate_env("TANTE_BLOCK_DEBUG");  // -DTANTE_ABLATE builds only
  hipLaunchKernelGGL((fused_block16_kernel<CB, HB>), dim3((sq.nseq + per_wg - 1) / per_wg), dim3(512), LDS, s, x, stream, sq, causal, eps, dbg);
}

template <int CB, int HB>
void launch_block(float* x, const char* stream, const TanteSeq& sq, int causal, float eps, hipStream_t s) {
  const int which = tante_opt("TANTE_BLOCK_KERNEL", 16);
  if (which == 16 && (sq.L == 32 || 16 % sq.L == 0)) return launch_block16<CB, HB>(x, stream, sq, causal, eps, s);
  constexpr int TH = 96 * CB * 4 * 16 + BIAS_BYTES, T2 = 64 * HB * 4 * 16 + BIAS_BYTES;
  constexpr int SLOT = TH > T2 ? TH : T2;
  const int spw = 32 / sq.L;
  const int waves = (sq.nseq + spw - 1) / spw;
  static TantePerDevice attr;
  attr.once([&] {
    hipFuncSetAttribute((const void*)fused_block_kernel<CB, HB>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SLOT);
  });
  hipLaunchKernelGGL((fused_block_kernel<CB, HB>), dim3((waves + 3) / 4), dim3(256), 3 * SLOT, s, x, stream, sq, causal, eps);
}

}  // namespace

// the round-1 kernels of this file (token-stationary waves, weights streamed through LDS): C in {64, 128, 256}, L | 32
static int block_ts_supported(int C, int n_head, int hidden, int L) {
  if (n_head <= 0 || C % n_head || C / n_head != 32) return 0;
  if (C != 64 && C != 128 && C != 256) return 0;
  if (hidden != C && hidden != 2 * C) return 0;
  if (hidden > 256 && C == 256) return 0;  // hidden 512 at C 256 does not fit the register file
  if (L <= 0 || L > 32 || 32 % L) return 0;
  return 1;
}

static int64_t block_ts_stream_bytes(int C, int hidden) {
  const long cpr = C / 8, cprh = hidden / 8;
  return (long)(C / 32) * (96 * cpr * 16 + BIAS_BYTES) + (long)(C / 64 + hidden / 64) * (64 * cpr * 16 + BIAS_BYTES) +
         (long)(C / 64) * (64 * cprh * 16 + BIAS_BYTES);
}

// which kernel family runs a shape: the feature-sliced kernel (block_sliced.hip) wherever it applies (C = 256, 8 heads, hidden 256,
// any L <= 128); TANTE_BLOCK_KERNEL=16 / 32 forces the round-1 kernels for A/B timing (both compute the same function)
static bool use_fs(int C, int n_head, int hidden, int L, int causal) {
  const int which = tante_opt("TANTE_BLOCK_KERNEL", 0);
  if (!tante_fs_supported(C, n_head, hidden, L, causal)) return false;
  return which == 0 || !block_ts_supported(C, n_head, hidden, L);
}

extern "C" int tante_block_fused_supported(int C, int n_head, int hidden, int L) {
  return block_ts_supported(C, n_head, hidden, L) || tante_fs_supported(C, n_head, hidden, L, 0);
}

// the packed block = [round-1 stream | feature-sliced stream] (the second part only for C = 256, hidden 256)
extern "C" int64_t tante_block_stream_bytes(int C, int hidden) { return block_ts_stream_bytes(C, hidden) + tante_fs_stream_bytes(C, hidden); }

extern "C" int tante_pack_block(const float* ln1_w, const float* ln1_b, const float* in_w, const float* in_b, const float* out_w,
                                const float* out_b, const float* ln2_w, const float* ln2_b, const float* fc1_w,
                                const float* fc1_b, const float* fc2_w, const float* fc2_b, int C, int hidden,
                                void* block_stream, void* stream) {
  if (!ln1_w || !ln1_b || !in_w || !in_b || !out_w || !out_b || !ln2_w || !ln2_b || !fc1_w || !fc1_b || !fc2_w || !fc2_b ||
      !block_stream)
    TANTE_FAIL(-1, "tante_pack_block: null pointer");
  if (!block_ts_supported(C, C / 32, hidden, 32)) TANTE_FAIL(-2, "tante_pack_block: unsupported C=%d hidden=%d", C, hidden);
  hipLaunchKernelGGL(pack_block_stream_kernel, dim3(C / 32 + 2 * (C / 64) + hidden / 64), dim3(256), 0, (hipStream_t)stream, in_w,
                     in_b, ln1_w, ln1_b, out_w, out_b, fc1_w, fc1_b, ln2_w, ln2_b, fc2_w, fc2_b, C, hidden, (char*)block_stream);
  TANTE_CHECK_LAUNCH();
  if (tante_fs_stream_bytes(C, hidden) > 0) {
    tante_fs_pack(ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b,
                  (char*)block_stream + block_ts_stream_bytes(C, hidden), (hipStream_t)stream);
    TANTE_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int tante_block_fused_tprop(float* x, const void* block_stream, int C, int n_head, int hidden, const TanteSeq* seq, int causal, float eps,
                                       const float* tprop, void* stream) {
  if (!x || !block_stream || !seq || !tprop) TANTE_FAIL(-1, "tante_block_fused_tprop: null pointer");
  if (seq->L != 4 || !tante_block_fused_supported(C, n_head, hidden, seq->L) || !use_fs(C, n_head, hidden, seq->L, causal))
    TANTE_FAIL(-2, "tante_block_fused_tprop: the feature-sliced kernel at L = 4 only (C=%d heads=%d hidden=%d L=%d)", C, n_head, hidden, seq->L);
  if (((uintptr_t)x % 16) || ((uintptr_t)block_stream % 16)) TANTE_FAIL(-1, "tante_block_fused_tprop: buffers must be 16-byte aligned");
  if (tante_fs_launch(x, (const char*)block_stream + block_ts_stream_bytes(C, hidden), *seq, causal, eps, (hipStream_t)stream, nullptr, tprop) != 0)
    TANTE_FAIL(-2, "tante_block_fused_tprop: too many sequences (%d)", seq->nseq);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_block_fused(float* x, const void* block_stream, int C, int n_head, int hidden, const TanteSeq* seq, int causal,
                                 float eps, void* stream) {
  if (!x || !block_stream || !seq) TANTE_FAIL(-1, "tante_block_fused: null pointer");
  if (!tante_block_fused_supported(C, n_head, hidden, seq->L))
    TANTE_FAIL(-2, "tante_block_fused: unsupported shape C=%d heads=%d hidden=%d L=%d", C, n_head, hidden, seq->L);
  if (((uintptr_t)x % 16) || ((uintptr_t)block_stream % 16)) TANTE_FAIL(-1, "tante_block_fused: buffers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const char* st = (const char*)block_stream;
  if (use_fs(C, n_head, hidden, seq->L, causal)) {
    if (tante_fs_launch(x, st + block_ts_stream_bytes(C, hidden), *seq, causal, eps, s) != 0)
      TANTE_FAIL(-2, "tante_block_fused: too many sequences (%d)", seq->nseq);
    TANTE_CHECK_LAUNCH();
    return 0;
  }
  const int key = (C / 32) * 100 + hidden / 32;
  switch (key) {
    case 202: launch_block<2, 2>(x, st, *seq, causal, eps, s); break;
    case 204: launch_block<2, 4>(x, st, *seq, causal, eps, s); break;
    case 404: launch_block<4, 4>(x, st, *seq, causal, eps, s); break;
    case 408: launch_block<4, 8>(x, st, *seq, causal, eps, s); break;
    case 808: launch_block<8, 8>(x, st, *seq, causal, eps, s); break;
    default: TANTE_FAIL(-2, "tante_block_fused: no instantiation for C=%d hidden=%d", C, hidden);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_block_fused_train(const float* x, const void* block_stream, int C, int n_head, int hidden, const TanteSeq* seq, int causal,
                                       float eps, const TanteBlockTrain* tr, void* stream) {
  if (!x || !block_stream || !seq || !tr || !tr->out || !tr->xh1 || !tr->o || !tr->xh2 || !tr->hpre || !tr->act || !tr->st1 ||
      !tr->st2)      // x1 may be null: only the per-operator backward (LayerNorm2 from its input) reads it; qkv may be null: tante_block_bwd_fused recomputes it
    TANTE_FAIL(-1, "tante_block_fused_train: null pointer");
  if (!tante_fs_supported(C, n_head, hidden, seq->L, causal) || seq->L > 64)
    TANTE_FAIL(-2, "tante_block_fused_train: unsupported shape C=%d heads=%d hidden=%d L=%d", C, n_head, hidden, seq->L);
  if (tr->p_drop < 0.0f || tr->p_drop >= 1.0f) TANTE_FAIL(-1, "tante_block_fused_train: dropout probability %f", (double)tr->p_drop);
  const char* st = (const char*)block_stream + block_ts_stream_bytes(C, hidden);
  if (tante_fs_launch((float*)x, st, *seq, causal, eps, (hipStream_t)stream, tr) != 0)
    TANTE_FAIL(-2, "tante_block_fused_train: launch refused (L=%d, %d sequences)", seq->L, seq->nseq);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_pack_block_train_multi(const TanteBlockWeights* blocks, int n, int C, int hidden, void* stream) {
  if (!blocks || n <= 0) TANTE_FAIL(-1, "tante_pack_block_train_multi: bad argument");
  if (tante_fs_stream_bytes(C, hidden) == 0) TANTE_FAIL(-2, "tante_pack_block_train_multi: unsupported C=%d hidden=%d", C, hidden);
  for (int g = 0; g < n; g += TANTE_FSP_MAX) {
    const int m = n - g < TANTE_FSP_MAX ? n - g : TANTE_FSP_MAX;
    const float* params[TANTE_FSP_MAX][8];
    char* dst[TANTE_FSP_MAX];
    for (int e = 0; e < m; ++e) {
      const TanteBlockWeights& b = blocks[g + e];
      if (!b.in_w || !b.in_b || !b.out_w || !b.out_b || !b.fc1_w || !b.fc1_b || !b.fc2_w || !b.fc2_b || !b.block_stream)
        TANTE_FAIL(-1, "tante_pack_block_train_multi: null pointer in entry %d", g + e);
      const float* q[8] = {b.in_w, b.in_b, b.out_w, b.out_b, b.fc1_w, b.fc1_b, b.fc2_w, b.fc2_b};
      for (int k = 0; k < 8; ++k) params[e][k] = q[k];
      dst[e] = (char*)b.block_stream + block_ts_stream_bytes(C, hidden);
    }
    tante_fs_pack_folded_multi(params, dst, m, (hipStream_t)stream);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_pack_block_train(const float* in_w_folded, const float* in_b_folded, const float* out_w, const float* out_b,
                                      const float* fc1_w_folded, const float* fc1_b_folded, const float* fc2_w, const float* fc2_b, int C,
                                      int hidden, void* block_stream, void* stream) {
  if (!in_w_folded || !in_b_folded || !out_w || !out_b || !fc1_w_folded || !fc1_b_folded || !fc2_w || !fc2_b || !block_stream)
    TANTE_FAIL(-1, "tante_pack_block_train: null pointer");
  if (tante_fs_stream_bytes(C, hidden) == 0) TANTE_FAIL(-2, "tante_pack_block_train: unsupported C=%d hidden=%d", C, hidden);
  tante_fs_pack_folded(in_w_folded, in_b_folded, out_w, out_b, fc1_w_folded, fc1_b_folded, fc2_w, fc2_b,
                       (char*)block_stream + block_ts_stream_bytes(C, hidden), (hipStream_t)stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}
