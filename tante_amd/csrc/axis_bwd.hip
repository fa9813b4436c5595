// Backward of an axis propagator  y = x + W2 gelu(W1 x + b1) + b2  (attn_backbone.py:111-119, 140-145) WITH its weight gradients, one launch,
// bf16 operands on the matrix cores -- the training path of the H / W propagators (axis lengths 16, 32, 48).
//
// Before: tante_axis_mlp_bwd (one lane per column, fp32 VALU: 3 n x n mat-vecs per column -- 64 us at n = 48, VALU-bound) wrote dx AND
// materialised h = gelu(pre) and dpre (2 x 25 MB) for two tante_axis_wgrad launches (fp32 MFMA, 2 x 24 - 45 us) that read them back with x
// and dy: 154 us and 325 MB per W-axis call at cfg3.  Here a workgroup keeps a tile of x and dy in LDS (2 line groups = 2 x 32 inner
// elements, all n positions), computes
//     pre = W1 x + b1,  da = W2^T dy,  a = gelu(pre),  dpre = da gelu'(pre),  dx = dy + W1^T dpre
// as three chained MFMA products per 16 columns (the layouts of axis_hw_exact_kernel, pointwise.hip: lanes own channels, the k slot (kk, e)
// of a 32-wide chunk ks holds position 32 ks + 4 e + kk), leaves a and dpre in two more LDS planes, and contracts
//     dW1 += dpre x^T,  dW2 += dy a^T,  db1 += sum dpre,  db2 += sum dy      (over the tile's columns)
// from the four planes: the contraction index is the CHANNEL, contiguous in every plane, so an operand fragment (row = position or hidden
// unit, 8 consecutive channels) is two 16-byte LDS reads.  HBM: x and dy in, dx out -- 75 MB per call.
//
// LDS planes [group][position][32 channels] fp32, 16-byte chunk c of position p stored at c ^ ((p >> 1) & 7): rows are 128 bytes apart, so
// the 16 rows of a fragment read would otherwise sit on two bank groups; the planes arrive by LDS-DMA (lane-linear), so the swizzle is
// applied on the global SOURCE side (as in wgrad_tr_kernel).  Weight gradients accumulate in registers across the tiles of a persistent
// workgroup (wave w owns row tile w of dW1 and of dW2, wave 3 the two bias gradients) and leave as one partial per workgroup, summed by a
// second small kernel (atomics without a workspace).
#include "common.hip.h"
#include <stdlib.h>

namespace {

constexpr int AB_CT = 32;                 // inner elements per line group
constexpr int AB_NG = 2;                  // line groups per tile
constexpr int AB_NT = 256;                // threads

__host__ __device__ constexpr int ab_gs(int N) { return N * AB_CT + 32; }             // group stride (floats): 32 (mod 64)
__host__ __device__ constexpr int ab_kb(int MT) { return (MT + 1) / 2; }
__host__ __device__ constexpr int ab_wbytes(int MT) { return 3 * MT * ab_kb(MT) * 1024 + 16 * MT * 4; }
__host__ __device__ constexpr int ab_lds(int MT) { return 4 * AB_NG * ab_gs(16 * MT) * 4 + ab_wbytes(MT); }

__device__ __forceinline__ int ab_sw(int p) { return (p >> 1) & 7; }

__device__ __forceinline__ f32x4 ab_mfma(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// three n x n matrices as bf16 A-operand fragments in LDS (fragment (mt, ks) of matrix m at ((m MT + mt) KB + ks) * 1024 bytes, lane l at
// + 16 l, k slot e at + 2 e), then b1 as floats:
//   m = 0: W1      rows = hidden j,   k slot <-> position p = 32 ks + 4 e + kk          (pre = W1 x)
//   m = 1: W1^T    rows = position p, k slot <-> hidden j in accumulator order          (dx += W1^T dpre)
//   m = 2: W2^T    rows = hidden j,   k slot <-> position p                             (da = W2^T dy)
template <int MT>
__device__ __forceinline__ void ab_stage(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, char* wst, int tid) {
  constexpr int N = 16 * MT, KB = ab_kb(MT);
  unsigned* z = (unsigned*)wst;
  for (int i = tid; i < 3 * MT * KB * 256; i += AB_NT) z[i] = 0u;       // odd MT: the upper half of the last k chunk stays zero
  __syncthreads();
  unsigned short* f = (unsigned short*)wst;
  for (int i = tid; i < N * N; i += AB_NT) {
    const int row = i / N, col = i - row * N;
    const int mt = row >> 4, l15 = row & 15;
    const int ks1 = col >> 5, e1 = (col & 31) >> 2, kk1 = col & 3;                             // position-ordered k slot of `col`
    const int tile = col >> 4, ks2 = tile >> 1, kk2 = (col & 15) >> 2, e2 = (tile & 1) * 4 + (col & 3);   // accumulator-ordered k slot of `col`
    __bf16 v;
    v = (__bf16)w1[i];                               // W1[row = j][col = p]
    f[((0 * MT + mt) * KB + ks1) * 512 + (kk1 * 16 + l15) * 8 + e1] = __builtin_bit_cast(unsigned short, v);
    v = (__bf16)w1[col * N + row];                   // W1^T[row = p][col = j]
    f[((1 * MT + mt) * KB + ks2) * 512 + (kk2 * 16 + l15) * 8 + e2] = __builtin_bit_cast(unsigned short, v);
    v = (__bf16)w2[col * N + row];                   // W2^T[row = j][col = p] = W2[p][j]
    f[((2 * MT + mt) * KB + ks1) * 512 + (kk1 * 16 + l15) * 8 + e1] = __builtin_bit_cast(unsigned short, v);
  }
  float* bs = (float*)(wst + 3 * MT * KB * 1024);
  if (tid < N) bs[tid] = b1[tid];
}

template <int MT>
__global__ __launch_bounds__(AB_NT, 2) void axis_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ dy, long outer, long inner,
                                                                  const float* __restrict__ w1, const float* __restrict__ b1,
                                                                  const float* __restrict__ w2, float* __restrict__ dx, float* __restrict__ dW1,
                                                                  float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2,
                                                                  float* __restrict__ ws) {
  constexpr int N = 16 * MT, KB = ab_kb(MT), GS = ab_gs(N), PL = AB_NG * GS, NE = 8 * KB;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* const X = sm;               // x, later untouched until the tile is done
  float* const G = sm + PL;          // dy; becomes dx at the end of the tile
  float* const A = sm + 2 * PL;      // gelu(pre)
  float* const D = sm + 3 * PL;      // dpre
  char* const wst = (char*)(sm + 4 * PL);
  const int tid = threadIdx.x, lane = tid & 63, kk = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ab_stage<MT>(w1, b1, w2, wst, tid);
  const u32x4* wf = (const u32x4*)wst;
  const float* bs = (const float*)(wst + 3 * MT * KB * 1024);
  const long tiles_per_outer = inner / (AB_CT * AB_NG), n_tiles = outer * tiles_per_outer;
  const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc1[MT], acc2[MT];          // wave < MT: dW1[row tile wave][.], dW2[row tile wave][.];  wave 3: db1[.], db2[.]
#pragma unroll
  for (int t = 0; t < MT; ++t) acc1[t] = acc2[t] = zero4;
  __syncthreads();

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long o = tile / tiles_per_outer;
    const long i0 = (tile - o * tiles_per_outer) * (AB_CT * AB_NG);
    const long gbase = o * N * inner + i0;                      // element (position p, group g, channel c) at gbase + p * inner + 32 g + c
    // ---- the tile: one DMA instruction = 8 positions x 128 bytes of one group (lane = (position, chunk slot)) ---------------------
    for (int q = wave; q < AB_NG * (N / 8) * 2; q += AB_NT / 64) {
      const int pl = q & 1, r = q >> 1, g = r / (N / 8), j = r - g * (N / 8);
      const int p = j * 8 + (lane >> 3), ch = (lane & 7) ^ ab_sw(p);
      const float* src = (pl ? dy : x) + gbase + (long)p * inner + g * AB_CT + ch * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)((pl ? G : X) + g * GS + j * 256), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0)
    __syncthreads();
    // ---- backward of the MLP: unit = (group, channel parity a); lane (l15, kk) owns channel 2 l15 + a ----------------------------------
    f32x4 outk[MT];                      // W1^T dpre of this wave's unit, kept until the planes' readers are done
    int ug = 0, uc = 0;
    for (int u = wave; u < 2 * AB_NG; u += AB_NT / 64) {
      const int g = u >> 1, c = 2 * l15 + (u & 1);
      ug = g; uc = c;
      const int coff = (c >> 2), cw = c & 3;                     // chunk and word of this channel inside a position's 128 bytes
#ifdef AB_SKIP_BWD
      for (int mt = 0; mt < MT; ++mt) outk[mt] = zero4;
      continue;
#endif
      u32x4 xb[KB], gb[KB];
#pragma unroll
      for (int ks = 0; ks < KB; ++ks) {
        float xv[8], gv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int p = ks * 32 + 4 * e + kk;
          if (ks * 32 + 4 * e < N) {
            const int off = g * GS + p * AB_CT + ((coff ^ ab_sw(p)) << 2) + cw;
            xv[e] = X[off];
            gv[e] = G[off];
          } else {
            xv[e] = gv[e] = 0.f;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          xb[ks][q] = pack_bf16x2(xv[2 * q], xv[2 * q + 1]);
          gb[ks][q] = pack_bf16x2(gv[2 * q], gv[2 * q + 1]);
        }
      }
      f32x4 pre[2 * KB], da[2 * KB];
#pragma unroll
      for (int mt = 0; mt < 2 * KB; ++mt) {
        pre[mt] = mt < MT ? *(const f32x4*)(bs + (mt < MT ? mt : 0) * 16 + 4 * kk) : zero4;
        da[mt] = zero4;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) {
          pre[mt] = ab_mfma(wf[((0 * MT + mt) * KB + ks) * 64 + lane], xb[ks], pre[mt]);
          da[mt] = ab_mfma(wf[((2 * MT + mt) * KB + ks) * 64 + lane], gb[ks], da[mt]);
        }
      // a = gelu(pre), dpre = da gelu'(pre) for hidden units 16 mt + 4 kk + r of this lane's channel -> the A and D planes
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 act = gelu_poly4<false>(pre[mt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          da[mt][r] *= gelu_erf_grad_fast(pre[mt][r]);
          const int j = 16 * mt + 4 * kk + r;
          const int off = g * GS + j * AB_CT + ((coff ^ ab_sw(j)) << 2) + cw;
          A[off] = act[r];
          D[off] = da[mt][r];
        }
      }
      u32x4 pb[KB];
#pragma unroll
      for (int ks = 0; ks < KB; ++ks) {
        const f32x4 lo = da[2 * ks], hi = da[2 * ks + 1];
        pb[ks] = u32x4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        outk[mt] = zero4;
#pragma unroll
        for (int ks = 0; ks < KB; ++ks) outk[mt] = ab_mfma(wf[((1 * MT + mt) * KB + ks) * 64 + lane], pb[ks], outk[mt]);
      }
    }
    __syncthreads();
    // ---- weight gradients of the tile: contraction over the 32 channels of each group ---------------------------------------------------
    auto frag = [&](const float* P, int g, int t) {      // rows 16 t + l15 of plane P, channels 8 kk .. 8 kk + 7, as a bf16 operand
      const int q = 16 * t + l15;
      const float* row = P + g * GS + q * AB_CT;
      const f32x4 lo = *(const f32x4*)(row + (((2 * kk) ^ ab_sw(q)) << 2)), hi = *(const f32x4*)(row + (((2 * kk + 1) ^ ab_sw(q)) << 2));
      return u32x4{pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
    };
#ifndef AB_SKIP_WGRAD
    if (wave < MT) {
#pragma unroll
      for (int g = 0; g < AB_NG; ++g) {
        const u32x4 fd = frag(D, g, wave), fg = frag(G, g, wave);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          acc1[t] = ab_mfma(fd, frag(X, g, t), acc1[t]);      // dW1[j in tile wave][p in tile t]
          acc2[t] = ab_mfma(fg, frag(A, g, t), acc2[t]);      // dW2[p in tile wave][j in tile t]
        }
      }
    } else if (wave == 3) {
#pragma unroll
      for (int g = 0; g < AB_NG; ++g)
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          acc1[t] = ab_mfma(frag(D, g, t), ones, acc1[t]);    // db1[16 t + 4 kk + r] in every column
          acc2[t] = ab_mfma(frag(G, g, t), ones, acc2[t]);    // db2
        }
    }
#endif
    __syncthreads();
    // ---- dx = dy + W1^T dpre into the G plane, then out in row form ----------------------------------------------------------------------
    if (wave < 2 * AB_NG) {
      const int coff = uc >> 2, cw = uc & 3;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = 16 * mt + 4 * kk + r;
          G[ug * GS + p * AB_CT + ((coff ^ ab_sw(p)) << 2) + cw] += outk[mt][r];
        }
    }
    __syncthreads();
    for (int q = wave; q < AB_NG * (N / 8); q += AB_NT / 64) {
      const int g = q / (N / 8), j = q - g * (N / 8);
      const int p = j * 8 + (lane >> 3), ch = lane & 7;
      const f32x4 v = *(const f32x4*)(G + g * GS + p * AB_CT + ((ch ^ ab_sw(p)) << 2));
      *(f32x4*)(dx + gbase + (long)p * inner + g * AB_CT + ch * 4) = v;
    }
    __syncthreads();      // the planes are free for the next tile's DMA
  }
  // ---- the workgroup's partial gradients --------------------------------------------------------------------------------------------------
#ifdef AB_SKIP_EPILOGUE
  if (acc1[0][0] != 12345.f) return;
#endif
  if (ws) {
    // the partial goes to the workgroup's slab ([dW1][dW2][db1][db2], plain stores) and ab_reduce_kernel sums the slabs: every workgroup
    // adding its 4 704 values atomically was most of this kernel's time (512 same-address atomics per value: 83 -> 57 us at n = 48 when
    // they went), and a last-arriver reduction inside the kernel (as tante_axis_wgrad_ws does for its 2 352 values) left a serial tail of
    // 19 rounds of 16 dependent loads on the last workgroup of every group (24 of the 57 us)
    constexpr int SLABF = 2 * N * N + 2 * N;
    float* mine = ws + (long)blockIdx.x * SLABF;
    if (wave < MT) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * wave + 4 * kk + r, col = 16 * t + l15;
          mine[row * N + col] = acc1[t][r];
          mine[N * N + row * N + col] = acc2[t][r];
        }
    } else if (wave == 3 && l15 == 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          mine[2 * N * N + 16 * t + 4 * kk + r] = acc1[t][r];
          mine[2 * N * N + N + 16 * t + 4 * kk + r] = acc2[t][r];
        }
    }
    return;
  }
  if (wave < MT) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + 4 * kk + r, col = 16 * t + l15;
        atomicAdd(&dW1[row * N + col], acc1[t][r]);
        atomicAdd(&dW2[row * N + col], acc2[t][r]);
      }
  } else if (wave == 3 && l15 == 0) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        atomicAdd(&db1[16 * t + 4 * kk + r], acc1[t][r]);
        atomicAdd(&db2[16 * t + 4 * kk + r], acc2[t][r]);
      }
  }
}

// second stage: value i of the slabs, summed over a chunk of the workgroups (grid.y chunks), added into the gradients
__global__ __launch_bounds__(256) void ab_reduce_kernel(const float* __restrict__ ws, int n_slab, int slabf, int nn, int n, float* __restrict__ dW1,
                                                        float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= slabf) return;
  const int per = (n_slab + gridDim.y - 1) / gridDim.y, s0 = blockIdx.y * per, s1 = min(n_slab, s0 + per);
  const float* p = ws + (long)s0 * slabf + i;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = s0;
  for (; s + 8 <= s1; s += 8, p += 8L * slabf) {      // eight loads in flight
    const float v0 = p[0], v1 = p[slabf], v2 = p[2L * slabf], v3 = p[3L * slabf], v4 = p[4L * slabf], v5 = p[5L * slabf], v6 = p[6L * slabf],
                v7 = p[7L * slabf];
    a0 += v0 + v4; a1 += v1 + v5; a2 += v2 + v6; a3 += v3 + v7;
  }
  for (; s < s1; ++s, p += slabf) a0 += p[0];
  if (s0 >= s1) return;
  float* dst = i < nn ? dW1 + i : (i < 2 * nn ? dW2 + (i - nn) : (i < 2 * nn + n ? db1 + (i - 2 * nn) : db2 + (i - 2 * nn - n)));
  atomicAdd(dst, (a0 + a1) + (a2 + a3));
}

// ---- the temporal propagator (n = 4): the same fusion on the vector units, fp32 throughout -------------------------------------------------
// (instantiated for N = 4 only: at N = 8 the per-thread state -- two weight matrices, 144 partial sums -- is 477 registers, and that build
// faulted on the GPU; no configuration has such an axis, so it stays on the three-launch path)
// A thread owns 4 consecutive inner elements for all n positions (float4 streams of x and dy, stride `inner`); the n x n products are
// 3 n^2 FMAs per element -- nothing at n = 4 -- and the parameter gradients are per-thread partial sums (2 n^2 + 2 n values), reduced
// over the wave, the workgroup and then as ab_reduce_kernel's slabs.  Before: tante_axis_mlp_bwd wrote gelu(pre) and dpre (2 x 25 MB)
// for two tante_axis_wgrad launches: 57 us and 225 MB per call at cfg3 against 75 MB here.  Exact erf GELU (the expression of the
// kernel it replaces), so it serves the fp32 parity path too.
template <int N>
__global__ __launch_bounds__(256) void axis_bwd_small_kernel(const float* __restrict__ x, const float* __restrict__ dy, long outer, long inner4,
                                                             const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w2, float* __restrict__ dx, float* __restrict__ dW1,
                                                             float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2,
                                                             float* __restrict__ ws) {
  constexpr int NV = 2 * N * N + 2 * N;
  __shared__ float part[4][NV];
  float W1[N][N], W2[N][N], B1[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    B1[j] = b1[j];
#pragma unroll
    for (int p = 0; p < N; ++p) { W1[j][p] = w1[j * N + p]; W2[j][p] = w2[j * N + p]; }
  }
  float g[NV];          // [dW1 (j, p)][dW2 (p, j)][db1 (j)][db2 (p)]
#pragma unroll
  for (int i = 0; i < NV; ++i) g[i] = 0.f;
  const long total = outer * inner4;
  for (long col = (long)blockIdx.x * 256 + threadIdx.x; col < total; col += (long)gridDim.x * 256) {
    const long o = col / inner4, i4 = col - o * inner4;
    const f32x4* xp = (const f32x4*)x + o * N * inner4 + i4;
    const f32x4* gp = (const f32x4*)dy + o * N * inner4 + i4;
    f32x4 xv[N], gy[N], gx[N];
#pragma unroll
    for (int p = 0; p < N; ++p) { xv[p] = xp[p * inner4]; gy[p] = gp[p * inner4]; gx[p] = gy[p]; }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      f32x4 pre = f32x4{B1[j], B1[j], B1[j], B1[j]}, da = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < N; ++p) { pre += W1[j][p] * xv[p]; da += W2[p][j] * gy[p]; }
      f32x4 act, dp;
#pragma unroll
      for (int e = 0; e < 4; ++e) {      // gelu and its derivative share the erf: Phi = (1 + erf(x / sqrt 2)) / 2, gelu = x Phi, gelu' = Phi + x phi(x)
        const float xe = pre[e], cdf = 0.5f * (1.0f + erff(xe * 0.70710678118654752440f));
        act[e] = xe * cdf;
        dp[e] = da[e] * (cdf + xe * 0.39894228040143267794f * expf(-0.5f * xe * xe));
      }
#pragma unroll
      for (int p = 0; p < N; ++p) {
        gx[p] += W1[j][p] * dp;
        const f32x4 a = dp * xv[p], b = gy[p] * act;
        g[j * N + p] += (a[0] + a[1]) + (a[2] + a[3]);                  // dW1[j][p] = <dpre_j, x_p>
        g[N * N + p * N + j] += (b[0] + b[1]) + (b[2] + b[3]);          // dW2[p][j] = <dy_p, gelu(pre)_j>
      }
      g[2 * N * N + j] += (dp[0] + dp[1]) + (dp[2] + dp[3]);
    }
#pragma unroll
    for (int p = 0; p < N; ++p) {
      g[2 * N * N + N + p] += (gy[p][0] + gy[p][1]) + (gy[p][2] + gy[p][3]);
      ((f32x4*)dx + o * N * inner4 + i4)[p * inner4] = gx[p];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float v = g[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (lane == 0) part[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int i = threadIdx.x;
    const float v = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    if (ws) {
      ws[(long)blockIdx.x * NV + i] = v;
    } else {
      float* dst = i < N * N ? dW1 + i : (i < 2 * N * N ? dW2 + (i - N * N) : (i < 2 * N * N + N ? db1 + (i - 2 * N * N) : db2 + (i - 2 * N * N - N)));
      atomicAdd(dst, v);
    }
  }
}

template <int MT>
void ab_launch(const float* x, const float* dy, long outer, long inner, const float* w1, const float* b1, const float* w2, float* dx, float* dW1,
               float* db1, float* dW2, float* db2, float* ws, hipStream_t s) {
  static TantePerDevice attr;
  attr.once([&] { (void)hipFuncSetAttribute((const void*)axis_bwd_fused_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, ab_lds(MT)); });
  const long n_tiles = outer * (inner / (AB_CT * AB_NG));
  const long cap_env = tante_opt("TANTE_AXIS_BWD_WGS", 0);
  long wgs = cap_env > 0 ? cap_env : (MT >= 3 ? 512 : 1024);   // as many as are resident (LDS: 2 per CU at n = 48, 4+ below), each walking its share of the tiles
  if (wgs > n_tiles) wgs = n_tiles;
  const long slabf = 2L * (16 * MT) * (16 * MT) + 2 * 16 * MT;
  if (ws && wgs * slabf > (long)TANTE_AW_MAXWG * TANTE_AW_SLAB) wgs = (long)TANTE_AW_MAXWG * TANTE_AW_SLAB / slabf;
  if (ws && wgs > TANTE_AW_MAXWG) wgs = TANTE_AW_MAXWG;
  hipLaunchKernelGGL((axis_bwd_fused_kernel<MT>), dim3((unsigned)wgs), dim3(AB_NT), ab_lds(MT), s, x, dy, outer, inner, w1, b1, w2, dx, dW1, db1, dW2, db2, ws);
  if (ws) {
    const int ny = wgs >= 64 ? 16 : 1;
    hipLaunchKernelGGL(ab_reduce_kernel, dim3((unsigned)((slabf + 255) / 256), (unsigned)ny), dim3(256), 0, s, ws, (int)wgs, (int)slabf, 16 * MT * 16 * MT,
                       16 * MT, dW1, db1, dW2, db2);
  }
}

}  // namespace

extern "C" int tante_axis_mlp_bwd_fused_supported(int n, int64_t inner) {
  if (n == 4) return inner > 0 && inner % 4 == 0;      // the temporal axis: the fp32 vector-unit form (exact in every compute mode)
  return (n == 16 || n == 32 || n == 48) && inner > 0 && inner % (AB_CT * AB_NG) == 0;
}

template <int N>
static void ab_launch_small(const float* x, const float* dy, long outer, long inner, const float* w1, const float* b1, const float* w2, float* dx,
                            float* dW1, float* db1, float* dW2, float* db2, float* ws, hipStream_t s) {
  const long total = outer * (inner / 4);
  const long cap_env = tante_opt("TANTE_AXIS_BWD_SMALL_WGS", 0);
  long wgs = (total + 255) / 256;
  const long cap = cap_env > 0 ? cap_env : 512;      // measured at cfg3 (393 k float4 columns): 37.5 us at 512, 40.6 at 256, 45.5 at 1024, 54 at 2048
  if (wgs > cap) wgs = cap;
  const int nv = 2 * N * N + 2 * N;
  hipLaunchKernelGGL((axis_bwd_small_kernel<N>), dim3((unsigned)wgs), dim3(256), 0, s, x, dy, outer, inner / 4, w1, b1, w2, dx, dW1, db1, dW2, db2, ws);
  if (ws)
    hipLaunchKernelGGL(ab_reduce_kernel, dim3((unsigned)((nv + 255) / 256), (unsigned)(wgs >= 64 ? 16 : 1)), dim3(256), 0, s, ws, (int)wgs, nv, N * N, N,
                       dW1, db1, dW2, db2);
}

extern "C" int tante_axis_mlp_bwd_fused_ws(const float* x, const float* dy, int64_t outer, int n, int64_t inner, const float* w1, const float* b1,
                                           const float* w2, float* dx, float* dW1, float* db1, float* dW2, float* db2, void* workspace,
                                           int64_t workspace_bytes, void* stream);
extern "C" int tante_axis_mlp_bwd_fused(const float* x, const float* dy, int64_t outer, int n, int64_t inner, const float* w1, const float* b1,
                                        const float* w2, float* dx, float* dW1, float* db1, float* dW2, float* db2, void* stream) {
  return tante_axis_mlp_bwd_fused_ws(x, dy, outer, n, inner, w1, b1, w2, dx, dW1, db1, dW2, db2, nullptr, 0, stream);
}
extern "C" int tante_axis_mlp_bwd_fused_ws(const float* x, const float* dy, int64_t outer, int n, int64_t inner, const float* w1, const float* b1,
                                           const float* w2, float* dx, float* dW1, float* db1, float* dW2, float* db2, void* workspace,
                                           int64_t workspace_bytes, void* stream) {
  if (!x || !dy || !w1 || !b1 || !w2 || !dx || !dW1 || !db1 || !dW2 || !db2 || outer <= 0) TANTE_FAIL(-1, "tante_axis_mlp_bwd_fused: bad argument");
  if (!tante_axis_mlp_bwd_fused_supported(n, inner))
    TANTE_FAIL(-2, "tante_axis_mlp_bwd_fused: n = 4 with inner %% 4 == 0, or n in {16, 32, 48} with inner %% 64 == 0 (got %d, %ld)", n, (long)inner);
  if ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15)) TANTE_FAIL(-2, "tante_axis_mlp_bwd_fused: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  float* ws = (workspace && workspace_bytes >= TANTE_AW_WS_FLOATS * (int64_t)sizeof(float) && ((uintptr_t)workspace % 16) == 0) ? (float*)workspace : nullptr;
  if (n == 4) { ab_launch_small<4>(x, dy, outer, inner, w1, b1, w2, dx, dW1, db1, dW2, db2, ws, s); TANTE_CHECK_LAUNCH(); return 0; }
  switch (n / 16) {
    case 1: ab_launch<1>(x, dy, outer, inner, w1, b1, w2, dx, dW1, db1, dW2, db2, ws, s); break;
    case 2: ab_launch<2>(x, dy, outer, inner, w1, b1, w2, dx, dW1, db1, dW2, db2, ws, s); break;
    default: ab_launch<3>(x, dy, outer, inner, w1, b1, w2, dx, dW1, db1, dW2, db2, ws, s); break;
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}
