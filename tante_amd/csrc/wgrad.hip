// Weight-gradient GEMM:  dW[i][j] (+)= sum_r U[r][i] * V[r][j]      (U: R x I, V: R x J, R = tokens / patches / columns)
//
// Both operands are "row matrices" gathered like the forward GEMM's left operand (dense / strided rows, k = s patches
// of channels-last or channels-first images), so the same kernel serves every parameter on the path:
//   Linear          U = dY rows,            V = X rows                      -> dW (N, K)
//   patch-embed     U = dPre rows,          V = patches of the input        -> Conv2d weight (Cout, Cin, P, P)
//   transposed conv U = X rows (pixels),    V = patches of the output grad  -> ConvTranspose2d weight (Cin, Cout, P, P)
//   axis propagator U = dY lines,           V = hidden lines (element stride = inner)
// The contraction runs over r, which is the slow axis of both operands, so tiles are staged through LDS TRANSPOSED
// ([column][32 rows], r contiguous): one 16-byte ds_read per lane is then an MFMA operand fragment.  The result tile is
// added with fp32 atomics at the parameter's own index (TANTE_W_* layouts), r is split over workgroups.
#include "common.hip.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float ldg(const void* p, int dtype, long i) {
  return dtype == TANTE_BF16 ? __uint_as_float(((unsigned)((const unsigned short*)p)[i]) << 16) : ((const float*)p)[i];
}

// element (row r, column c) of a row matrix
__device__ __forceinline__ float rm_elem(const TanteRowMat& m, long r, int c) {
  if (m.mode == TANTE_A_LINEAR) {
    return ldg(m.p, m.dtype, (r / m.n0) * m.s1 + (r % m.n0) * m.s0 + m.off + (long)c * m.es);
  }
  const int Wo = m.Win / m.P, Ho = m.Hin / m.P;
  const long img = r / (Ho * Wo);
  const int rem = (int)(r % (Ho * Wo)), ho = rem / Wo, wo = rem % Wo;
  const long img_off = (img / m.n0) * m.s1 + (img % m.n0) * (long)m.Cin * m.Hin * m.Win + m.off;
  if (m.mode == TANTE_A_PATCH_NHWC) {  // c = (kh, kw, ci)
    const int seg = m.P * m.Cin, kh = c / seg, rest = c % seg;
    return ldg(m.p, m.dtype, img_off + ((long)(ho * m.P + kh) * m.Win + (long)wo * m.P) * m.Cin + rest);
  }
  const int pp = m.P * m.P, ci = c / pp, kh = (c % pp) / m.P, kw = c % m.P;  // c = (ci, kh, kw)
  return ldg(m.p, m.dtype, img_off + ((long)ci * m.Hin + ho * m.P + kh) * m.Win + (long)wo * m.P + kw);
}

__device__ __forceinline__ long out_index(int layout, int i, int j, int I, int J, int P, int Co, int swap) {
  // (n, k) of the parameter in pack-layout terms
  const int n = swap ? j : i, k = swap ? i : j;
  const int N = swap ? J : I, K = swap ? I : J;
  switch (layout) {
    case TANTE_W_CONV_NHWC: { const int kh = k / (P * Co), kw = (k / Co) % P, ci = k % Co; return (((long)n * Co + ci) * P + kh) * P + kw; }
    case TANTE_W_DECONV_NHWC: { const int kh = n / (P * Co), kw = (n / Co) % P, co = n % Co; return (((long)k * Co + co) * P + kh) * P + kw; }
    case TANTE_W_DECONV_NCHW: return (long)k * N + n;
    default: return (long)n * K + k;
  }
}

constexpr int RC = 64;  // rows (contraction index) per staged chunk

// staging modes of an operand
enum { ST_GENERIC = 0, ST_ROWMAJOR = 1, ST_COLMAJOR = 2, ST_PATCH_NHWC = 3, ST_PATCH_NCHW2 = 4 };

// Stage a [RC rows][T columns] chunk of a row matrix into LDS transposed ([column][row], row contiguous).
// Every thread owns 4 x 4 (row x column) blocks; the 4 values of one column over 4 consecutive rows are written
// with one 8-byte (bf16) / 16-byte (fp32) LDS store.
template <bool BF16, int T, int MODE>
struct Stager {
  static constexpr int COLG = T / 4, NBLK = COLG * (RC / 4) / 256;  // blocks per thread
  float v[NBLK][4][4];                                              // [block][col e][row q]

  __device__ __forceinline__ void blk_pos(int tid, int k, int& cg, int& rg) const {
    const int id = tid + k * 256;
    if constexpr (MODE == ST_COLMAJOR) { rg = id % (RC / 4); cg = id / (RC / 4); }   // lanes run along rows: contiguous source
    else {
      // a wave covers 8 column groups x 8 row groups: 64-byte global segments, and its 8-byte LDS stores
      // ((4cg+e) * STRIDE + 4 rg elements) then touch 32 distinct banks
      const int g = id >> 6;
      cg = (id & 7) + 8 * (g % (COLG / 8));
      rg = ((id >> 3) & 7) + 8 * (g / (COLG / 8));
    }
  }

  __device__ __forceinline__ void load(const TanteRowMat& m, long r0, long r_end, int c0, int ncols, int tid) {
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      int cg, rg;
      blk_pos(tid, k, cg, rg);
      const int c = c0 + cg * 4;
      const long r = r0 + rg * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[k][e][q] = 0.f;
      if constexpr (MODE == ST_ROWMAJOR) {
        if (c + 4 <= ncols) {
        // one (64-bit) divide per 4-row block, none for dense matrices; a block never straddles two n0-groups when n0 % 4 == 0
        const bool dense = m.s1 == 0;
        const long blk = dense ? 0 : r / m.n0, rin = dense ? r : r - blk * m.n0;
        const bool same = dense || (rin + 3 < m.n0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (r + q < r_end) {
            const long base = (same ? blk * m.s1 + (rin + q) * m.s0 : ((r + q) / m.n0) * m.s1 + ((r + q) % m.n0) * m.s0) + m.off + c;
            if (m.dtype == TANTE_BF16) {
              const u32x2 u = *(const u32x2*)((const unsigned short*)m.p + base);
              v[k][0][q] = bf16_lo(u[0]); v[k][1][q] = bf16_hi(u[0]); v[k][2][q] = bf16_lo(u[1]); v[k][3][q] = bf16_hi(u[1]);
            } else {
              const f32x4 f = *(const f32x4*)((const float*)m.p + base);
              v[k][0][q] = f[0]; v[k][1][q] = f[1]; v[k][2][q] = f[2]; v[k][3][q] = f[3];
            }
          }
        }
      } else if constexpr (MODE == ST_PATCH_NHWC) {
        // k = s patches of a channels-last image: columns (kh, kw, ci) -- 4 consecutive columns are 4 contiguous elements of the
        // run (kw, ci) of image row ho*P + kh (Cin % 4 == 0).  One index decomposition per ROW, not per element.
        if (c + 4 <= ncols) {
          const int seg = m.P * m.Cin, kh = c / seg, rest = c - kh * seg;
          const int Wo = m.Win / m.P, Ho = m.Hin / m.P;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (r + q < r_end) {
              const long rr = r + q, img = rr / ((long)Ho * Wo);
              const int rem = (int)(rr - img * Ho * Wo), ho = rem / Wo, wo = rem - ho * Wo;
              const long base = (img / m.n0) * m.s1 + (img % m.n0) * (long)m.Cin * m.Hin * m.Win + m.off +
                                ((long)(ho * m.P + kh) * m.Win + (long)wo * m.P) * m.Cin + rest;
              if (m.dtype == TANTE_BF16) {
                const u32x2 u = *(const u32x2*)((const unsigned short*)m.p + base);
                v[k][0][q] = bf16_lo(u[0]); v[k][1][q] = bf16_hi(u[0]); v[k][2][q] = bf16_lo(u[1]); v[k][3][q] = bf16_hi(u[1]);
              } else {
                const f32x4 f = *(const f32x4*)((const float*)m.p + base);
                v[k][0][q] = f[0]; v[k][1][q] = f[1]; v[k][2][q] = f[2]; v[k][3][q] = f[3];
              }
            }
        }
      } else if constexpr (MODE == ST_PATCH_NCHW2) {
        // 2 x 2 patches of a channels-first image (the first patch-embed stage reads the input frames): columns (ci, kh, kw), so a
        // 4-column group is ONE channel's patch and the block's 4 rows (wo .. wo + 3 of one image row: Wo % 4 == 0) are 8 contiguous
        // pixels of image rows 2 ho and 2 ho + 1.  One 32-bit index decomposition per block; the element-wise generic gather spent
        // ~100 integer instructions (64-bit divisions) per element and ran this stage's weight gradient at 0.6 TB/s.
        if (c + 4 <= ncols && r < r_end) {
          const unsigned Wo = (unsigned)m.Win >> 1, Ho = (unsigned)m.Hin >> 1, hw = Ho * Wo;
          const unsigned rr = (unsigned)r, img = rr / hw, rem = rr - img * hw, ho = rem / Wo, wo = rem - ho * Wo;
          const unsigned ib = img / (unsigned)m.n0, ii = img - ib * (unsigned)m.n0;
          const long base = (long)ib * m.s1 + (long)ii * m.Cin * m.Hin * m.Win + m.off +
                            ((long)(c >> 2) * m.Hin + 2 * ho) * m.Win + 2 * wo;
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            float px[8];
            if (m.dtype == TANTE_BF16) {
              const u32x4 u = *(const u32x4*)((const unsigned short*)m.p + base + (long)kh * m.Win);
#pragma unroll
              for (int i = 0; i < 4; ++i) { px[2 * i] = bf16_lo(u[i]); px[2 * i + 1] = bf16_hi(u[i]); }
            } else {
              const f32x4 f0 = *(const f32x4*)((const float*)m.p + base + (long)kh * m.Win), f1 = *(const f32x4*)((const float*)m.p + base + (long)kh * m.Win + 4);
#pragma unroll
              for (int i = 0; i < 4; ++i) { px[i] = f0[i]; px[4 + i] = f1[i]; }
            }
#pragma unroll
            for (int kw = 0; kw < 2; ++kw)
#pragma unroll
              for (int q = 0; q < 4; ++q) v[k][2 * kh + kw][q] = (r + q < r_end) ? px[2 * q + kw] : 0.f;
          }
        }
      } else if constexpr (MODE == ST_COLMAJOR) {   // 4 consecutive rows of one column are contiguous (fp32 source)
        if (r + 4 <= r_end) {
          const long base = (r / m.n0) * m.s1 + (r % m.n0) * m.s0 + m.off;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < ncols) {
              const f32x4 f = *(const f32x4*)((const float*)m.p + base + (long)(c + e) * m.es);
              v[k][e][0] = f[0]; v[k][e][1] = f[1]; v[k][e][2] = f[2]; v[k][e][3] = f[3];
            }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (r + q < r_end && c + e < ncols) v[k][e][q] = ((const float*)m.p)[((r + q) / m.n0) * m.s1 + ((r + q) % m.n0) * m.s0 + m.off + (long)(c + e) * m.es];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (r + q < r_end && c + e < ncols) v[k][e][q] = rm_elem(m, r + q, c + e);
      }
    }
  }

  template <class E>
  __device__ __forceinline__ void store(E* lds, int stride, int tid) const {
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
      int cg, rg;
      blk_pos(tid, k, cg, rg);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        E* d = lds + (cg * 4 + e) * stride + rg * 4;
        if constexpr (BF16) {
          u32x2 w;
          w[0] = pack_bf16x2(v[k][e][0], v[k][e][1]);
          w[1] = pack_bf16x2(v[k][e][2], v[k][e][3]);
          *(u32x2*)d = w;
        } else {
          *(f32x4*)d = f32x4{v[k][e][0], v[k][e][1], v[k][e][2], v[k][e][3]};
        }
      }
    }
  }
};

// T x T output tile; wave (wi, wj) of the 2 x 2 wave grid owns a (T/2) x (T/2) part.  The staging modes are template
// parameters: inlining the generic gather (integer divisions) at all 64 element sites made a 125k-instruction kernel.
template <bool BF16, int T, int MU, int MV>
__global__ __launch_bounds__(256) void wgrad_kernel(const TanteRowMat U, const TanteRowMat V, long R, int I, int J, long rows_per_split,
                                                    float* __restrict__ dW, float* __restrict__ dbias, int layout, int P, int Co, int swap,
                                                    float* __restrict__ slab = nullptr, float* __restrict__ bias_slab = nullptr) {
  using elem_t = typename std::conditional<BF16, unsigned short, float>::type;
  constexpr int STRIDE = RC + (BF16 ? 8 : 4);  // elements per LDS row (one operand column, RC rows + pad; multiple of 16 bytes)
  constexpr int NT = T / 32;                   // 16 x 16 MFMA tiles per wave per side
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  elem_t* Ut = (elem_t*)smem_raw;
  elem_t* Vt = Ut + T * STRIDE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  const int i0 = blockIdx.x * T, j0 = blockIdx.y * T;
  const long r_begin = (long)blockIdx.z * rows_per_split, r_end = min(R, r_begin + rows_per_split);
  const int wi = wave >> 1, wj = wave & 1;
  f32x4 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;  // bias gradient: column sums of U (threads 0..T-1 of the j-tile-0 workgroups)
  const bool do_bias = dbias != nullptr && blockIdx.y == 0;
  Stager<BF16, T, MU> su;
  Stager<BF16, T, MV> sv;
  su.load(U, r_begin, r_end, i0, I, tid);
  sv.load(V, r_begin, r_end, j0, J, tid);
  for (long r0 = r_begin; r0 < r_end; r0 += RC) {
    __syncthreads();  // the previous chunk's fragments have been consumed
    su.store(Ut, STRIDE, tid);
    sv.store(Vt, STRIDE, tid);
    __syncthreads();
    if (r0 + RC < r_end) {  // next chunk's global loads fly under the MFMAs
      su.load(U, r0 + RC, r_end, i0, I, tid);
      sv.load(V, r0 + RC, r_end, j0, J, tid);
    }
    if (do_bias && tid < T) {
      float sacc = 0.f;
#pragma unroll
      for (int q = 0; q < RC; ++q) {
        if constexpr (BF16) sacc += __uint_as_float(((unsigned)Ut[tid * STRIDE + q]) << 16);
        else sacc += Ut[tid * STRIDE + q];
      }
      bsum += sacc;
    }
#pragma unroll
    for (int ks = 0; ks < (BF16 ? 2 : 4); ++ks) {
      u32x4 af[NT], bf[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a)
        af[a] = *(const u32x4*)(Ut + (wi * (T / 2) + a * 16 + l15) * STRIDE + ks * (BF16 ? 32 : 16) + kk * (BF16 ? 8 : 4));
#pragma unroll
      for (int b = 0; b < NT; ++b)
        bf[b] = *(const u32x4*)(Vt + (wj * (T / 2) + b * 16 + l15) * STRIDE + ks * (BF16 ? 32 : 16) + kk * (BF16 ? 8 : 4));
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          if constexpr (BF16) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]), __builtin_bit_cast(bf16x8, bf[b]), acc[a][b], 0, 0, 0);
          } else {  // lane chunk = rows 16*ks + 4*kk .. +3, the same order for both operands
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[a][s]), __uint_as_float(bf[b][s]), acc[a][b], 0, 0, 0);
          }
        }
    }
  }
  // D[row = i][col = j]: lane holds j = l15, i = 4*kk + reg
  if (slab) {   // single-tile outputs with a workspace: the split's partial (valid entries only, [i][j]) for wgrad_gen_reduce_kernel
    float* mine = slab + (long)blockIdx.z * I * J;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int i = wi * (T / 2) + a * 16 + kk * 4 + rg, j = wj * (T / 2) + b * 16 + l15;
          if (i < I && j < J) mine[i * J + j] = acc[a][b][rg];
        }
    if (dbias && tid < T && tid < I) bias_slab[(long)blockIdx.z * I + tid] = bsum;
    return;
  }
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int i = i0 + wi * (T / 2) + a * 16 + kk * 4 + rg, j = j0 + wj * (T / 2) + b * 16 + l15;
        if (i < I && j < J) atomicAdd(&dW[out_index(layout, i, j, I, J, P, Co, swap)], acc[a][b][rg]);
      }
  if (do_bias && tid < T && i0 + tid < I) atomicAdd(&dbias[i0 + tid], bsum);
}

// second stage of the single-tile form: entry e of the partials ([i][j], then the bias partials), summed over a chunk of the splits
__global__ __launch_bounds__(256) void wgrad_gen_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias_slab, int n_split, int I, int J,
                                                               float* __restrict__ dW, float* __restrict__ dbias, int layout, int P, int Co, int swap) {
  const int e = blockIdx.x * 256 + threadIdx.x, ne = I * J + (dbias ? I : 0);
  if (e >= ne) return;
  const int per = (n_split + gridDim.y - 1) / gridDim.y, s0 = blockIdx.y * per, s1 = min(n_split, s0 + per);
  if (s0 >= s1) return;
  const bool isb = e >= I * J;
  const long st = isb ? I : (long)I * J;
  const float* p = (isb ? bias_slab + (e - I * J) : slab + e) + (long)s0 * st;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = s0;
  for (; s + 8 <= s1; s += 8, p += 8 * st) {
    const float v0 = p[0], v1 = p[st], v2 = p[2 * st], v3 = p[3 * st], v4 = p[4 * st], v5 = p[5 * st], v6 = p[6 * st], v7 = p[7 * st];
    a0 += v0 + v4; a1 += v1 + v5; a2 += v2 + v6; a3 += v3 + v7;
  }
  for (; s < s1; ++s, p += st) a0 += p[0];
  const float sum = (a0 + a1) + (a2 + a3);
  if (isb) atomicAdd(&dbias[e - I * J], sum);
  else atomicAdd(&dW[out_index(layout, e / J, e % J, I, J, P, Co, swap)], sum);
}

template <bool BF16, int T>
void launch_wgrad(const TanteRowMat& U, const TanteRowMat& V, long R, int I, int J, float* dW, float* dbias, int layout, int P, int Co, int swap,
                  int mu, int mv, hipStream_t s, float* ws = nullptr, int64_t ws_bytes = 0) {
  const int ti = (I + T - 1) / T, tj = (J + T - 1) / T;
  const int wg_target = tante_opt("TANTE_WGRAD_WGS", 512);
  // a single output tile with a workspace (the skinny weight gradients of the convolution stages: 98 k rows into 64 x 16 values): the
  // kernel is latency-bound -- one staged chunk in flight per workgroup -- and 384 workgroups of 4 chunks each left the chip 1.5
  // workgroups per CU (0.7 TB/s); with the partials stored and summed by a second kernel (no 384-deep same-address atomics) the row
  // range is cut into up to 1 536 pieces of at least one chunk
  const int slab_target = tante_opt("TANTE_WGRAD_SLAB_WGS", 1536);
  const bool slab_ok = ws != nullptr && ti == 1 && tj == 1 && slab_target > 0 && R >= 8 * RC;
  long split = (slab_ok ? slab_target : wg_target) / ((long)ti * tj);   // two workgroups per CU; fewer splits = less atomic traffic
  if (split < 1) split = 1;
  const long max_split = slab_ok ? (R + RC - 1) / RC : (R + 4 * RC - 1) / (4 * RC);
  if (split > max_split) split = max_split;
  if (slab_ok && split * ((long)I * J + I) * 4 > ws_bytes) split = ws_bytes / (((long)I * J + I) * 4);
  if (split > 65535) split = 65535;
  long per = (R + split - 1) / split;
  per = (per + RC - 1) / RC * RC;
  split = (R + per - 1) / per;
  const size_t lds = 2 * (size_t)T * (RC + (BF16 ? 8 : 4)) * (BF16 ? 2 : 4);
  const dim3 grid(ti, tj, (unsigned)split);
  const bool use_slab = slab_ok && split > 1;
  float* slab = use_slab ? ws : nullptr;
  float* bias_slab = use_slab ? ws + split * (long)I * J : nullptr;
#define TANTE_WG(MUV, MVV) hipLaunchKernelGGL((wgrad_kernel<BF16, T, MUV, MVV>), grid, dim3(256), lds, s, U, V, R, I, J, per, dW, dbias, layout, P, Co, swap, slab, bias_slab)
  if (mu == ST_ROWMAJOR && mv == ST_ROWMAJOR) TANTE_WG(ST_ROWMAJOR, ST_ROWMAJOR);        // linear layers
  else if (mu == ST_COLMAJOR && mv == ST_COLMAJOR) TANTE_WG(ST_COLMAJOR, ST_COLMAJOR);   // axis propagators
  else if (mu == ST_ROWMAJOR && mv == ST_PATCH_NHWC) TANTE_WG(ST_ROWMAJOR, ST_PATCH_NHWC);   // conv / transposed-conv stages on channels-last images
  else if (mu == ST_ROWMAJOR && mv == ST_PATCH_NCHW2) TANTE_WG(ST_ROWMAJOR, ST_PATCH_NCHW2);   // first patch-embed stage: 2 x 2 patches of channels-first frames
  else if (mu == ST_ROWMAJOR) TANTE_WG(ST_ROWMAJOR, ST_GENERIC);                         // V = patches of the channels-first input / output
  else TANTE_WG(ST_GENERIC, ST_GENERIC);
#undef TANTE_WG
  if (use_slab) {
    const int ne = I * J + (dbias ? I : 0);
    hipLaunchKernelGGL(wgrad_gen_reduce_kernel, dim3((unsigned)((ne + 255) / 256), (unsigned)(split >= 64 ? 16 : 1)), dim3(256), 0, s, slab, bias_slab,
                       (int)split, I, J, dW, dbias, layout, P, Co, swap);
  }
}

// ---- fast path: both operands dense row-major bf16, I and J multiples of 128, R a multiple of 32 -------------------------------
// The generic kernel above is LATENCY-bound: one chunk of global -> register -> (transposing) LDS staging in flight per
// workgroup.  Here the rows go HBM -> LDS by LDS-DMA through a 4-deep ring of 32-row chunks with no register round trip, stored
// row-major; the contraction index r is the slow axis of both operands, so the MFMA fragments (8 consecutive r of one column)
// come out of `ds_read_b64_tr_b16`, gfx950's transposing LDS read (4 rows x 16 columns per 16-lane group, delivered column-major).
// The image is the XOR-swizzled plain-row layout of cdna_hip_programming.md T10 (b): chunk' = chunk ^ (((row & 3) << 2) |
// ((row >> 2) & 3)); LDS-DMA writes lane-linear, so the swizzle is applied to the global SOURCE chunk each lane fetches.
// Bias gradient: one extra MFMA per A tile against an all-ones B fragment.
constexpr int WT = 128, WRC = 32;
constexpr int WCHUNK = WRC * WT * 2;   // bytes of one operand chunk

__device__ __forceinline__ int wg_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ u32x2 ds_read_tr16_b64(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}

// up to WSEG (U, V) operand pairs of R_seg rows each, contracted as ONE row range of n_seg * R_seg rows: the BPTT uses of a weight share
// one launch and one atomic epilogue (tante_wgrad_multi).  A split never straddles two segments (rows_per_split divides R_seg).
constexpr int WSEG = 8;
struct WgSegs {
  const unsigned short* U[WSEG];
  const unsigned short* V[WSEG];
  long R_seg;
};

// RG = 2 (experiment, off by default -- see wgrad_tr_launch): eight waves, two ROW GROUPS of four (each with its own ring) that split the
// workgroup's row range in halves and add their partial tiles through LDS before anything leaves the CU: HALF the partial-tile traffic
// (a partial tile per row split is 64 KiB out, 64 KiB back into the reduce kernel: at 128 splits of a 256 x 256 weight that is
// 2 x 33 MB beside 100 MB of operands).
template <int WNBUF, int RG>   // ring depth: 4 (64 KB per row group) or 8 (128 KB: a lone 4-wave workgroup keeps 7 chunks = 112 KB in flight)
__device__ __forceinline__ void wgrad_tr_body(const WgSegs& SG, long ldu, long ldv, long R, int I, int J, long rows_per_split, float* __restrict__ dW,
                                              float* __restrict__ dbias, int layout, int P, int Co, int swap, int debug, int n_split,
                                              float* __restrict__ slab, float* __restrict__ bias_slab, const unsigned bid, const int seg_splits = 0,
                                              const int xcd_rot = 0) {
  extern __shared__ __attribute__((aligned(16))) char wsm[];   // ring: [buf][U chunk | V chunk]
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, grp = tid >> 8, kk = lane >> 4, l15 = lane & 15;
  // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs by linear id, and every output tile of one row range re-reads
  // that range's operands, so the tiles of a split must share an XCD (one L2) -- with a (tile, tile, split) grid they land on different
  // ones and every tile pulls its operand rows from HBM again.  id -> (xcd = id % 8, slot = id / 8); the slots of an XCD walk the tiles
  // of split 8 * (slot / ntile) + xcd.
  const int ti = I / WT, ntile = ti * (J / WT);
  const int xcd = bid & 7, slot = bid >> 3;
  // (xcd_rot: the jobs launch starts each job's splits on the XCD where the previous job's ended, so that split counts that are not
  // multiples of 8 still load the eight XCDs evenly)
  const int bz = (slot / ntile) * 8 + ((xcd - xcd_rot) & 7), tile = slot % ntile;
  if (bz >= n_split) return;
  const int i0 = (tile % ti) * WT, j0 = (tile / ti) * WT;
  const bool first_j = (tile / ti) == 0;
  // seg_splits > 0 (the jobs launch): every segment is cut into seg_splits ranges of whole chunks whose lengths differ by at most one
  // chunk -- any split count fits any segment, so the grid can be sized to the chip instead of to the divisors of R_seg
  int seg;
  long r_begin, r_end;
  if (seg_splits > 0) {
    seg = bz / seg_splits;
    const long sp = bz - seg * seg_splits, nch_seg = SG.R_seg / WRC;
    r_begin = (sp * nch_seg / seg_splits) * WRC;
    r_end = ((sp + 1) * nch_seg / seg_splits) * WRC;
  } else {
    const long g_begin = (long)bz * rows_per_split;                  // row index in the concatenation of the segments
    seg = (int)(g_begin / SG.R_seg);
    r_begin = g_begin - (long)seg * SG.R_seg + (long)grp * (rows_per_split / RG);      // rows_per_split is a multiple of RG * WRC
    r_end = min(SG.R_seg, r_begin + rows_per_split / RG);
  }
  char* const ring = wsm + grp * (WNBUF * 2 * WCHUNK);
  const unsigned short* __restrict__ U = SG.U[seg];
  const unsigned short* __restrict__ V = SG.V[seg];
  const int nchunk = (debug & 2) ? 0 : (int)((r_end - r_begin) / WRC);
  const int wi = wave >> 1, wj = wave & 1;
  // DMA: a wave instruction moves 4 rows (lanes 16 q .. 16 q + 15 = row q); wave w copies rows 8 w .. 8 w + 7 of the chunk
  auto issue = [&](int c) {
    char* buf = ring + (c % WNBUF) * (2 * WCHUNK);
    const long r0 = r_begin + (long)c * WRC;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = wave * 8 + h * 4 + (lane >> 4);
      const int ch = (lane & 15) ^ wg_swz(row);                 // the logical chunk that belongs at this lane's LDS slot
      const unsigned short* gu = U + (r0 + row) * ldu + i0 + ch * 8;
      const unsigned short* gv = V + (r0 + row) * ldv + j0 + ch * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gu,
                                       (__attribute__((address_space(3))) void*)(buf + (wave * 8 + h * 4) * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gv,
                                       (__attribute__((address_space(3))) void*)(buf + WCHUNK + (wave * 8 + h * 4) * 256), 16, 0, 0);
    }
  };
  f32x4 acc[4][4], bacc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    bacc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = dbias != nullptr && first_j && wj == 0;
  const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};   // bf16 1.0 pairs
  // transposed-read addressing: lane 4 q + p of a 16-lane group supplies row (rb + q), columns 4 p .. 4 p + 3 of the 16-column tile
  const int q = l15 >> 2, p = l15 & 3;
  unsigned trU[4][2], trV[4][2];   // byte offsets inside an operand chunk, per tile and per half (rows 8 kk + 4 h + q)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * kk + 4 * h + q;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int cu = (wi * 64 + a * 16) / 8 + (p >> 1), cv = (wj * 64 + a * 16) / 8 + (p >> 1);
      trU[a][h] = 256 * row + 16 * (cu ^ wg_swz(row)) + 8 * (p & 1);
      trV[a][h] = 256 * row + 16 * (cv ^ wg_swz(row)) + 8 * (p & 1);
    }
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)ring;

  // Software pipeline: the fragments of chunk c + 1 are read from LDS (into the other register set) while the MFMAs of chunk c run --
  // read -> wait -> MFMA in series left the matrix pipe idle for the LDS latency of every chunk, and with 2 waves per SIMD there is
  // nobody to fill it.  Per iteration: chunk c + 1 landed + my reads of chunk c done -> barrier (everybody's are: buffer c is free)
  // -> DMA of chunk c + WNBUF into it -> LDS reads of chunk c + 1 -> MFMAs of chunk c.
  auto wait_landed = [&](int later) {          // all but the `later` most recently issued chunks (4 DMA instructions per lane and chunk)
    if (later >= 7) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    else if (later == 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (later == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (later == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (later == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  auto load_frags = [&](int c, u32x4 (&af)[4], u32x4 (&bf)[4]) {
    const unsigned ub = lds0 + (c % WNBUF) * (2 * WCHUNK), vb = ub + WCHUNK;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const u32x2 lo = ds_read_tr16_b64(ub + trU[a][0]), hi = ds_read_tr16_b64(ub + trU[a][1]);
      af[a] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      const u32x2 lo2 = ds_read_tr16_b64(vb + trV[a][0]), hi2 = ds_read_tr16_b64(vb + trV[a][1]);
      bf[a] = u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
    }
  };
  auto step = [&](int c, u32x4 (&af)[4], u32x4 (&bf)[4], u32x4 (&naf)[4], u32x4 (&nbf)[4]) {
    if (c + 1 < nchunk) wait_landed(min(nchunk - 2 - c, WNBUF - 2));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // fragments of chunk c are in registers
    __syncthreads();                                             // chunk c + 1 complete for every wave; buffer c free
    if (c + WNBUF < nchunk) issue(c + WNBUF);
    if (c + 1 < nchunk) load_frags(c + 1, naf, nbf);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]), __builtin_bit_cast(bf16x8, bf[b]), acc[a][b], 0, 0, 0);
      if (do_bias) bacc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[a]), __builtin_bit_cast(bf16x8, ones), bacc[a], 0, 0, 0);
    }
  };
#pragma unroll
  for (int c = 0; c < WNBUF; ++c)
    if (c < nchunk) issue(c);
  u32x4 afA[4], bfA[4], afB[4], bfB[4];
  if (nchunk > 0) {
    wait_landed(min(nchunk - 1, WNBUF - 1));
    __syncthreads();
    load_frags(0, afA, bfA);
  }
  for (int c = 0; c < nchunk; c += 2) {
    step(c, afA, bfA, afB, bfB);
    if (c + 1 < nchunk) step(c + 1, afB, bfB, afA, bfA);
  }
  if (RG == 2) {
    // the second row group hands its partial tile over through LDS (the rings are idle now): [wave][a][b][lane] 16-byte pieces
    __syncthreads();
    f32x4* const xch = (f32x4*)wsm + (wave * 16) * 64 + lane;
    if (grp == 1) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) xch[(a * 4 + b) * 64] = acc[a][b];
      if (do_bias) {
#pragma unroll
        for (int a = 0; a < 4; ++a) ((f32x4*)wsm)[4 * 16 * 64 + (wave * 4 + a) * 64 + lane] = bacc[a];
      }
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += xch[(a * 4 + b) * 64];
    if (do_bias) {
#pragma unroll
      for (int a = 0; a < 4; ++a) bacc[a] += ((f32x4*)wsm)[4 * 16 * 64 + (wave * 4 + a) * 64 + lane];
    }
  }
  if (slab) {
    // two-stage reduction: the partial tile goes out with plain 16-byte stores in REGISTER order ([wave][a][b][lane][4], 1 KiB per wave
    // instruction) and wgrad_reduce_kernel sums the splits -- fp32 atomics run at ~1.3 TB/s chip-wide, a fifth of the store rate, and
    // 64 KiB of them per workgroup capped this kernel at 128 workgroups (half the CUs, one per CU)
    float* t = slab + ((long)bz * ntile + tile) * (WT * WT) + (wave * 16) * 256 + lane * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) *(f32x4*)(t + (a * 4 + b) * 256) = acc[a][b];
    if (do_bias && l15 == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a) *(f32x4*)(bias_slab + (long)bz * I + i0 + wi * 64 + a * 16 + kk * 4) = bacc[a];
    }
    return;
  }
  // D[row = i][col = j]: lane holds j = l15, i = 4*kk + reg
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int i = i0 + wi * 64 + a * 16 + kk * 4 + rg, j = j0 + wj * 64 + b * 16 + l15;
        if ((debug & 1) && acc[a][b][rg] != 12345.f) continue;
        atomicAdd(&dW[out_index(layout, i, j, I, J, P, Co, swap)], acc[a][b][rg]);
      }
  if (do_bias && l15 == 0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) atomicAdd(&dbias[i0 + wi * 64 + a * 16 + kk * 4 + rg], bacc[a][rg]);
  }
}

template <int WNBUF, int RG>
__global__ __launch_bounds__(256 * RG, (WNBUF == 4 && RG == 1) ? 2 : 1) void wgrad_tr_kernel(const WgSegs SG, long ldu,
                                                          long ldv, long R, int I, int J, long rows_per_split, float* __restrict__ dW,
                                                          float* __restrict__ dbias, int layout, int P, int Co, int swap, int debug, int n_split,
                                                          float* __restrict__ slab, float* __restrict__ bias_slab) {
  wgrad_tr_body<WNBUF, RG>(SG, ldu, ldv, R, I, J, rows_per_split, dW, dbias, layout, P, Co, swap, debug, n_split, slab, bias_slab, blockIdx.x);
}

// ---- several weights in ONE launch (tante_wgrad_jobs_ws) ------------------------------------------------------------------------------------
// A launch wants ~512 workgroups whatever the weight; alone, a 256 x 256 weight gets them by cutting its rows into 128 splits, and every
// split costs a 64 KiB partial tile out and back (2 x 33 MB beside 100 MB of operands) plus the launch's own ramp, tail and reduce pass --
// about half of the 45 + 8 us per weight.  The four weights of a block are gradients over the SAME number of rows at the same time: as
// jobs of one launch they share the 512 workgroups (a quarter of the splits each, a quarter of the partial traffic), one ramp and one
// reduce launch.  Every job keeps its own XCD-aware (split, tile) numbering from a workgroup id that starts at a multiple of 8.
constexpr int WJOBS = 4;
struct WgJob {
  WgSegs SG;
  long ldu, ldv, R, per;
  float *dW, *dbias, *slab, *bias_slab;
  int I, J, layout, P, Co, swap, n_split, wg_begin, seg_splits, xcd_rot;
};
struct WgJobs {
  WgJob j[WJOBS];
  int n;
};
#ifndef WGJ_NBUF
#define WGJ_NBUF 4      // ring depth of the jobs kernel (x 16 KB) ...
#define WGJ_OCC 2       // ... and the workgroups per CU it is sized for (192 VGPRs; 3 / 3 and 2 / 4 spill: train step 10.8 / 18.4 against 8.58 ms)
#endif
__global__ __launch_bounds__(256, WGJ_OCC) void wgrad_tr_jobs_kernel(const WgJobs JB) {
  int k = 0;
#pragma unroll
  for (int q = 1; q < WJOBS; ++q)
    if (q < JB.n && blockIdx.x >= (unsigned)JB.j[q].wg_begin) k = q;
  const WgJob& jb = JB.j[k];
  wgrad_tr_body<WGJ_NBUF, 1>(jb.SG, jb.ldu, jb.ldv, jb.R, jb.I, jb.J, jb.per, jb.dW, jb.dbias, jb.layout, jb.P, jb.Co, jb.swap, 0, jb.n_split, jb.slab,
                      jb.bias_slab, blockIdx.x - (unsigned)jb.wg_begin, jb.seg_splits, jb.xcd_rot);
}

// second stage: dW[i][j] += sum over the splits of a slab chunk (grid.y chunks of splits; one thread per 16-byte piece of a tile in the
// register order wgrad_tr_kernel stored it in, so consecutive threads read consecutive 16 bytes); dbias likewise from the bias slab
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ slab, const float* __restrict__ bias_slab, int n_split,
                                                  int ntile, int ti, int I, int J, float* __restrict__ dW, float* __restrict__ dbias,
                                                  int layout, int P, int Co, int swap) {
  const int per = (n_split + gridDim.y - 1) / gridDim.y, s0 = blockIdx.y * per, s1 = min(n_split, s0 + per);
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < ntile * 4096 && s0 < s1) {
    const int tile = t >> 12, rem = t & 4095, wave = rem >> 10, ab = (rem >> 6) & 15, lane = rem & 63;
    const float* p = slab + ((long)s0 * ntile + tile) * (WT * WT) + rem * 4;
    const long st = (long)ntile * (WT * WT);
    f32x4 s_a = f32x4{0.f, 0.f, 0.f, 0.f}, s_b = s_a, s_c = s_a, s_d = s_a;
    int s = s0;
    for (; s + 8 <= s1; s += 8, p += 8 * st) {      // eight loads in flight per lane: the pass is latency-bound otherwise
      const f32x4 v0 = *(const f32x4*)p, v1 = *(const f32x4*)(p + st), v2 = *(const f32x4*)(p + 2 * st), v3 = *(const f32x4*)(p + 3 * st);
      const f32x4 v4 = *(const f32x4*)(p + 4 * st), v5 = *(const f32x4*)(p + 5 * st), v6 = *(const f32x4*)(p + 6 * st), v7 = *(const f32x4*)(p + 7 * st);
      s_a += v0; s_b += v1; s_c += v2; s_d += v3;
      s_a += v4; s_b += v5; s_c += v6; s_d += v7;
    }
    for (; s + 4 <= s1; s += 4, p += 4 * st) {
      const f32x4 v0 = *(const f32x4*)p, v1 = *(const f32x4*)(p + st), v2 = *(const f32x4*)(p + 2 * st), v3 = *(const f32x4*)(p + 3 * st);
      s_a += v0; s_b += v1; s_c += v2; s_d += v3;
    }
    for (; s < s1; ++s, p += st) s_a += *(const f32x4*)p;
    const f32x4 sum = (s_a + s_b) + (s_c + s_d);
    const int i0 = (tile % ti) * WT, j0 = (tile / ti) * WT, wi = wave >> 1, wj = wave & 1, a = ab >> 2, b = ab & 3, kk = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      atomicAdd(&dW[out_index(layout, i0 + wi * 64 + a * 16 + kk * 4 + rg, j0 + wj * 64 + b * 16 + l15, I, J, P, Co, swap)], sum[rg]);
  }
  if (dbias && t < I && s0 < s1) {   // every chunk sums its own splits, eight loads in flight: one serial loop over all the splits took
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, b4 = 0.f, b5 = 0.f, b6 = 0.f, b7 = 0.f;   // as long as the whole tile reduction
    const float* q = bias_slab + (long)s0 * I + t;
    int s = s0;
    for (; s + 8 <= s1; s += 8, q += 8L * I) {
      b0 += q[0]; b1 += q[(long)I]; b2 += q[2L * I]; b3 += q[3L * I]; b4 += q[4L * I]; b5 += q[5L * I]; b6 += q[6L * I]; b7 += q[7L * I];
    }
    for (; s < s1; ++s, q += I) b0 += q[0];
    atomicAdd(&dbias[t], ((b0 + b1) + (b2 + b3)) + ((b4 + b5) + (b6 + b7)));
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias_slab, int n_split,
                                                          int ntile, int ti, int I, int J, float* __restrict__ dW, float* __restrict__ dbias,
                                                          int layout, int P, int Co, int swap) {
  wgrad_reduce_body(slab, bias_slab, n_split, ntile, ti, I, J, dW, dbias, layout, P, Co, swap);
}
struct RdJob {
  const float *slab, *bias_slab;
  float *dW, *dbias;
  int n_split, ntile, ti, I, J, layout, P, Co, swap;
};
struct RdJobs {
  RdJob j[WJOBS];
};
__global__ __launch_bounds__(256) void wgrad_reduce_jobs_kernel(const RdJobs JB) {      // grid.z = the job
  const RdJob& jb = JB.j[blockIdx.z];
  if ((int)blockIdx.x >= jb.ntile * 16) return;
  wgrad_reduce_body(jb.slab, jb.bias_slab, jb.n_split, jb.ntile, jb.ti, jb.I, jb.J, jb.dW, jb.dbias, jb.layout, jb.P, jb.Co, jb.swap);
}

static bool wgrad_tr_ok(const TanteRowMat& m, long R, int ncols) {
  return m.mode == TANTE_A_LINEAR && m.dtype == TANTE_BF16 && m.es == 1 && (m.s1 == 0 || m.n0 >= R) && m.s0 % 8 == 0 && m.off % 8 == 0 &&
         ((uintptr_t)m.p % 16) == 0 && ncols % WT == 0 && m.s0 >= ncols;
}

}  // namespace

static int check_rowmat(const TanteRowMat& m, const char* name) {
  if (!m.p) TANTE_FAIL(-1, "tante_wgrad: %s is null", name);
  if (m.mode == TANTE_A_LINEAR) {
    if (m.n0 <= 0 || m.es <= 0) TANTE_FAIL(-1, "tante_wgrad: %s needs n0 > 0 and es > 0", name);
  } else if (m.mode == TANTE_A_PATCH_NHWC || m.mode == TANTE_A_PATCH_NCHW) {
    if (m.P <= 0 || m.Cin <= 0 || m.Hin % m.P || m.Win % m.P || m.n0 <= 0) TANTE_FAIL(-1, "tante_wgrad: %s has a bad patch geometry", name);
  } else {
    TANTE_FAIL(-1, "tante_wgrad: %s has a bad mode", name);
  }
  return 0;
}

static int rm_stage_mode(const TanteRowMat& m, int ncols) {
  if (m.mode == TANTE_A_PATCH_NHWC && ((uintptr_t)m.p % 16) == 0 && m.Cin % 4 == 0 && m.off % 4 == 0 && m.s1 % 4 == 0 && ncols % 4 == 0)
    return ST_PATCH_NHWC;
  // 2 x 2 patches of a channels-first image: 8 contiguous pixels per block and image row, 16-byte aligned (bf16) / 32-byte (fp32) runs
  if (m.mode == TANTE_A_PATCH_NCHW && m.P == 2 && ((uintptr_t)m.p % 16) == 0 && (m.Win / 2) % 4 == 0 && m.Win % 8 == 0 && m.off % 8 == 0 &&
      m.s1 % 8 == 0 && ((long)m.Hin * m.Win) % 8 == 0 && ncols % 4 == 0)
    return ST_PATCH_NCHW2;
  if (m.mode != TANTE_A_LINEAR || ((uintptr_t)m.p % 16)) return ST_GENERIC;
  const int al = m.dtype == TANTE_BF16 ? 4 : 4;
  if (m.es == 1 && m.s1 % al == 0 && m.s0 % al == 0 && m.off % al == 0 && ncols % 4 == 0) return ST_ROWMAJOR;
  // lines: consecutive rows are consecutive addresses (s0 == 1) inside blocks of n0 rows
  if (m.dtype == TANTE_F32 && m.s0 == 1 && m.es % 4 == 0 && m.n0 % 4 == 0 && m.s1 % 4 == 0 && m.off % 4 == 0) return ST_COLMAJOR;
  return ST_GENERIC;
}

// the LDS-DMA / transposed-read path for n_seg operand pairs of R rows each (dense bf16 rows, I and J multiples of 128, R % 32 == 0)
static bool wgrad_tr_launch(const TanteRowMat* U, const TanteRowMat* V, int n_seg, long R, int I, int J, float* dW, float* dbias, int layout,
                            int P, int C_other, int swap, hipStream_t s, void* ws = nullptr, int64_t ws_bytes = 0) {
  const bool no_tr = tante_opt("TANTE_WGRAD_NO_TR", 0) != 0;
  if (no_tr || n_seg < 1 || n_seg > WSEG || R % WRC) return false;
  for (int g = 0; g < n_seg; ++g)
    if (!wgrad_tr_ok(U[g], R, I) || !wgrad_tr_ok(V[g], R, J) || U[g].s0 != U[0].s0 || V[g].s0 != V[0].s0) return false;
  const int ti = I / WT, tj = J / WT;
  // every workgroup ends with 128 x 128 fp32 atomics: with the main loop at memory speed the split count is a trade between
  // parallelism and atomic traffic (64 KiB per workgroup) -- measured best near 128 workgroups for <= 4 tiles, 256 otherwise
  const int wg_env = tante_opt("TANTE_WGRAD_TR_WGS", tante_opt("TANTE_WGRAD_WGS", 0));
  // with a workspace the partial tiles are stored and summed by a second kernel instead of added atomically: the atomic traffic no
  // longer limits the split count, so the grid fills the chip twice (two 64 KB workgroups per CU)
  const bool no_slab = tante_opt("TANTE_WGRAD_NO_SLAB", 0) != 0;
  const bool want_slab = ws != nullptr && !no_slab;
  // TANTE_WGRAD_RG=2 (experiment): eight-wave workgroups of two row groups, one per CU, which halves the partial-tile traffic -- and
  // LOSES to two independent 4-wave workgroups per CU (tools/wgrad_multi_time.py, 4 x 24 576 rows: 44.5 vs 40.1 us for 256 x 256,
  // 105 vs 90 us for 768 x 256): one barrier per chunk over eight waves keeps the two halves in step, two workgroups drift apart
  const int rg_env = tante_opt("TANTE_WGRAD_RG", 1);
  const int deep_env = tante_opt("TANTE_WGRAD_DEEP", -1);
  const int RGv = (want_slab && rg_env == 2 && deep_env <= 0) ? 2 : 1;
  const int wg_target = wg_env > 0 ? wg_env : (want_slab ? 512 / RGv : (ti * tj <= 4 ? 128 : 256));
  // splits PER SEGMENT (a split never straddles two segments): the workgroup target is shared by the segments
  long split = wg_target / ((long)ti * tj * n_seg);
  if (split < 1) split = 1;
  const long nch = R / WRC;
  if (split > nch / (4 * RGv)) split = nch / (4 * RGv) > 0 ? nch / (4 * RGv) : 1;
  long per = ((nch + split - 1) / split) * WRC;
  if (per % (RGv * WRC)) per += WRC;
  while (R % per && per < R) per += RGv * WRC;      // rows_per_split must divide the segment (and be whole chunks per row group)
  if (R % per || per % (RGv * WRC)) return false;
  split = R / per;
  const long total = split * n_seg;
  if (total > 65535) return false;
  WgSegs SG;
  for (int g = 0; g < WSEG; ++g) {
    SG.U[g] = (const unsigned short*)U[g < n_seg ? g : 0].p + U[g < n_seg ? g : 0].off;
    SG.V[g] = (const unsigned short*)V[g < n_seg ? g : 0].p + V[g < n_seg ? g : 0].off;
  }
  SG.R_seg = R;
  static const int wdebug = tante_ablate_env("TANTE_WGRAD_DEBUG");  // -DTANTE_ABLATE builds only
  static TantePerDevice attr;
  attr.once([&] {
    hipFuncSetAttribute((const void*)wgrad_tr_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * WCHUNK);
    hipFuncSetAttribute((const void*)wgrad_tr_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * WCHUNK);
    hipFuncSetAttribute((const void*)wgrad_tr_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 4 * 2 * WCHUNK);
  });
  const unsigned n_wg = 8u * (unsigned)((total + 7) / 8) * (unsigned)(ti * tj);
  const int64_t need = ((int64_t)total * ti * tj * WT * WT + (int64_t)total * I) * (int64_t)sizeof(float);
  const bool use_slab = want_slab && total > 1 && need <= ws_bytes && ((uintptr_t)ws % 16) == 0;
  float* slab = use_slab ? (float*)ws : nullptr;
  float* bias_slab = use_slab ? slab + (int64_t)total * ti * tj * WT * WT : nullptr;
  // the 8-deep ring (one workgroup per CU with 112 KB in flight) is kept for experiments only: measured on the train step it LOSES to
  // the 4-deep one (33.2 vs 31.0 ms when used for grids of <= 256 workgroups, 32.3 ms when forced everywhere)
  if (RGv == 2 && use_slab)
    hipLaunchKernelGGL((wgrad_tr_kernel<4, 2>), dim3(n_wg), dim3(512), (size_t)2 * 4 * 2 * WCHUNK, s, SG, (long)U[0].s0, (long)V[0].s0, R * n_seg, I, J, per, dW,
                       dbias, layout, P, C_other, swap, wdebug, (int)total, slab, bias_slab);
  else if (deep_env > 0)
    hipLaunchKernelGGL((wgrad_tr_kernel<8, 1>), dim3(n_wg), dim3(256), (size_t)8 * 2 * WCHUNK, s, SG, (long)U[0].s0, (long)V[0].s0, R * n_seg, I, J, per, dW,
                       dbias, layout, P, C_other, swap, wdebug, (int)total, slab, bias_slab);
  else
    hipLaunchKernelGGL((wgrad_tr_kernel<4, 1>), dim3(n_wg), dim3(256), (size_t)4 * 2 * WCHUNK, s, SG, (long)U[0].s0, (long)V[0].s0, R * n_seg, I, J, per, dW,
                       dbias, layout, P, C_other, swap, wdebug, (int)total, slab, bias_slab);
  if (use_slab) {
    const int ny_env = tante_opt("TANTE_WGRAD_REDUCE_NY", 0);
    const int ntile = ti * tj, ny = ny_env > 0 ? ny_env : (total >= 16 ? 4 : 1);   // more chunks = more atomics per output: 16 chunks measured 39 us against 14 us for 4
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(ntile * 16), (unsigned)ny), dim3(256), 0, s, slab, bias_slab, (int)total, ntile, ti, I, J,
                       dW, dbias, layout, P, C_other, swap);
  }
  return true;
}

static int wgrad_one(const TanteRowMat* U, const TanteRowMat* V, int64_t R, int I, int J, float* dW, float* dbias, int layout, int P, int C_other,
                     int swap, int compute, hipStream_t s, float* ws = nullptr, int64_t ws_bytes = 0) {
  if (compute == TANTE_BF16 && wgrad_tr_launch(U, V, 1, (long)R, I, J, dW, dbias, layout, P, C_other, swap, s)) {
    TANTE_CHECK_LAUNCH();
    return 0;
  }
  const int mu = rm_stage_mode(*U, I), mv = rm_stage_mode(*V, J);
  if (compute == TANTE_BF16) {
    if (I > 64 || J > 64) launch_wgrad<true, 128>(*U, *V, (long)R, I, J, dW, dbias, layout, P, C_other, swap, mu, mv, s, ws, ws_bytes);
    else launch_wgrad<true, 64>(*U, *V, (long)R, I, J, dW, dbias, layout, P, C_other, swap, mu, mv, s, ws, ws_bytes);
  } else {
    launch_wgrad<false, 64>(*U, *V, (long)R, I, J, dW, dbias, layout, P, C_other, swap, mu, mv, s, ws, ws_bytes);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_wgrad_ws(const TanteRowMat* U, const TanteRowMat* V, int64_t R, int I, int J, float* dW, float* dbias, int layout, int P,
                              int C_other, int swap, int compute, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
extern "C" int tante_wgrad(const TanteRowMat* U, const TanteRowMat* V, int64_t R, int I, int J, float* dW, float* dbias, int layout, int P,
                           int C_other, int swap, int compute, int accumulate, void* stream) {
  return tante_wgrad_ws(U, V, R, I, J, dW, dbias, layout, P, C_other, swap, compute, accumulate, nullptr, 0, stream);
}
extern "C" int tante_wgrad_ws(const TanteRowMat* U, const TanteRowMat* V, int64_t R, int I, int J, float* dW, float* dbias, int layout, int P,
                              int C_other, int swap, int compute, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!U || !V || !dW || R <= 0 || I <= 0 || J <= 0) TANTE_FAIL(-1, "tante_wgrad: bad argument");
  int rc = check_rowmat(*U, "U");
  if (rc) return rc;
  rc = check_rowmat(*V, "V");
  if (rc) return rc;
  if (layout < TANTE_W_LINEAR || layout > TANTE_W_DECONV_NCHW) TANTE_FAIL(-1, "tante_wgrad: bad output layout");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && tante_zero_async(dW, (size_t)I * J * sizeof(float), s) != hipSuccess) TANTE_FAIL(-3, "tante_wgrad: memset failed");
  if (!accumulate && dbias && tante_zero_async(dbias, (size_t)I * sizeof(float), s) != hipSuccess) TANTE_FAIL(-3, "tante_wgrad: memset failed");
  float* ws = (workspace && workspace_bytes > 0 && ((uintptr_t)workspace % 16) == 0) ? (float*)workspace : nullptr;
  return wgrad_one(U, V, R, I, J, dW, dbias, layout, P, C_other, swap, compute, s, ws, ws ? workspace_bytes : 0);
}

extern "C" int tante_wgrad_multi(const TanteRowMat* U, const TanteRowMat* V, int n_seg, int64_t R, int I, int J, float* dW, float* dbias,
                                 int layout, int P, int C_other, int swap, int compute, int accumulate, void* stream) {
  return tante_wgrad_multi_ws(U, V, n_seg, R, I, J, dW, dbias, layout, P, C_other, swap, compute, accumulate, nullptr, 0, stream);
}

extern "C" int tante_wgrad_multi_ws(const TanteRowMat* U, const TanteRowMat* V, int n_seg, int64_t R, int I, int J, float* dW, float* dbias,
                                    int layout, int P, int C_other, int swap, int compute, int accumulate, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
  if (!U || !V || !dW || n_seg <= 0 || R <= 0 || I <= 0 || J <= 0) TANTE_FAIL(-1, "tante_wgrad_multi: bad argument");
  for (int g = 0; g < n_seg; ++g) {
    int rc = check_rowmat(U[g], "U");
    if (rc) return rc;
    rc = check_rowmat(V[g], "V");
    if (rc) return rc;
  }
  if (layout < TANTE_W_LINEAR || layout > TANTE_W_DECONV_NCHW) TANTE_FAIL(-1, "tante_wgrad_multi: bad output layout");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && tante_zero_async(dW, (size_t)I * J * sizeof(float), s) != hipSuccess) TANTE_FAIL(-3, "tante_wgrad_multi: memset failed");
  if (!accumulate && dbias && tante_zero_async(dbias, (size_t)I * sizeof(float), s) != hipSuccess) TANTE_FAIL(-3, "tante_wgrad_multi: memset failed");
  int g = 0;
  while (g < n_seg) {                                  // groups of up to WSEG segments share a launch when the shape allows
    const int n = n_seg - g < WSEG ? n_seg - g : WSEG;
    if (compute == TANTE_BF16 && n > 1 && wgrad_tr_launch(U + g, V + g, n, (long)R, I, J, dW, dbias, layout, P, C_other, swap, s, workspace, workspace_bytes)) {
      TANTE_CHECK_LAUNCH();
      g += n;
      continue;
    }
    const int rc = wgrad_one(U + g, V + g, R, I, J, dW, dbias, layout, P, C_other, swap, compute, s);
    if (rc) return rc;
    ++g;
  }
  return 0;
}

/* Up to four weight gradients (each: n_seg operand pairs of R rows, as tante_wgrad_multi_ws with accumulate = 1) as the jobs of ONE
 * launch + one reduce launch, sharing the chip's workgroups: see wgrad_tr_jobs_kernel.  Jobs outside the dense-bf16 shape rules, a
 * single job, or a workspace too small: each job runs as tante_wgrad_multi_ws. */
extern "C" int tante_wgrad_jobs_ws(const TanteWgradJob* jobs, int n_jobs, int compute, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!jobs || n_jobs <= 0) TANTE_FAIL(-1, "tante_wgrad_jobs_ws: bad argument");
  hipStream_t s = (hipStream_t)stream;
  auto one_by_one = [&]() -> int {
    for (int k = 0; k < n_jobs; ++k) {
      const TanteWgradJob& jb = jobs[k];
      const int rc = tante_wgrad_multi_ws(jb.U, jb.V, jb.n_seg, jb.R, jb.I, jb.J, jb.dW, jb.dbias, jb.layout, jb.P, jb.C_other, jb.swap, compute, 1, workspace,
                                          workspace_bytes, stream);
      if (rc) return rc;
    }
    return 0;
  };
  bool ok = compute == TANTE_BF16 && n_jobs >= 2 && n_jobs <= WJOBS && workspace && ((uintptr_t)workspace % 16) == 0 && tante_opt("TANTE_WGRAD_JOBS", 1) &&
            !tante_opt("TANTE_WGRAD_NO_SLAB", 0) && !tante_opt("TANTE_WGRAD_NO_TR", 0);
  double work[WJOBS] = {0, 0, 0, 0}, work_sum = 0;
  for (int k = 0; ok && k < n_jobs; ++k) {
    const TanteWgradJob& jb = jobs[k];
    if (!jb.U || !jb.V || !jb.dW || jb.n_seg < 1 || jb.n_seg > WSEG || jb.R <= 0 || jb.R % WRC || jb.I <= 0 || jb.J <= 0 || jb.I % WT || jb.J % WT) { ok = false; break; }
    if (jb.layout < TANTE_W_LINEAR || jb.layout > TANTE_W_DECONV_NCHW) { ok = false; break; }
    for (int g = 0; g < jb.n_seg; ++g)
      if (!jb.U[g].p || !jb.V[g].p || !wgrad_tr_ok(jb.U[g], jb.R, jb.I) || !wgrad_tr_ok(jb.V[g], jb.R, jb.J) || jb.U[g].s0 != jb.U[0].s0 || jb.V[g].s0 != jb.V[0].s0) { ok = false; break; }
    work[k] = (double)(jb.I / WT) * (jb.J / WT) * jb.n_seg * (double)jb.R;
    work_sum += work[k];
  }
  if (!ok) return one_by_one();
  WgJobs JB;
  RdJobs RB;
  JB.n = n_jobs;
  // Two 64 KB workgroups fit a CU -- 64 per XCD -- and each should cover the same number of row chunks: the smallest chunk count c per
  // workgroup for which no XCD gets more than its 64 (splits are dealt to the XCDs round-robin, every job starting where the last one
  // ended).  (Row ranges used to be divisors of the segment length: at cfg3, 512 chunks per segment, that meant 384 workgroups of 128
  // chunks -- half the CUs with two workgroups, half with one -- or 768 of 64 in one and a half rounds; 480 of 102 / 103 now.)
  const int wg_total = tante_opt("TANTE_WGRAD_JOBS_WGS", 256 * WGJ_OCC);
  long cper = 4;
  int rot[WJOBS] = {0, 0, 0, 0};
  {
    // workgroups on the busiest XCD when every workgroup covers at most c chunks; split b of a job runs on XCD (b + rot) % 8
    auto busiest = [&](long c) {
      long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      int r = 0;
      for (int k = 0; k < n_jobs; ++k) {
        const long nch = jobs[k].R / WRC, ntile = (long)(jobs[k].I / WT) * (jobs[k].J / WT);
        const long total = ((nch + c - 1) / c) * jobs[k].n_seg;
        rot[k] = r;
        for (int x = 0; x < 8; ++x) load[(x + r) & 7] += (total / 8 + (x < total % 8 ? 1 : 0)) * ntile;
        r = (int)((r + total) & 7);
      }
      long m = 0;
      for (int x = 0; x < 8; ++x) m = std::max(m, load[x]);
      return m;
    };
    long lo = 4, hi = 4;
    for (int k = 0; k < n_jobs; ++k) hi = std::max<long>(hi, jobs[k].R / WRC);
    while (lo < hi) {      // (close to monotone in c; the final check below settles it)
      const long mid = (lo + hi) / 2;
      if (busiest(mid) * 8 <= wg_total) hi = mid; else lo = mid + 1;
    }
    cper = lo;
    while (busiest(cper) * 8 > wg_total && cper < hi) ++cper;
    (void)busiest(cper);      // rot[] of the chosen c
  }
  unsigned wg_begin = 0, max_red_x = 0;
  int64_t ws_off = 0;      // floats
  for (int k = 0; k < n_jobs; ++k) {
    const TanteWgradJob& jb = jobs[k];
    const int ti = jb.I / WT, tj = jb.J / WT, ntile = ti * tj;
    const long nch = jb.R / WRC;
    long split = (nch + cper - 1) / cper;      // splits per segment
    if (split < 1) split = 1;
    if (split > nch) split = nch;
    const long per = 0;
    const long total = split * jb.n_seg;
    if (total > 65535) return one_by_one();
    WgJob& w = JB.j[k];
    for (int g = 0; g < WSEG; ++g) {
      const TanteRowMat& u = jb.U[g < jb.n_seg ? g : 0];
      const TanteRowMat& v = jb.V[g < jb.n_seg ? g : 0];
      w.SG.U[g] = (const unsigned short*)u.p + u.off;
      w.SG.V[g] = (const unsigned short*)v.p + v.off;
    }
    w.SG.R_seg = (long)jb.R;
    w.ldu = (long)jb.U[0].s0; w.ldv = (long)jb.V[0].s0; w.R = (long)jb.R * jb.n_seg; w.per = per;
    w.dW = jb.dW; w.dbias = jb.dbias;
    w.slab = (float*)workspace + ws_off;
    w.bias_slab = w.slab + (int64_t)total * ntile * WT * WT;
    ws_off += ((int64_t)total * ntile * WT * WT + (int64_t)total * jb.I + 3) / 4 * 4;
    w.I = jb.I; w.J = jb.J; w.layout = jb.layout; w.P = jb.P; w.Co = jb.C_other; w.swap = jb.swap; w.n_split = (int)total; w.wg_begin = (int)wg_begin; w.seg_splits = (int)split; w.xcd_rot = rot[k];
    wg_begin += 8u * (unsigned)((total + 7) / 8) * (unsigned)ntile;
    RdJob& r = RB.j[k];
    r.slab = w.slab; r.bias_slab = w.bias_slab; r.dW = jb.dW; r.dbias = jb.dbias; r.n_split = (int)total; r.ntile = ntile; r.ti = ti; r.I = jb.I; r.J = jb.J;
    r.layout = jb.layout; r.P = jb.P; r.Co = jb.C_other; r.swap = jb.swap;
    if ((unsigned)(ntile * 16) > max_red_x) max_red_x = (unsigned)(ntile * 16);
  }
  for (int k = n_jobs; k < WJOBS; ++k) { JB.j[k] = JB.j[0]; JB.j[k].wg_begin = 0x7fffffff; RB.j[k] = RB.j[0]; RB.j[k].ntile = 0; }
  if (ws_off * (int64_t)sizeof(float) > workspace_bytes) return one_by_one();
  static TantePerDevice attr;
  attr.once([&] { hipFuncSetAttribute((const void*)wgrad_tr_jobs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WGJ_NBUF * 2 * WCHUNK); });
  hipLaunchKernelGGL(wgrad_tr_jobs_kernel, dim3(wg_begin), dim3(256), (size_t)WGJ_NBUF * 2 * WCHUNK, s, JB);
  hipLaunchKernelGGL(wgrad_reduce_jobs_kernel, dim3(max_red_x, 1, (unsigned)n_jobs), dim3(256), 0, s, RB);
  TANTE_CHECK_LAUNCH();
  return 0;
}
