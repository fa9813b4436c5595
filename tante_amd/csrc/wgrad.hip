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
#include "common.cuh"

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

constexpr int TI = 64, TJ = 64, RC = 32;

template <bool BF16>
__global__ __launch_bounds__(256) void wgrad_kernel(const TanteRowMat U, const TanteRowMat V, long R, int I, int J, long rows_per_split,
                                                    float* __restrict__ dW, int layout, int P, int Co, int swap) {
  using elem_t = typename std::conditional<BF16, unsigned short, float>::type;
  constexpr int STRIDE = RC + (BF16 ? 8 : 4);  // elements per LDS row (one column of the operand, 32 r values + pad)
  __shared__ __attribute__((aligned(16))) elem_t Ut[TI * STRIDE];
  __shared__ __attribute__((aligned(16))) elem_t Vt[TJ * STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  const int i0 = blockIdx.x * TI, j0 = blockIdx.y * TJ;
  const long r_begin = (long)blockIdx.z * rows_per_split, r_end = min(R, r_begin + rows_per_split);
  const int wi = wave >> 1, wj = wave & 1;  // this wave's 32 x 32 part of the 64 x 64 tile
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int srow = tid >> 3, scol = (tid & 7) * 8;  // staging: 8 consecutive columns of one of the 32 rows
  const bool fastU = U.mode == TANTE_A_LINEAR && U.es == 1, fastV = V.mode == TANTE_A_LINEAR && V.es == 1;
  for (long r0 = r_begin; r0 < r_end; r0 += RC) {
    float u[8], v[8];
    const long r = r0 + srow;
    const bool rok = r < r_end;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = i0 + scol + e, cj = j0 + scol + e;
      u[e] = (rok && ci < I) ? (fastU ? ldg(U.p, U.dtype, (r / U.n0) * U.s1 + (r % U.n0) * U.s0 + U.off + ci) : rm_elem(U, r, ci)) : 0.f;
      v[e] = (rok && cj < J) ? (fastV ? ldg(V.p, V.dtype, (r / V.n0) * V.s1 + (r % V.n0) * V.s0 + V.off + cj) : rm_elem(V, r, cj)) : 0.f;
    }
    __syncthreads();  // the previous chunk's fragments have been consumed
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (BF16) {
        Ut[(scol + e) * STRIDE + srow] = (unsigned short)(pack_bf16x2(u[e], 0.f) & 0xffff);
        Vt[(scol + e) * STRIDE + srow] = (unsigned short)(pack_bf16x2(v[e], 0.f) & 0xffff);
      } else {
        Ut[(scol + e) * STRIDE + srow] = u[e];
        Vt[(scol + e) * STRIDE + srow] = v[e];
      }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const elem_t* up = Ut + (wi * 32 + a * 16 + l15) * STRIDE;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const elem_t* vp = Vt + (wj * 32 + b * 16 + l15) * STRIDE;
        if constexpr (BF16) {  // one k-step of 32 rows: lane chunk = rows 8*kk .. 8*kk+7
          const u32x4 af = *(const u32x4*)(up + kk * 8), bf = *(const u32x4*)(vp + kk * 8);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[a][b], 0, 0, 0);
        } else {  // two blocks of 16 rows, 4 MFMA steps each; lane chunk = rows 16*blk + 4*kk .. +3 (same order for both operands)
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) {
            const f32x4 af = *(const f32x4*)(up + blk * 16 + kk * 4), bf = *(const f32x4*)(vp + blk * 16 + kk * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc[a][b], 0, 0, 0);
          }
        }
      }
    }
  }
  // D[row = i][col = j]: lane holds j = l15, i = 4*kk + reg
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int i = i0 + wi * 32 + a * 16 + kk * 4 + rg, j = j0 + wj * 32 + b * 16 + l15;
        if (i < I && j < J) atomicAdd(&dW[out_index(layout, i, j, I, J, P, Co, swap)], acc[a][b][rg]);
      }
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

extern "C" int tante_wgrad(const TanteRowMat* U, const TanteRowMat* V, int64_t R, int I, int J, float* dW, int layout, int P, int C_other,
                           int swap, int compute, int accumulate, void* stream) {
  if (!U || !V || !dW || R <= 0 || I <= 0 || J <= 0) TANTE_FAIL(-1, "tante_wgrad: bad argument");
  int rc = check_rowmat(*U, "U");
  if (rc) return rc;
  rc = check_rowmat(*V, "V");
  if (rc) return rc;
  if (layout < TANTE_W_LINEAR || layout > TANTE_W_DECONV_NCHW) TANTE_FAIL(-1, "tante_wgrad: bad output layout");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && hipMemsetAsync(dW, 0, (size_t)I * J * sizeof(float), s) != hipSuccess) TANTE_FAIL(-3, "tante_wgrad: memset failed");
  const int ti = (I + TI - 1) / TI, tj = (J + TJ - 1) / TJ;
  long split = 1024 / ((long)ti * tj);
  if (split < 1) split = 1;
  const long max_split = (R + 255) / 256;
  if (split > max_split) split = max_split;
  if (split > 65535) split = 65535;
  long per = (R + split - 1) / split;
  per = (per + RC - 1) / RC * RC;
  split = (R + per - 1) / per;
  const dim3 grid(ti, tj, (unsigned)split);
  if (compute == TANTE_BF16)
    hipLaunchKernelGGL(wgrad_kernel<true>, grid, dim3(256), 0, s, *U, *V, (long)R, I, J, per, dW, layout, P, C_other, swap);
  else
    hipLaunchKernelGGL(wgrad_kernel<false>, grid, dim3(256), 0, s, *U, *V, (long)R, I, J, per, dW, layout, P, C_other, swap);
  TANTE_CHECK_LAUNCH();
  return 0;
}
