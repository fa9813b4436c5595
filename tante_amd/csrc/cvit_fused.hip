// CViT blocks at width 512 (cfg4: emb_dim = dec_emb_dim = 512, 8 heads x 64, mlp_ratio 1), bf16 compute: everything behind the attention
// of a SelfAttnBlock / CrossAttnBlock in ONE launch (reference models/cvit.py:112-169), and -- for the decoder's last block -- the model's
// tail as well (cvit.py:459-466: norm2 -> Mlp (x = LN(x + gelu(dense(x)))) -> output layer):
//
//   MODE 0   x1 = out_proj(a) + resid;  x2 = x1 + fc2(gelu(fc1(LN2(x1))))                                   -> x2 (M, 512) fp32
//   MODE 1   ... z = norm2(x2);  y = z + gelu(dense(z));  out = output_layer(LN(y))                         -> out (M, out_dim <= 16) fp32
//
// Before: 3 (MODE 0) / 7 (MODE 1) launches that each moved the (M, 512) token matrix through HBM in fp32 (65 536 queries per sample:
// 0.8 - 1.3 GB and 360 - 420 us per launch at B = 4, 15 % of the matrix peak).  Here a workgroup owns 64 tokens and keeps them on chip.
//
// The scheme is block_sliced.hip's, twice as wide: 8 waves, wave w owns OUTPUT features 64 w .. 64 w + 63 of every GEMM for all 64
// tokens (4 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16, D[feature][token]).  Its weights are its own: they stream L2 -> registers
// as pre-packed 1 KiB operand fragments (64 KiB per wave and matrix, a register ring two k-steps ahead that runs across the matrices
// and the phases between them).  What the waves share are the activations: two bf16 images [token][512] in LDS (16-byte chunks
// XOR-swizzled by the row, every ds_read_b128 fragment conflict-free and feeding 4 MFMAs), ping-ponged between the stages.  The fp32
// residual (x1, later z) stays in the accumulator layout in registers; LayerNorm statistics cross the waves through a 4 KB table as
// (mean, M2) pairs of the 64-feature slices, combined exactly (no E[x^2] - E[x]^2 cancellation).  The 16-row output layer needs no
// image at all: the normalised accumulator tiles, packed to bf16 in pairs, ARE its B operand (its weight fragments are packed in the
// matching k order), each wave contributes the partial product over its 64 features and four waves add the eight partials.
//
// MODE 2 / 3 put the front of a CrossAttnBlock in the same launch (LN1 -> q projection -> attention against K / V that are the same for
// every query of a sample): see xfront below.
#include "common.hip.h"
#include "fused_common.hip.h"
#include "fs_common.hip.h"

namespace {

constexpr int XC = 512;              // width
constexpr int XROW = 1024;           // bytes per image row
// tokens per workgroup = 16 NTT: 64 for the decoder's 65 536 queries per sample; 16 for the encoder's few hundred tokens, where the
// launch is a latency chain and a quarter of the tokens per workgroup means four times the CUs (one 64-token workgroup per 64 tokens
// of a 256-token sample left 252 CUs idle: 39 us per launch against 26 for the three per-op launches it replaced)
__host__ __device__ constexpr int x_img(int ntt) { return 16 * ntt * XROW; }
__host__ __device__ constexpr int x_lds(int ntt) { return 2 * x_img(ntt) + 8 * 16 * ntt * 2 * 4 + (ntt < 4 ? 8 * ntt * 1024 : 0); }
// weight ring, k-steps ahead: 2 where a k-step is 16 MFMAs per wave (64 tokens); 6 for the 16-token form, whose k-step is 4 MFMAs = 64
// clocks against an L2 round trip of many hundred (its 129 registers leave the room)
#ifndef X_PF_SMALL
#define X_PF_SMALL 6
#endif
__host__ __device__ constexpr int x_pf(int ntt) { return ntt >= 4 ? 2 : X_PF_SMALL; }
constexpr int XMAT = XC * XC * 2;    // one packed matrix: 512 KiB

struct ChainArgs {
  const unsigned short* a;   // (M, 512) bf16 rows: the attention output
  const float* resid;        // (resid_period, 512) fp32: row (token % resid_period) is added after out_proj
  long resid_period;
  const char* w;             // packed matrices: out_proj | fc1 (LN2 folded) | fc2 [| dense]
  const float* bias;         // (n_mat, 512) fp32 (folded where the weights are)
  const float* g2;           // norm2 affine (MODE 1)
  const float* b2;
  float eps_ln2, eps_norm2, eps_mlp;
  const char* wout;          // output layer: 16 fragments of 1 KiB (wave, pair), LN of the Mlp folded in
  const float* bout;         // 16 floats
  int out_dim;
  float* out;
  unsigned short* qkv;       // MODE 2: (M, 1536) bf16, the NEXT block's q | k | v rows
};

// byte offset of fragment (global k-step g, row tile j) of a wave's stream: matrix g / 16, inside it [wave][ks][j]
__host__ __device__ constexpr long x_woff(int g, int j) { return (long)(g >> 4) * XMAT + (long)(((g & 15) * 4 + j) * FS_FRAG); }

// acc[j][tt] += W[row tile j of the wave's slice][all 512 k] . image[token tile tt]; the ring keeps prefetching into the NEXT matrix
template <int M, int GEND, int NTT>
__device__ __forceinline__ void x_gemm(const FsW& wu, u32x4 (&wb)[x_pf(NTT) + 1][4], const unsigned (&ab)[4], f32x4 (&acc)[4][NTT]) {
  constexpr int XPF = x_pf(NTT);
  mfma_stream<16 * NTT, 4>(
      [&](auto ic) {
        constexpr int i = decltype(ic)::value, ks = i / NTT, tt = i % NTT;
        return LdsAddr<tt * 16384 + (ks >> 2) * 256>{ab[ks & 3]};
      },
      [&](auto ic, const u32x4& tf) {
        constexpr int i = decltype(ic)::value, ks = i / NTT, tt = i % NTT, g = 16 * M + ks;
        if constexpr (tt == 0 && g + XPF < GEND) {
#pragma unroll
          for (int j = 0; j < 4; ++j) wb[(g + XPF) % (XPF + 1)][j] = ldg_frag(wu, (int)x_woff(g + XPF, j));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][tt] = mfma_bf16(wb[g % (XPF + 1)][j], tf, acc[j][tt]);
      });
}

// accumulator tiles -> the wave's 64 columns of a bf16 image (8 bytes per lane and tile: features 4 kk .. 4 kk + 3 of token l15)
__device__ __forceinline__ void x_img_write(char* img, int wave, int kk, int l15, int j, int tt, const f32x4& v) {
  const int chunk = 8 * (wave & 1) + 2 * j + (kk >> 1);
  u32x2 p;
  p[0] = pack_bf16x2(v[0], v[1]);
  p[1] = pack_bf16x2(v[2], v[3]);
  *(u32x2*)(img + (16 * tt + l15) * XROW + (wave >> 1) * 256 + ((chunk ^ l15) << 4) + (kk & 1) * 8) = p;
}

// LayerNorm statistics of the 512 features of every token from the waves' 64-feature slices: (mean, M2) per slice, combined by
// Chan's formula.  ONE barrier inside; returns mean and 1 / sqrt(var + eps) of token 16 tt + l15.
template <int NTT>
__device__ __forceinline__ void x_ln_stats(const f32x4 (&x)[4][NTT], float* tab, int wave, int kk, int l15, float eps, float (&mean)[NTT], float (&rstd)[NTT]) {
  constexpr int XT = 16 * NTT;
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (x[j][tt][0] + x[j][tt][1]) + (x[j][tt][2] + x[j][tt][3]);
    const float m = rows_sum(s) * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = x[j][tt][r] - m; q = fmaf(d, d, q); }
    q = rows_sum(q);
    if (kk == 0) *(float2*)(tab + ((wave * XT + 16 * tt + l15) << 1)) = float2{m, q};
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    float2 t[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) t[w] = *(const float2*)(tab + ((w * XT + 16 * tt + l15) << 1));
    float m = 0.f, q = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) m += t[w].x;
    m *= 0.125f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { const float d = t[w].x - m; q += t[w].y + 64.0f * d * d; }
    mean[tt] = m;
    rstd[tt] = rsqrtf(q * (1.0f / XC) + eps);
  }
}

template <int MODE, int NTT>
__global__ __launch_bounds__(512) void chain512_kernel(ChainArgs A) {
  constexpr int XT = 16 * NTT, XIMG = x_img(NTT), XPF = x_pf(NTT);
  extern __shared__ __attribute__((aligned(16))) char xs[];
  char* img0 = xs;
  char* img1 = xs + XIMG;
  float* tab = (float*)(xs + 2 * XIMG);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  const long tok0 = (long)blockIdx.x * XT;
  constexpr int NMAT = MODE == 0 ? 3 : (MODE == 1 ? 4 : 6), GEND = 16 * NMAT;
  const FsW wu = fs_wstream(A.w, (unsigned)(wave * 65536 + lane * 16));      // base in SGPRs, one VGPR lane offset, fragment offsets by the scalar unit
  const int f0 = 64 * wave + 4 * kk;      // + 16 j: the lane's four features of row tile j

  u32x4 wb[XPF + 1][4];
#pragma unroll
  for (int p = 0; p < XPF; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) wb[p][j] = ldg_frag(wu, (int)x_woff(p, j));

  // out_proj accumulators start from bias + residual rows
  f32x4 acc[4][NTT];
  {
    const float* rp = A.resid + ((tok0 % A.resid_period) + l15) * XC + f0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 b = *(const f32x4*)(A.bias + f0 + 16 * j);
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) acc[j][tt] = *(const f32x4*)(rp + (long)(16 * tt) * XC + 16 * j) + b;
    }
  }
  // the attention output rows -> image 0 (a wave instruction = one 1 KiB row)
#pragma unroll
  for (int p = 0; p < 2 * NTT; ++p) {
    const int row = p * 8 + wave, c = lane;
    const u32x4 v = *(const u32x4*)(A.a + (tok0 + row) * XC + c * 8);
    *(u32x4*)(img0 + row * XROW + (c >> 4) * 256 + (((c & 15) ^ (row & 15)) << 4)) = v;
  }
  unsigned ab0[4], ab1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ab0[j] = lds_addr(img0) + l15 * XROW + (((4 * j + kk) ^ l15) << 4);
    ab1[j] = ab0[j] + XIMG;
  }
  __syncthreads();

  // ---- x1 = out_proj(a) + resid -------------------------------------------------------------------------------------------------
  x_gemm<0, GEND, NTT>(wu, wb, ab0, acc);
  f32x4 x1[4][NTT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) x1[j][tt] = acc[j][tt];
  {
    float mean[NTT], rstd[NTT];
    x_ln_stats<NTT>(x1, tab, wave, kk, l15, A.eps_ln2, mean, rstd);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) x_img_write(img1, wave, kk, l15, j, tt, (x1[j][tt] - splat4(mean[tt])) * splat4(rstd[tt]));
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 b = *(const f32x4*)(A.bias + XC + f0 + 16 * j);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) acc[j][tt] = b;
  }
  __syncthreads();

  // ---- h = gelu(fc1'(LN2(x1))) ----------------------------------------------------------------------------------------------------
  x_gemm<1, GEND, NTT>(wu, wb, ab1, acc);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) x_img_write(img0, wave, kk, l15, j, tt, gelu_poly4<false>(acc[j][tt]));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x4 b = *(const f32x4*)(A.bias + 2 * XC + f0 + 16 * j);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) acc[j][tt] = x1[j][tt] + b;
  }
  __syncthreads();

  // ---- x2 = x1 + fc2(h) -----------------------------------------------------------------------------------------------------------
  x_gemm<2, GEND, NTT>(wu, wb, ab0, acc);
  if constexpr (MODE == 0 || MODE == 2) {
    float* op = A.out + (tok0 + l15) * XC + f0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) *(f32x4*)(op + (long)(16 * tt) * XC + 16 * j) = acc[j][tt];
  }
  if constexpr (MODE == 2) {
    // ---- the NEXT SelfAttnBlock's q | k | v = in_proj'(LN1(x2)) (cvit.py:129-134 of the block that follows; LN1 folded into the weights):
    // the token rows are here, normalised once, and the three 512 x 512 matrices ride the same weight stream -- instead of a launch
    // of its own that re-reads x2 and, on the encoder's few hundred tokens, is one more latency chain per block
    {
      float mean[NTT], rstd[NTT];
      x_ln_stats<NTT>(acc, tab, wave, kk, l15, A.eps_norm2, mean, rstd);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) x_img_write(img1, wave, kk, l15, j, tt, (acc[j][tt] - splat4(mean[tt])) * splat4(rstd[tt]));
    }
    __syncthreads();
    auto part = [&](auto pc) {
      constexpr int P = decltype(pc)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 b = *(const f32x4*)(A.bias + (3 + P) * XC + f0 + 16 * j);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) acc[j][tt] = b;
      }
      x_gemm<3 + P, GEND, NTT>(wu, wb, ab1, acc);
      unsigned short* qp = A.qkv + (tok0 + l15) * (3 * XC) + P * XC + f0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          u32x2 v;
          v[0] = pack_bf16x2(acc[j][tt][0], acc[j][tt][1]);
          v[1] = pack_bf16x2(acc[j][tt][2], acc[j][tt][3]);
          *(u32x2*)(qp + (long)(16 * tt) * (3 * XC) + 16 * j) = v;
        }
    };
    part(std::integral_constant<int, 0>{});
    part(std::integral_constant<int, 1>{});
    part(std::integral_constant<int, 2>{});
  }
  if constexpr (MODE == 1) {
    // ---- z = norm2(x2) (affine: z is also the Mlp's residual) ---------------------------------------------------------------------
    {
      float mean[NTT], rstd[NTT];
      x_ln_stats<NTT>(acc, tab, wave, kk, l15, A.eps_norm2, mean, rstd);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 g = *(const f32x4*)(A.g2 + f0 + 16 * j), b = *(const f32x4*)(A.b2 + f0 + 16 * j);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          x1[j][tt] = (acc[j][tt] - splat4(mean[tt])) * splat4(rstd[tt]) * g + b;
          x_img_write(img1, wave, kk, l15, j, tt, x1[j][tt]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 b = *(const f32x4*)(A.bias + 3 * XC + f0 + 16 * j);
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) acc[j][tt] = b;
    }
    __syncthreads();
    // ---- y = z + gelu(dense(z)) ---------------------------------------------------------------------------------------------------
    x_gemm<3, GEND, NTT>(wu, wb, ab1, acc);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) x1[j][tt] += gelu_poly4<false>(acc[j][tt]);
    float mean[NTT], rstd[NTT];
    x_ln_stats<NTT>(x1, tab, wave, kk, l15, A.eps_mlp, mean, rstd);
    // ---- out = output_layer'(LN(y)): the normalised tiles, packed in pairs, are the B operand of the wave's two k-steps ----------------
    const u32x4 wo0 = ldg_frag(A.wout + (wave * 2 + 0) * FS_FRAG + lane * 16), wo1 = ldg_frag(A.wout + (wave * 2 + 1) * FS_FRAG + lane * 16);
    // [wave][tt][lane]: inside image 0 (last read by the fc2 GEMM, two barriers ago) where it is large enough, behind the table otherwise
    f32x4* part = NTT == 4 ? (f32x4*)img0 : (f32x4*)(xs + 2 * XIMG + 8 * XT * 2 * 4);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const f32x4 m4 = splat4(mean[tt]), r4 = splat4(rstd[tt]);
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
      o = mfma_bf16(wo0, pack8((x1[0][tt] - m4) * r4, (x1[1][tt] - m4) * r4), o);
      o = mfma_bf16(wo1, pack8((x1[2][tt] - m4) * r4, (x1[3][tt] - m4) * r4), o);
      part[(wave * NTT + tt) * 64 + lane] = o;
    }
    __syncthreads();
    if (wave < NTT) {
      const int tt = wave;
      f32x4 o = *(const f32x4*)(A.bout + 4 * kk);
#pragma unroll
      for (int w = 0; w < 8; ++w) o += part[(w * NTT + tt) * 64 + lane];
      float* op = A.out + (tok0 + 16 * tt + l15) * A.out_dim + 4 * kk;
      if (A.out_dim == 16) {
        *(f32x4*)op = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * kk + r < A.out_dim) op[r] = o[r];
      }
    }
  }
}

TantePerDevice g_chain_attr[6];

template <int MODE, int NTT>
void chain_launch(const ChainArgs& A, long M, hipStream_t s) {
  g_chain_attr[MODE * 2 + (NTT == 4)].once(
      [] { (void)hipFuncSetAttribute((const void*)chain512_kernel<MODE, NTT>, hipFuncAttributeMaxDynamicSharedMemorySize, x_lds(NTT)); });
  hipLaunchKernelGGL((chain512_kernel<MODE, NTT>), dim3((unsigned)(M / (16 * NTT))), dim3(512), x_lds(NTT), s, A);
}

}  // namespace

extern "C" int tante_cvit_chain512(const void* a, const float* resid, int64_t resid_period, const void* w, const float* bias, const float* g2,
                                   const float* b2, float eps_ln2, float eps_norm2, float eps_mlp, const void* wout, const float* bout,
                                   int out_dim, int64_t M, int mode, float* out, void* stream) {
  if (mode != 0 && mode != 1) TANTE_FAIL(-1, "tante_cvit_chain512: mode must be 0 (block tail) or 1 (block tail + model tail), got %d", mode);
  if (M <= 0 || M % 16) TANTE_FAIL(-1, "tante_cvit_chain512: M = %lld must be a positive multiple of 16", (long long)M);
  if (resid_period <= 0 || resid_period % 16) TANTE_FAIL(-1, "tante_cvit_chain512: resid_period = %lld must be a positive multiple of 16", (long long)resid_period);
  if (!a || !resid || !w || !bias || !out) TANTE_FAIL(-1, "tante_cvit_chain512: null operand");
  if (mode == 1 && (!g2 || !b2 || !wout || !bout || out_dim < 1 || out_dim > 16))
    TANTE_FAIL(-1, "tante_cvit_chain512: mode 1 needs norm2, the packed output layer and 1 <= out_dim <= 16 (got %d)", out_dim);
  if (((uintptr_t)a | (uintptr_t)resid | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)out | (uintptr_t)g2 | (uintptr_t)b2 | (uintptr_t)wout | (uintptr_t)bout) & 15)
    TANTE_FAIL(-1, "tante_cvit_chain512: operands must be 16-byte aligned");
  ChainArgs A{(const unsigned short*)a, resid, (long)resid_period, (const char*)w, bias, g2, b2, eps_ln2, eps_norm2, eps_mlp, (const char*)wout, bout, out_dim, out, nullptr};
  hipStream_t s = (hipStream_t)stream;
  // 64-token workgroups once they fill the chip twice over (and the shapes allow), 16-token workgroups for the short launches
  const int tw = tante_opt("TANTE_CVIT_CHAIN_TOKENS", 0);
  const bool wide = tw ? tw == 64 : (M >= 64 * 512);
  if (wide && M % 64 == 0 && resid_period % 64 == 0) {
    if (mode == 0) chain_launch<0, 4>(A, (long)M, s); else chain_launch<1, 4>(A, (long)M, s);
  } else {
    if (mode == 0) chain_launch<0, 1>(A, (long)M, s); else chain_launch<1, 1>(A, (long)M, s);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}


/* mode 0 of tante_cvit_chain512 followed, in the same launch, by the NEXT SelfAttnBlock's input projection: qkv (M, 1536) bf16 =
 * in_proj'(LN1_next(out)) with LN1 folded into the weights.  w: out_proj | fc1 | fc2 | Wq | Wk | Wv (six packed matrices), bias (6, 512). */
extern "C" int tante_cvit_chain512_qkv(const void* a, const float* resid, int64_t resid_period, const void* w, const float* bias, float eps_ln2,
                                       float eps_ln1_next, int64_t M, float* out, void* qkv, void* stream) {
  if (M <= 0 || M % 16) TANTE_FAIL(-1, "tante_cvit_chain512_qkv: M = %lld must be a positive multiple of 16", (long long)M);
  if (resid_period <= 0 || resid_period % 16) TANTE_FAIL(-1, "tante_cvit_chain512_qkv: resid_period = %lld must be a positive multiple of 16", (long long)resid_period);
  if (!a || !resid || !w || !bias || !out || !qkv) TANTE_FAIL(-1, "tante_cvit_chain512_qkv: null operand");
  if (((uintptr_t)a | (uintptr_t)resid | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)out | (uintptr_t)qkv) & 15) TANTE_FAIL(-1, "tante_cvit_chain512_qkv: operands must be 16-byte aligned");
  ChainArgs A{(const unsigned short*)a, resid, (long)resid_period, (const char*)w, bias, nullptr, nullptr, eps_ln2, eps_ln1_next, 0.0f, nullptr, nullptr, 0, out, (unsigned short*)qkv};
  hipStream_t s = (hipStream_t)stream;
  const int tw = tante_opt("TANTE_CVIT_CHAIN_TOKENS", 0);
  const bool wide = tw ? tw == 64 : (M >= 64 * 512);
  if (wide && M % 64 == 0 && resid_period % 64 == 0) chain_launch<2, 4>(A, (long)M, s);
  else chain_launch<2, 1>(A, (long)M, s);
  TANTE_CHECK_LAUNCH();
  return 0;
}
