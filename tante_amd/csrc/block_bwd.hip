// Backward of a TransformerBlock's tail in ONE launch (bf16 training path, C = 256, hidden 256): everything between the gradient of the
// block output and the gradient of the attention output --
//
//     out = x1 + drop_mlp( W2 gelu_tanh(hpre) + b2 ),   hpre = W1' LayerNorm2(x1) + b1',   x1 = xs + drop_out( Wo o + bo )
//
//     dy2   = mask_mlp . dout / (1 - p)                              (U operand of fc2's weight gradient)
//     dhpre = (W2^T dy2) . gelu_tanh'(hpre)                          (U operand of fc1's weight gradient)
//     dxh   = W1'^T dhpre ;  dx1 = dout + rstd (dxh - mean(dxh) - xh2 mean(dxh xh2))     (LayerNorm2 backward + the skip path)
//     dy1   = mask_out . dx1 / (1 - p)                               (U operand of the out-proj weight gradient)
//     do    = Wo^T dy1                                               (what tante_attention_bwd takes)
//
// (attn_backbone.py:81-82 backwards; trainer/trainer.py:191).  Unfused this is six launches over the token matrix (dropout_bwd, two
// data-gradient GEMMs with their epilogues, LayerNorm backward, dropout_bwd, a third GEMM); here the fp32 gradient rows are read once
// and the five results written once.  Same structure as the forward kernel (block_sliced.hip): 4 waves per workgroup, a wave owns 64
// output features of every GEMM for the workgroup's 48 / 64 tokens, bf16 operand images in LDS, TRANSPOSED weights streamed from L2
// as pre-packed fragments.  The block is token-wise here (no sequences): workgroup b takes tokens 16 NTT b ...
#include "common.hip.h"
#include "fused_common.hip.h"
#include "fs_common.hip.h"
#include "block_sliced.h"

namespace {

constexpr int BT_W_BYTES = 3 * FS_C * FS_C * 2;     // W2^T | W1'^T | Wo^T as bf16 fragments, 128 KiB each

struct BtArgs {
  const float* dout;
  const unsigned short *hpre, *xh2;
  const float* st2;
  const char* w;
  float* dx1;
  unsigned short *dy2, *dhpre, *dy1, *d_o;
  long M;
  float p;
  unsigned long long seed_out, seed_mlp;
  const unsigned long long* seed_mix;   // see tante_set_seed_mix
};

// d/dx [ x Phi_tanh(x) ]  =  s (1 + 2 c x (1 - s) (1 + 3 k x^2)),  s = sigmoid(2 c (x + k x^3)),  c = sqrt(2 / pi), k = 0.044715:
// the closed form of act_df(TANTE_ACT_GELU_TANH) with one exp2 and one rcp instead of tanhf
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(2.30220805f * x * fmaf(0.044715f, x2, 1.0f));     // exp(2 u): 2 c log2(e) = 2.3022...
  const float r = __builtin_amdgcn_rcpf(1.0f + e);                                           // 1 - s
  const float s = 1.0f - r;
  return s * fmaf(1.59576912f * x * r, fmaf(0.134145f, x2, 1.0f), 1.0f);
}

__device__ __forceinline__ u32x2 pack4(const f32x4& v) {
  u32x2 u;
  u[0] = pack_bf16x2(v[0], v[1]);
  u[1] = pack_bf16x2(v[2], v[3]);
  return u;
}
__device__ __forceinline__ f32x4 unpack4(const u32x2& u) { return f32x4{bf16_lo(u[0]), bf16_hi(u[0]), bf16_lo(u[1]), bf16_hi(u[1])}; }

template <int NTT>
__global__ __launch_bounds__(256, 2) void block_tail_bwd_kernel(BtArgs A) {
#ifdef BT_PRIO
  if ((__builtin_amdgcn_s_getreg(0x1804) & 1u) == 1u) __builtin_amdgcn_s_setprio(3);      // as FS_PRIO in block_sliced.hip: the odd hardware wave slots
#endif
  constexpr int RT = 4, NW = 4, PF = 2;
  constexpr int IMG = 16 * NTT * FS_ROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const imgA = smem;               // dy2, later dy1
  char* const imgB = smem + IMG;         // dhpre
  char* const stat = smem + 2 * IMG;     // float2 [16 NTT tokens][4 waves]
  const int tid = threadIdx.x, lane = tid & 63, kk = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long tok0 = (long)blockIdx.x * (16 * NTT);
  int rdo[4], wro[RT];
#pragma unroll
  for (int j = 0; j < 4; ++j) rdo[j] = l15 * FS_ROW + ((((j ^ (l15 >> 2)) << 2) | (kk ^ (l15 & 3))) << 4);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) wro[rt] = l15 * FS_ROW + (((2 * RT * wave + 2 * rt + (kk >> 1)) ^ l15) << 4) + (kk & 1) * 8;
  const char* const wq = A.w + (RT * wave) * FS_FRAG + lane * 16;
  u32x4 wb[PF + 1][RT];
  fs_wring_prime<0, RT, PF>(wq, wb);
  const int col0 = 16 * RT * wave + 4 * kk;     // this lane's first column; row tile rt adds 16 rt
  bool lv[NTT];
  long off[NTT];                                // element offset of (token, col0)
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const long t = tok0 + 16 * tt + l15;
    lv[tt] = t < A.M;
    off[tt] = (lv[tt] ? t : 0) * FS_C + col0;
  }
  const float ksc = A.p > 0.0f ? 1.0f / (1.0f - A.p) : 1.0f;
  const unsigned long long smix = A.seed_mix ? *A.seed_mix : 0ull;
  auto drop4 = [&](const f32x4& v, unsigned long long seed, long i0) {
    if (A.p <= 0.0f) return v;
    const unsigned k01 = dropout_keep2(seed, (unsigned long long)i0, A.p), k23 = dropout_keep2(seed, (unsigned long long)i0 + 2, A.p);
    return f32x4{(k01 & 1u) ? v[0] * ksc : 0.0f, (k01 & 2u) ? v[1] * ksc : 0.0f, (k23 & 1u) ? v[2] * ksc : 0.0f, (k23 & 2u) ? v[3] * ksc : 0.0f};
  };

  // ---- P0: gradient rows in, dropped branch gradient -> image A and memory -----------------------------------------------------------
  f32x4 g[RT][NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) g[rt][tt] = *(const f32x4*)(A.dout + off[tt] + 16 * rt);
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const u32x2 u = pack4(lv[tt] ? drop4(g[rt][tt], A.seed_mlp ^ smix, off[tt] + 16 * rt) : f32x4{0.f, 0.f, 0.f, 0.f});
      *(u32x2*)(imgA + tt * 8192 + wro[rt]) = u;
      if (lv[tt]) *(u32x2*)(A.dy2 + off[tt] + 16 * rt) = u;
    }
  __syncthreads();

  // ---- P1: dact = W2^T dy2 ; dhpre = dact . gelu'(hpre) -> image B and memory ----------------------------------------------------------
  {
    f32x4 acc[RT][NTT];
    u32x2 hp[RT][NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        hp[rt][tt] = *(const u32x2*)(A.hpre + off[tt] + 16 * rt);      // in flight under the GEMM
      }
    fs_slice_gemm<0, 8, NTT, RT, false, PF>(wq, wb, imgA, rdo, acc);
    fs_wring_prime<1, RT, PF>(wq, wb);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 h = unpack4(hp[rt][tt]);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = acc[rt][tt][e] * gelu_tanh_grad(h[e]);
        const u32x2 u = pack4(d);
        *(u32x2*)(imgB + tt * 8192 + wro[rt]) = u;
        if (lv[tt]) *(u32x2*)(A.dhpre + off[tt] + 16 * rt) = u;
      }
  }
  __syncthreads();

  // ---- P2: dxh = W1'^T dhpre ; LayerNorm2 backward + skip -> dx1 ; its dropped copy -> image A and memory -------------------------------
  {
    f32x4 acc[RT][NTT];
    u32x2 xh[RT][NTT];
    float rstd[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      rstd[tt] = A.st2[2 * (lv[tt] ? tok0 + 16 * tt + l15 : 0) + 1];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        acc[rt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        xh[rt][tt] = *(const u32x2*)(A.xh2 + off[tt] + 16 * rt);
      }
    }
    fs_slice_gemm<1, 16, NTT, RT, false, PF>(wq, wb, imgB, rdo, acc);
    fs_wring_prime<2, RT, PF>(wq, wb);
    // per-token sums over ALL 256 features: this wave's 64, then the four waves' partials through LDS
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 xv = unpack4(xh[rt][tt]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1 += acc[rt][tt][e];
          s2 = fmaf(acc[rt][tt][e], xv[e], s2);
        }
      }
      s1 = rows_sum(s1);
      s2 = rows_sum(s2);
      if (kk == 0) *(float2*)(stat + ((16 * tt + l15) * NW + wave) * 8) = make_float2(s1, s2);
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const f32x4* sp = (const f32x4*)(stat + (16 * tt + l15) * NW * 8);
      const f32x4 a = sp[0], b = sp[1];
      const float m1 = ((a[0] + a[2]) + (b[0] + b[2])) * (1.0f / FS_C), m2 = ((a[1] + a[3]) + (b[1] + b[3])) * (1.0f / FS_C);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 xv = unpack4(xh[rt][tt]);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = g[rt][tt][e] + rstd[tt] * (acc[rt][tt][e] - m1 - xv[e] * m2);
        const u32x2 u = pack4(lv[tt] ? drop4(d, A.seed_out ^ smix, off[tt] + 16 * rt) : f32x4{0.f, 0.f, 0.f, 0.f});
        *(u32x2*)(imgA + tt * 8192 + wro[rt]) = u;      // image A's last readers (P1's GEMM) passed two barriers ago
        if (lv[tt]) {
          *(f32x4*)(A.dx1 + off[tt] + 16 * rt) = d;
          *(u32x2*)(A.dy1 + off[tt] + 16 * rt) = u;
        }
      }
    }
  }
  __syncthreads();

  // ---- P3: do = Wo^T dy1 -> memory ---------------------------------------------------------------------------------------------------------
  {
    f32x4 acc[RT][NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    fs_slice_gemm<2, 24, NTT, RT, false, PF>(wq, wb, imgA, rdo, acc);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
      if (lv[tt]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) *(u32x2*)(A.d_o + off[tt] + 16 * rt) = pack4(acc[rt][tt]);
      }
  }
}

// ---- the front of the block's backward in ONE launch: q | k | v data gradient + LayerNorm1 backward + the skip gradient -----------------
//
//     dxh = dqkv W_in'        (M x 768 . 768 x 256; W_in' = the LayerNorm-folded in-projection weight)
//     dx  = dx1 + rstd1 (dxh - mean(dxh) - xh1 mean(dxh xh1))          (attn_backbone.py:79-80 backwards)
//
// Unfused: a 768-deep data-gradient GEMM (18 - 20 us) that writes dxh (bf16) and a LayerNorm backward (16 us) that reads it back with the
// fp32 skip gradient.  Same scheme as the tail kernel: 4 waves, a wave owns 64 of the 256 output features for the workgroup's 48 / 64
// tokens; the B operand is the workgroup's dqkv rows as THREE bf16 images (q, k, v parts) filled by LDS-DMA -- it writes lane-linear, so the
// images' row swizzle is applied to the global SOURCE chunk each lane fetches (chunk c of row r lives at c ^ (r & 15): the four lanes of a
// quad stay inside one 64-byte piece) --, the weights are the three 256 x 256 row blocks of W_in' transposed, i.e. exactly the tail
// kernel's fragment stream (tante_pack_block_tail_bwd on the three row blocks).  The fp32 skip gradient and the result move in row form
// through a private piece of an image once the GEMM's last reader is behind the statistics barrier.
struct BhArgs {
  const unsigned short *dqkv, *xh1;
  const float *st1, *dx1;
  const char* w;
  float* dx;
  long M;
};

template <int NTT>
__global__ __launch_bounds__(256, 2) void block_head_bwd_kernel(BhArgs A) {
#ifdef BT_PRIO
  if ((__builtin_amdgcn_s_getreg(0x1804) & 1u) == 1u) __builtin_amdgcn_s_setprio(3);
#endif
  constexpr int RT = 4, NW = 4, PF = 2;
  constexpr int IMG = 16 * NTT * FS_ROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const stat = smem + 3 * IMG;     // float2 [16 NTT tokens][4 waves]
  const int tid = threadIdx.x, lane = tid & 63, kk = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long tok0 = (long)blockIdx.x * (16 * NTT);
  int rdo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rdo[j] = l15 * FS_ROW + ((((j ^ (l15 >> 2)) << 2) | (kk ^ (l15 & 3))) << 4);
  const char* const wq = A.w + (RT * wave) * FS_FRAG + lane * 16;
  u32x4 wb[PF + 1][RT];
  fs_wring_prime<0, RT, PF>(wq, wb);
  // ---- dqkv rows -> the three images: one instruction = 2 rows of one image (32 lanes x 16 B per 512-byte row) -------------------------
  {
    const int half = lane >> 5, s = lane & 31;           // row within the pair, physical 16-byte slot within the row
    for (int q = wave; q < 3 * 8 * NTT; q += NW) {       // 8 NTT row pairs per image
      const int m = q / (8 * NTT), rp = q - m * (8 * NTT), r = 2 * rp + half;
      long t = tok0 + r;
      if (t >= A.M) t = A.M - 1;                         // dead rows: any valid row (their results are never stored)
      const int c = (s & 16) | ((s & 15) ^ (r & 15));    // the logical chunk that belongs in this slot
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A.dqkv + t * (3 * FS_C) + m * FS_C + 8 * c),
                                       (__attribute__((address_space(3))) void*)(smem + m * IMG + rp * 1024), 16, 0, 0);
    }
  }
  const int col0 = 16 * RT * wave + 4 * kk;
  bool lv[NTT];
  long off[NTT];
  float rstd[NTT];
  u32x2 xh[RT][NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const long t = tok0 + 16 * tt + l15;
    lv[tt] = t < A.M;
    off[tt] = (lv[tt] ? t : 0) * FS_C + col0;
    rstd[tt] = A.st1[2 * (lv[tt] ? t : 0) + 1];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) xh[rt][tt] = *(const u32x2*)(A.xh1 + off[tt] + 16 * rt);      // in flight under the GEMM
  }
  // the skip gradient, row form (16 lanes x 16 B per token row of this wave's 64-feature slice), in flight under the GEMM too
  constexpr int CPR = 4 * RT, RPI = 64 / CPR, ROWB = CPR * 16, TILEB = 16 * ROWB;
  static_assert(IMG / NW >= TILEB, "staging piece too small");
  const int rrow = lane / CPR, rchunk = lane % CPR;
  f32x4 graw[NTT][RT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const long t = tok0 + 16 * tt + RPI * j + rrow;
      graw[tt][j] = *(const f32x4*)(A.dx1 + (t < A.M ? t : 0) * FS_C + 16 * RT * wave + 4 * rchunk);
    }
  // the images must be complete before the GEMM; the row loads above were requested right behind the DMA passes and land with them
  // (a counted wait would have to rely on the compiler keeping them behind the DMA instructions)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc[RT][NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  fs_slice_gemm<0, 24, NTT, RT, false, PF>(wq, wb, smem, rdo, acc);
  fs_slice_gemm<1, 24, NTT, RT, false, PF>(wq, wb, smem + IMG, rdo, acc);
  fs_slice_gemm<2, 24, NTT, RT, false, PF>(wq, wb, smem + 2 * IMG, rdo, acc);
  // per-token sums over ALL 256 features: this wave's 64, then the four waves' partials through LDS
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 xv = unpack4(xh[rt][tt]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1 += acc[rt][tt][e];
        s2 = fmaf(acc[rt][tt][e], xv[e], s2);
      }
    }
    s1 = rows_sum(s1);
    s2 = rows_sum(s2);
    if (kk == 0) *(float2*)(stat + ((16 * tt + l15) * NW + wave) * 8) = make_float2(s1, s2);
  }
  __syncthreads();
  char* const sb = smem + wave * (IMG / NW);      // the images' last readers are behind the barrier: a private staging piece
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const f32x4* sp = (const f32x4*)(stat + (16 * tt + l15) * NW * 8);
    const f32x4 a = sp[0], b = sp[1];
    const float m1 = ((a[0] + a[2]) + (b[0] + b[2])) * (1.0f / FS_C), m2 = ((a[1] + a[3]) + (b[1] + b[3])) * (1.0f / FS_C);
    // skip gradient of this tile: row form -> accumulator layout
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int r = RPI * j + rrow;
      *(f32x4*)(sb + r * ROWB + ((rchunk ^ (r & (CPR - 1))) << 4)) = graw[tt][j];
    }
    f32x4 d[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 g = *(const f32x4*)(sb + l15 * ROWB + (((4 * rt + kk) ^ (l15 & (CPR - 1))) << 4));
      const f32x4 xv = unpack4(xh[rt][tt]);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[rt][e] = g[e] + rstd[tt] * (acc[rt][tt][e] - m1 - xv[e] * m2);
    }
    // and the result back: accumulator layout -> row form -> memory
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *(f32x4*)(sb + l15 * ROWB + (((4 * rt + kk) ^ (l15 & (CPR - 1))) << 4)) = d[rt];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int r = RPI * j + rrow;
      const long t = tok0 + 16 * tt + r;
      const f32x4 v = *(const f32x4*)(sb + r * ROWB + ((rchunk ^ (r & (CPR - 1))) << 4));
#ifdef BH_PLAIN_STORE
      if (t < A.M) *(f32x4*)(A.dx + t * FS_C + 16 * RT * wave + 4 * rchunk) = v;
#else   // write-through (common.hip.h): the launch's only output, written at its very end
      if (t < A.M) st_wt16(A.dx + t * FS_C + 16 * RT * wave + 4 * rchunk, v);
#endif
    }
  }
}

// fragment f = (m * 8 + ks) * 16 + g of the TRANSPOSED matrix m: lane (l15, kk) holds  W_m^T[16 g + l15][32 ks + 8 kk + e] =
// W_m[32 ks + 8 kk + e][16 g + l15];  m = 0: fc2 weight, 1: the folded fc1 weight (W1 diag(gamma2)), 2: out-proj weight
__device__ __forceinline__ void bt_pack_body(int bid, const float* __restrict__ w2, const float* __restrict__ w1f, const float* __restrict__ wo,
                                             char* __restrict__ dst) {
  const int f = bid * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, l15 = lane & 15, kk = lane >> 4;
  const int g = f % 16, ks = (f / 16) % 8, m = f / 128;
  const float* src = m == 0 ? w2 : (m == 1 ? w1f : wo);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = src[(long)(32 * ks + 8 * kk + e) * FS_C + 16 * g + l15];
  u32x4 o;
  o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
  *(u32x4*)(dst + (long)f * FS_FRAG + lane * 16) = o;
}
__global__ void bt_pack_kernel(const float* __restrict__ w2, const float* __restrict__ w1f, const float* __restrict__ wo, char* __restrict__ dst) {
  bt_pack_body((int)blockIdx.x, w2, w1f, wo, dst);
}
// the transposed-fragment streams of every block in one launch (tante_pack_block_tail_bwd_multi): two per block and train step otherwise
constexpr int BTP_MAX = 24, BTP_BLOCKS = 3 * 128 / 4;
struct BtPackBatch {
  const float* a[BTP_MAX]; const float* b[BTP_MAX]; const float* c[BTP_MAX];
  char* dst[BTP_MAX];
};
__global__ void bt_pack_multi_kernel(BtPackBatch B) {
  const int e = blockIdx.x / BTP_BLOCKS;
  bt_pack_body((int)blockIdx.x - e * BTP_BLOCKS, B.a[e], B.b[e], B.c[e], B.dst[e]);
}

template <int NTT>
void bt_launch(const BtArgs& A, hipStream_t s) {
  constexpr int LDS = 2 * 16 * NTT * FS_ROW + 16 * NTT * 4 * 8;
  static TantePerDevice attr;
  attr.once([&] { (void)hipFuncSetAttribute((const void*)block_tail_bwd_kernel<NTT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
  const long per = 16 * NTT;
  hipLaunchKernelGGL((block_tail_bwd_kernel<NTT>), dim3((unsigned)((A.M + per - 1) / per)), dim3(256), LDS, s, A);
}

template <int NTT>
void bh_launch(const BhArgs& A, hipStream_t s) {
  constexpr int LDS = 3 * 16 * NTT * FS_ROW + 16 * NTT * 4 * 8;
  static TantePerDevice attr;
  attr.once([&] { (void)hipFuncSetAttribute((const void*)block_head_bwd_kernel<NTT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
  const long per = 16 * NTT;
  hipLaunchKernelGGL((block_head_bwd_kernel<NTT>), dim3((unsigned)((A.M + per - 1) / per)), dim3(256), LDS, s, A);
}

}  // namespace

extern "C" int tante_block_head_bwd(const void* dqkv, const void* xh1, const float* st1, const float* dx1, const void* head_bwd_stream, int64_t M,
                                    int C, float* dx, void* stream) {
  if (!dqkv || !xh1 || !st1 || !dx1 || !head_bwd_stream || !dx || M <= 0) TANTE_FAIL(-1, "tante_block_head_bwd: bad argument");
  if (C != FS_C) TANTE_FAIL(-2, "tante_block_head_bwd: unsupported C=%d", C);
  if ((((uintptr_t)dqkv | (uintptr_t)xh1 | (uintptr_t)dx1 | (uintptr_t)dx | (uintptr_t)head_bwd_stream) & 15) || ((uintptr_t)st1 & 7))
    TANTE_FAIL(-1, "tante_block_head_bwd: 16-byte alignment");
  BhArgs A;
  A.dqkv = (const unsigned short*)dqkv; A.xh1 = (const unsigned short*)xh1; A.st1 = st1; A.dx1 = dx1; A.w = (const char*)head_bwd_stream;
  A.dx = dx; A.M = M;
  const long w3 = (M + 47) / 48, w4 = (M + 63) / 64;      // 3 images of 24 / 32 KiB: two 48-token workgroups per CU, one 64-token one
  const long r3 = ((w3 + 511) / 512) * 3, r4 = ((w4 + 255) / 256) * 4;
  if (r3 <= r4) bh_launch<3>(A, (hipStream_t)stream);
  else bh_launch<4>(A, (hipStream_t)stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t tante_block_tail_bwd_stream_bytes(int C, int hidden) { return (C == FS_C && hidden == FS_C) ? BT_W_BYTES : 0; }

extern "C" int tante_pack_block_tail_bwd(const float* fc2_w, const float* fc1_w_folded, const float* out_w, int C, int hidden, void* bwd_stream,
                                         void* stream) {
  if (!fc2_w || !fc1_w_folded || !out_w || !bwd_stream) TANTE_FAIL(-1, "tante_pack_block_tail_bwd: null pointer");
  if (C != FS_C || hidden != FS_C) TANTE_FAIL(-2, "tante_pack_block_tail_bwd: unsupported C=%d hidden=%d", C, hidden);
  hipLaunchKernelGGL(bt_pack_kernel, dim3(3 * 128 / 4), dim3(256), 0, (hipStream_t)stream, fc2_w, fc1_w_folded, out_w, (char*)bwd_stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_pack_block_tail_bwd_multi(const TanteMat3* mats, int n, int C, int hidden, void* stream) {
  if (!mats || n <= 0) TANTE_FAIL(-1, "tante_pack_block_tail_bwd_multi: bad argument");
  if (C != FS_C || hidden != FS_C) TANTE_FAIL(-2, "tante_pack_block_tail_bwd_multi: unsupported C=%d hidden=%d", C, hidden);
  for (int g = 0; g < n; g += BTP_MAX) {
    BtPackBatch B;
    const int m = n - g < BTP_MAX ? n - g : BTP_MAX;
    for (int e = 0; e < m; ++e) {
      const TanteMat3& t = mats[g + e];
      if (!t.a || !t.b || !t.c || !t.dst) TANTE_FAIL(-1, "tante_pack_block_tail_bwd_multi: null pointer in entry %d", g + e);
      B.a[e] = t.a; B.b[e] = t.b; B.c[e] = t.c; B.dst[e] = (char*)t.dst;
    }
    hipLaunchKernelGGL(bt_pack_multi_kernel, dim3((unsigned)(m * BTP_BLOCKS)), dim3(256), 0, (hipStream_t)stream, B);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_block_tail_bwd(const float* dout, const void* hpre, const void* xh2, const float* st2, const void* bwd_stream, int64_t M,
                                    int C, int hidden, float p_drop, uint64_t seed_out, uint64_t seed_mlp, float* dx1, void* dy2, void* dhpre,
                                    void* dy1, void* d_o, void* stream) {
  if (!dout || !hpre || !xh2 || !st2 || !bwd_stream || !dx1 || !dy2 || !dhpre || !dy1 || !d_o || M <= 0)
    TANTE_FAIL(-1, "tante_block_tail_bwd: bad argument");
  if (C != FS_C || hidden != FS_C) TANTE_FAIL(-2, "tante_block_tail_bwd: unsupported C=%d hidden=%d", C, hidden);
  if (p_drop < 0.0f || p_drop >= 1.0f) TANTE_FAIL(-1, "tante_block_tail_bwd: dropout probability %f", (double)p_drop);
  BtArgs A;
  A.dout = dout; A.hpre = (const unsigned short*)hpre; A.xh2 = (const unsigned short*)xh2; A.st2 = st2; A.w = (const char*)bwd_stream;
  A.dx1 = dx1; A.dy2 = (unsigned short*)dy2; A.dhpre = (unsigned short*)dhpre; A.dy1 = (unsigned short*)dy1; A.d_o = (unsigned short*)d_o;
  A.M = M; A.p = p_drop; A.seed_out = seed_out; A.seed_mlp = seed_mlp; A.seed_mix = tante_seed_mix_ptr();
  // 48-token workgroups when that fills the chip's 512 resident slots better (cfg3: 24 576 tokens = 512 x 48), else 64
  const long w3 = (M + 47) / 48, w4 = (M + 63) / 64;
  const long r3 = ((w3 + 511) / 512) * 3, r4 = ((w4 + 511) / 512) * 4;
  if (r3 < r4) bt_launch<3>(A, (hipStream_t)stream);
  else bt_launch<4>(A, (hipStream_t)stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}
