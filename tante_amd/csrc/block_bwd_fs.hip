// The WHOLE backward of a TransformerBlock in ONE launch (bf16 training path, C = 256 / 8 heads x 32 / hidden 256; attn_backbone.py:59-83
// backwards, trainer/trainer.py:191):
//
//     out = x1 + drop_mlp( W2 gelu_tanh(hpre) + b2 ),  hpre = W1' LayerNorm2(x1) + b1',  x1 = x + drop_out( Wo o + bo ),
//     o = Attention( q | k | v ),  q | k | v = W_in' LayerNorm1(x) + b_in'
//
//     dy2   = mask_mlp . dout / (1 - p)                                  (U operand of fc2's weight gradient)
//     dhpre = (W2^T dy2) . gelu_tanh'(hpre)                              (U operand of fc1's weight gradient)
//     dx1   = dout + rstd2 (dxh2 - mean(dxh2) - xh2 mean(dxh2 xh2)),  dxh2 = W1'^T dhpre
//     dy1   = mask_out . dx1 / (1 - p)                                   (U operand of the out-proj weight gradient)
//     do    = Wo^T dy1
//     dq | dk | dv = AttentionBackward(q, k, v, do)                       (U operand of the in-projection's weight gradient)
//     dx    = dx1 + rstd1 (dxh1 - mean(dxh1) - xh1 mean(dxh1 xh1)),  dxh1 = [dq | dk | dv] W_in'
//
// Round 4 ran this as three launches (block_tail_bwd -> attn_bwd_* -> block_head_bwd, block_bwd.hip / backward.hip) with every hand-off
// through memory: d_o (bf16) and dx1 (fp32) written by the first and read straight back, dq | dk | dv written by the second and read by
// the third, q | k | v saved by the training forward for the second alone -- 110-160 MB of the 325 MB a block's backward moved, 36 times
// per train step.  Here a workgroup owns WHOLE SEQUENCES (48 tokens at cfg3: 12 x T, 3 x H, 1 x W), the same partition as the forward
// kernel (block_sliced.hip), so the attention's backward sits between the two token-wise halves with nothing but LDS and registers in
// between:
//   * dx1 stays in the accumulator registers that held dout (fp32, 48 per lane) from LayerNorm2's backward to the final add;
//   * d_o leaves the out-proj data-gradient GEMM as accumulators = the attention backward's operand fragments;
//   * q, k, v are RECOMPUTED from LayerNorm1's saved image (three more 256 x 256 slices on the matrix pipe, which this launch does not
//     saturate) with the forward kernel's own weight stream: bit-identical to what the forward's attention multiplied, and the training
//     forward no longer stores the packed projection (37.8 MB per block call at cfg3) for anyone;
//   * dq | dk | dv go from the attention's accumulators into the three LDS images the closing GEMM reads, and to memory once (row form).
// Per workgroup: 4 waves, wave w = output features 64 w .. 64 w + 63 of every GEMM = heads 2 w, 2 w + 1 of the attention; three bf16
// activation images [token][256] (XOR-swizzled 16-byte chunks); weights stream L2 -> registers as pre-packed fragments (fs_common.hip.h).
//
// Attention backward, per head, on the forward kernel's operand scheme (head dim 32 = one MFMA k-step):
//   pass 1 (queries in the accumulator columns):  S^T = K Q^T, dP^T = V dO^T -> row statistics (max, 1/l, delta) and dS^T -> dQ^T = K^T dS^T
//   pass 2 (keys in the accumulator columns):     S = Q K^T, dP = dO V^T     -> dS, dropped P                          -> dK^T, dV^T
// each score tile is computed in both orientations instead of being transposed; the transposed operands (K^T, Q^T, dO^T: rows = head dims,
// k = tokens) are ds_read_b64_tr_b16 reads of the row-major images the wave wrote its q, k, d_o slices into (its own columns: no barrier).
// The softmax runs in the forward's log2 domain: the forward stream's Wq carries log2(e) / sqrt(32), so dq takes 1 / sqrt(32) and dk takes
// ln 2 (dk = scale dS^T q = ln 2 dS^T q').
#include "common.hip.h"
#include "fused_common.hip.h"
#include "fs_common.hip.h"
#include "block_sliced.h"

// Structure switches (each pair computes the same function; tools/build_variant.sh builds the other forms, tools/ab_train.sh times them on
// the cfg3 train step on ONE box).  Measured, round 5, B = 8, ms per train step, two interleaved rounds (gpurun_out/r05_ab_train_1.log ->
// profiles/r05_block_bwd_ab.log): three launches 9.45 / 9.45; DMA rows + burst stores + dx1 in registers 9.34 / 9.40 (the default);
// everything off (the first form of the round) 9.39 / 9.40; + stores riding the GEMMs 9.38 / 9.48; + dx1 parked in memory 9.71 / 9.67.
#ifndef BF_DMA_ROWS
#define BF_DMA_ROWS 1      // 1: hpre / xh2 / xh1 rows take turns in image C by LDS-DMA; 0: accumulator-layout global loads (xh1's image by DMA at the start)
#endif
#ifndef BF_HOOK_STORES
#define BF_HOOK_STORES 0   // 1: bf16 row stores ride the k-steps of the following GEMM; 0: issued together behind the phase that wrote the image
#endif
#ifndef BF_LATE_HPRE
#define BF_LATE_HPRE 1     // 1: the first saved rows (fc1's pre-activation) are requested behind the arrival of the gradient rows; 0: with them
#endif
#ifndef BF_PARK_DX1
#define BF_PARK_DX1 0      // 1: dx1 waits in the output buffer between LayerNorm2's backward and the end (+50 MB per launch); 0: in 48 registers
#endif

namespace {

struct BfArgs {
  const float* dout;
  const unsigned short *xh1, *hpre, *xh2;
  const float *st1, *st2;
  const char *wt, *wf, *wh;      // tail stream (W2^T | W1'^T | Wo^T), forward stream (q | k | v | ... + biases), head stream (Wq'^T | Wk'^T | Wv'^T)
  float* dx;
  unsigned short *dy2, *dhpre, *dy1, *dqkv;
  TanteSeq sq;
  int causal, spw;
  unsigned magic;                // ceil(65536 / L)
  float p;
  unsigned long long seed_attn, seed_out, seed_mlp;
  const unsigned long long* seed_mix;
  unsigned long long* stamps;    // diagnostic builds (-DTANTE_ABLATE) only, else null
};

// d/dx [ x Phi_tanh(x) ]  (block_bwd.hip's closed form: one exp2, one rcp)
__device__ __forceinline__ float bf_gelu_tanh_grad(float x) {
  const float x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(2.30220805f * x * fmaf(0.044715f, x2, 1.0f));
  const float r = __builtin_amdgcn_rcpf(1.0f + e);
  const float s = 1.0f - r;
  return s * fmaf(1.59576912f * x * r, fmaf(0.134145f, x2, 1.0f), 1.0f);
}
__device__ __forceinline__ u32x2 bf_pack4(const f32x4& v) {
  u32x2 u;
  u[0] = pack_bf16x2(v[0], v[1]);
  u[1] = pack_bf16x2(v[2], v[3]);
  return u;
}
__device__ __forceinline__ f32x4 bf_unpack4(const u32x2& u) { return f32x4{bf16_lo(u[0]), bf16_hi(u[0]), bf16_lo(u[1]), bf16_hi(u[1])}; }
__device__ __forceinline__ u32x2 bf_tr(unsigned addr) {      // 4 rows x 16 columns of a row-major bf16 tile, column-major out (one 16-lane group)
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
__device__ __forceinline__ u32x4 bf_lds128(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}

// In-kernel stamps (cdna_hip_programming.md 7): a -DTANTE_ABLATE build records the shader clock at the phase boundaries of every wave into a
// buffer of its own (tools/bf_stamps.py reads the phase shares from it); in the product build the macro is empty.
#ifdef TANTE_ABLATE
#define BF_STAMP(k)                                                                                        \
  do {                                                                                                     \
    if (A.stamps) {                                                                                        \
      unsigned long long t_;                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (lane == 0) A.stamps[((long)blockIdx.x * 4 + wave) * 24 + (k)] = t_;                            \
    }                                                                                                      \
  } while (0)
unsigned long long* g_bf_stamps = nullptr;
#else
#define BF_STAMP(k)
#endif

struct BfSeqMap {
  float rs0, rl0;
  __device__ __forceinline__ explicit BfSeqMap(const TanteSeq& q) : rs0(__builtin_amdgcn_rcpf((float)q.n_s0)), rl0(__builtin_amdgcn_rcpf((float)q.n_l0)) {}
  static __device__ __forceinline__ void divmod(int a, int n, float rn, int& q, int& r) {
    q = (int)((float)a * rn);
    r = a - q * n;
    if (r >= n) { ++q; r -= n; }
    if (r < 0) { --q; r += n; }
  }
  __device__ __forceinline__ long token(const TanteSeq& q, int s, int l) const {
    int a, b, c, d;
    divmod(s, q.n_s0, rs0, a, b);
    divmod(l, q.n_l0, rl0, c, d);
    return (long)a * q.S1 + (long)b * q.S0 + (long)c * q.P1 + (long)d * q.P0;
  }
};

// TPS = token tiles per sequence (L = 16 TPS), 1 also for L | 16 (several sequences per tile, block-diagonal mask).  NTT = token tiles per
// workgroup (a multiple of TPS).  Two workgroups per CU at NTT <= 3 (3 x 24 KiB of images each).
// DROP: dropout compiled in (p > 0) or out -- the mask arithmetic is a third of the attention phase's instructions
template <int TPS, int NTT, bool DROP>
__global__ __launch_bounds__(256, NTT <= 3 ? 2 : 1) void block_bwd_fs_kernel(BfArgs A) {
#ifdef BT_PRIO
  if ((__builtin_amdgcn_s_getreg(0x1804) & 1u) == 1u) __builtin_amdgcn_s_setprio(3);      // as FS_PRIO in block_sliced.hip: the odd hardware wave slots
#endif
  static_assert(NTT % TPS == 0 && 16 * NTT <= 64, "whole sequences, one slot per lane");
  constexpr int RT = 4, NW = 4, PF = 2, HPW = 2, NK = TPS;
  constexpr int IMG = 16 * NTT * FS_ROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const imgA = smem;                 // dy2 -> dy1 -> q -> dk
  char* const imgB = smem + IMG;           // dhpre -> d_o -> dv
  char* const imgC = smem + 2 * IMG;       // LayerNorm1's rows (xh1) -> k -> dq
  char* const stat = smem + 3 * IMG;       // float2 [16 NTT tokens][4 waves]
  float* const lbias = (float*)(stat + 16 * NTT * NW * 8);      // q (scaled) | k | v biases of the forward stream
  const int tid = threadIdx.x, lane = tid & 63, kk = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned long long smix = A.seed_mix ? *A.seed_mix : 0ull;
  const int L = A.sq.L;
  const int seq0 = (int)blockIdx.x * A.spw;
  const int nlive = min(A.spw, A.sq.nseq - seq0) * L;
  const float ksc = DROP ? 1.0f / (1.0f - A.p) : 1.0f;

  const FsW wqt = fs_wstream(A.wt, (unsigned)((RT * wave) * FS_FRAG + lane * 16));
  const FsW wqf = fs_wstream(A.wf, (unsigned)((RT * wave) * FS_FRAG + lane * 16));
  const FsW wqh = fs_wstream(A.wh, (unsigned)((RT * wave) * FS_FRAG + lane * 16));
  u32x4 wb[PF + 1][RT];
  BF_STAMP(0);
#ifdef TANTE_ABLATE
  if (A.stamps && lane == 0) {
    A.stamps[((long)blockIdx.x * 4 + wave) * 24 + 20] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    A.stamps[((long)blockIdx.x * 4 + wave) * 24 + 21] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    A.stamps[((long)blockIdx.x * 4 + wave) * 24 + 22] = __builtin_amdgcn_s_memrealtime();
  }
#endif

  // ---- slot -> token (-1 = dead) ---------------------------------------------------------------------------------------------------------
  int tslot;
  {
    const BfSeqMap smap(A.sq);
    const int s = (int)(((unsigned)lane * A.magic) >> 16), p = lane - s * L;
    tslot = lane < nlive ? (int)smap.token(A.sq, seq0 + s, p) : -1;
  }
  // (every call is a fresh ds_bpermute of a laundered copy: left to common-subexpression elimination, the compiler computes the ~30 row
  // tokens of a lane once, keeps them and their shifted forms alive across the launch and SPILLS them -- and a scratch reload is a
  // vector-memory wait in a queue that returns in order: each one drained every row store and DMA in flight)
  auto tok_of = [&](int slot) {
    int ts = tslot;
    asm volatile("" : "+v"(ts));
    return __shfl(ts, slot & 63);
  };
  int tokidx[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) tokidx[tt] = tok_of(16 * tt + l15);

  // q | k | v biases: floats [0, 256) and [1024, 1536) of the forward stream's bias block, 192 x 16 bytes; requested first (the wave's memory
  // queue returns in order: behind the DMA below this load would wait for cold rows), written to LDS behind P0
  f32x4 bias_in = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 3 * FS_C / 4) bias_in = *(const f32x4*)((const float*)(A.wf + FS_W_BYTES) + 4 * (tid < 64 ? tid : tid + 192));

  // ---- saved bf16 rows -> image C by LDS-DMA: fc1's pre-activation (P1's epilogue), then LayerNorm2's image (P2's epilogue), then
  // LayerNorm1's image (the q | k | v recompute and LayerNorm1's backward) take turns in it; every wave fetches a quarter of the rows and a
  // barrier behind its s_waitcnt vmcnt(0) publishes them.  One instruction = 2 rows of the image (32 lanes x 16 B per 512-byte row); it
  // writes lane-linear, so the image's row swizzle is applied to the global SOURCE chunk each lane fetches (chunk c of row r lives at
  // c ^ (r & 15)).  (Round 5, first form: these rows were loaded in ACCUMULATOR layout -- 8 bytes per lane, 16 different rows per
  // instruction, every 64-byte line fetched twice -- and cost the launch 13-22 us: tools/build_variant.sh + -DBF_EXP_NO_ACCLOADS.)
  auto fresh_lane = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    return ln;
  };
  auto dma_rows = [&](const unsigned short* __restrict__ src) {
    const int ln_ = fresh_lane(), half = ln_ >> 5, sl = ln_ & 31;
#pragma unroll
    for (int i = 0; i < 2 * NTT; ++i) {
      const int rp = wave + NW * i, r = 2 * rp + half;
      const int t = tok_of(r);
      const int c = (sl & 16) | ((sl & 15) ^ (r & 15));
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)(t < 0 ? 0 : t) * FS_C + 8 * c),
                                       (__attribute__((address_space(3))) void*)(imgC + rp * 1024), 16, 0, 0);
    }
  };
  asm volatile("" ::: "memory");
  fs_wring_prime<0, RT, PF>(wqt, wb);
  BF_STAMP(1);

  // ---- fp32 rows between memory (row form: 16 lanes x 16 B per token row of this wave's 64 features) and the accumulator layout ----------
  constexpr int CPR = 4 * RT, RPI = 64 / CPR, ROWB = CPR * 16, TILEB = 16 * ROWB;
  static_assert(IMG / NW >= TILEB, "staging piece too small");
  // Every helper below derives its row / chunk indices from a LAUNDERED copy of the lane id: left alone, loop-invariant code motion computes
  // the ~40 per-lane addresses of all their call sites once, keeps them alive across the launch and spills them -- and a scratch reload is a
  // vector-memory wait in a queue that returns in order (it drains every row store and DMA in flight).  Recomputing them is a few VALU each.
#define BF_RLANE const int ln_ = fresh_lane(), rrow = ln_ / CPR, rchunk = ln_ % CPR
#define BF_BLANE const int ln_ = fresh_lane(), brow = ln_ / CPB, bchunk = ln_ % CPB
  auto slice_load = [&](const float* __restrict__ src, f32x4 (&raw)[NTT][RT]) {
    BF_RLANE;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int t = tok_of(16 * tt + RPI * j + rrow);
        raw[tt][j] = *(const f32x4*)(src + (long)(t < 0 ? 0 : t) * FS_C + 16 * RT * wave + 4 * rchunk);
      }
  };
  auto slice_to_acc = [&](char* stg, const f32x4 (&raw)[NTT][RT], f32x4 (&acc)[RT][NTT]) {      // stg: a private IMG / NW piece of a free image
    BF_RLANE;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int r = RPI * j + rrow;
        *(f32x4*)(stg + r * ROWB + ((rchunk ^ (r & (CPR - 1))) << 4)) = raw[tt][j];
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][tt] = *(const f32x4*)(stg + l15 * ROWB + (((4 * rt + kk) ^ (l15 & (CPR - 1))) << 4));
    }
  };
  auto slice_store = [&](char* stg, float* __restrict__ dst, const f32x4 (&acc)[RT][NTT], auto wt_c) {
    BF_RLANE;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) *(f32x4*)(stg + l15 * ROWB + (((4 * rt + kk) ^ (l15 & (CPR - 1))) << 4)) = acc[rt][tt];
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int r = RPI * j + rrow, t = tok_of(16 * tt + r);
        const f32x4 v = *(const f32x4*)(stg + r * ROWB + ((rchunk ^ (r & (CPR - 1))) << 4));
        if (t >= 0) {
          if constexpr (decltype(wt_c)::value) st_wt16(dst + (long)t * FS_C + 16 * RT * wave + 4 * rchunk, v);      // the launch's last output, at its very end
          else *(f32x4*)(dst + (long)t * FS_C + 16 * RT * wave + 4 * rchunk) = v;
        }
      }
    }
  };
  // bf16 rows: this wave's columns of image tile tt -> memory in row form (8 lanes x 16 B per token row, 8 rows per instruction)
  constexpr int CPB = 2 * RT, RPB = 64 / CPB;
  // one sub-step = one instruction = 8 rows; a tensor's 2 NTT sub-steps ride the k-steps of the GEMM that FOLLOWS the phase that wrote the
  // image (rows_hook): issued together behind their phase, the 36 row stores of a wave (100 MB per launch at cfg3) formed bursts that the
  // next GEMM's weight loads queued behind (the wave's memory queue returns in order) -- 15-20 us of the launch (-DBF_EXP_NO_ROWSTORES)
  auto img_rows_store1 = [&](const char* img, auto tt_c, auto j_c, unsigned short* __restrict__ dst, long dstride, int dcol) {
    constexpr int tt = decltype(tt_c)::value, j = decltype(j_c)::value;
    BF_BLANE;
    const int r = RPB * j + brow;
    const int t = tok_of(16 * tt + r);
    const u32x4 v = *(const u32x4*)(img + tt * 8192 + r * FS_ROW + (((CPB * wave + bchunk) ^ r) << 4));
#ifdef BF_EXP_NO_ROWSTORES      // timing experiment only (wrong results): what the bf16 row stores cost
    if (t >= 0 && v[0] == 0x12345u) *(u32x4*)(dst + (long)t * dstride + dcol + 16 * RT * wave + 8 * bchunk) = v;
#else
    if (t >= 0) *(u32x4*)(dst + (long)t * dstride + dcol + 16 * RT * wave + 8 * bchunk) = v;
#endif
  };
  auto rows_hook = [&](const char* img, unsigned short* __restrict__ dst, long dstride, int dcol) {
    return [&, img, dst, dstride, dcol](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
      if constexpr (ks < 2 * NTT) img_rows_store1(img, std::integral_constant<int, ks / 2>{}, std::integral_constant<int, ks % 2>{}, dst, dstride, dcol);
    };
  };
  static_assert(16 / RPB == 2 && 2 * NTT <= 8, "a tensor's row stores fit the k-steps of one GEMM");
  auto rows_burst = [&](const char* img, unsigned short* __restrict__ dst, long dstride, int dcol) {      // BF_HOOK_STORES == 0: all at once
    static_for<2 * NTT>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      img_rows_store1(img, std::integral_constant<int, i / 2>{}, std::integral_constant<int, i % 2>{}, dst, dstride, dcol);
    });
  };
#if BF_HOOK_STORES
#define BF_HOOK(img, dst, stride, col) rows_hook(img, dst, stride, col)
#define BF_BURST(img, dst, stride, col)
#else
#define BF_HOOK(img, dst, stride, col) FsNoHook{}
#define BF_BURST(img, dst, stride, col) rows_burst(img, dst, stride, col)
#endif

  // ---- per-lane LDS addressing (block_sliced.hip) ----------------------------------------------------------------------------------------
  int rdo[4], wro[RT];
#pragma unroll
  for (int j = 0; j < 4; ++j) rdo[j] = l15 * FS_ROW + ((((j ^ (l15 >> 2)) << 2) | (kk ^ (l15 & 3))) << 4);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) wro[rt] = l15 * FS_ROW + (((2 * RT * wave + 2 * rt + (kk >> 1)) ^ l15) << 4) + (kk & 1) * 8;
  const int col0 = 16 * RT * wave + 4 * kk;      // this lane's first column; row tile rt adds 16 rt
  bool lv[NTT];
  long off[NTT];                                 // element offset of (token, col0) in a (tokens, 256) tensor
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    lv[tt] = tokidx[tt] >= 0;
    off[tt] = (long)(lv[tt] ? tokidx[tt] : 0) * FS_C + col0;
  }
  auto drop4 = [&](const f32x4& v, unsigned long long seed, long i0) {
    if constexpr (!DROP) return v;
    // (laundered: the mask depends on indices only, and left alone the compiler hashes LayerNorm2's masks in P0, two barriers early, and
    // spills them -- every scratch reload then drains the wave's memory queue)
    asm volatile("" : "+v"(i0));
    const unsigned k01 = dropout_keep2(seed, (unsigned long long)i0, A.p), k23 = dropout_keep2(seed, (unsigned long long)i0 + 2, A.p);
    return f32x4{(k01 & 1u) ? v[0] * ksc : 0.0f, (k01 & 2u) ? v[1] * ksc : 0.0f, (k23 & 1u) ? v[2] * ksc : 0.0f, (k23 & 2u) ? v[3] * ksc : 0.0f};
  };
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

  // ================================ P0: gradient rows in; dropped branch gradient -> image A and memory ==================================
  f32x4 g[RT][NTT];      // dout, later dx1
  float rstd2[NTT];
  {
    f32x4 raw[NTT][RT];
    slice_load(A.dout, raw);      // the rows everything waits for: first in the memory queue
    asm volatile("" ::: "memory");
#if !BF_LATE_HPRE
#if BF_DMA_ROWS
    dma_rows(A.hpre);
#else
    dma_rows(A.xh1);
#endif
#endif
    slice_to_acc(imgB + wave * (IMG / NW), raw, g);      // image B: first written behind barrier 1
#if BF_LATE_HPRE
    // the saved rows are requested only now: the single resident round starts in step, and every workgroup asking for its gradient rows AND
    // its saved rows at once made a 37 MB burst that the gradient rows -- the only thing P0 needs -- waited out (stamps: 17.6 k cycles)
    asm volatile("" ::: "memory");
#if BF_DMA_ROWS
    dma_rows(A.hpre);
#else
    dma_rows(A.xh1);
#endif
#endif
  }
  BF_STAMP(20);
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) *(u32x2*)(imgA + tt * 8192 + wro[rt]) = bf_pack4(drop4(g[rt][tt], A.seed_mlp ^ smix, off[tt] + 16 * rt));
  }
  BF_STAMP(21);
  BF_BURST(imgA, A.dy2, FS_C, 0);
  if (tid < 3 * FS_C / 4) *(f32x4*)(lbias + 4 * tid) = bias_in;
  BF_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the pre-activation rows has landed in image C (the barrier publishes it)
  __syncthreads();      // 1
  BF_STAMP(3);

  // ================================ P1: dact = W2^T dy2 ; dhpre = dact . gelu'(hpre) -> image B and memory ===============================
  {
    f32x4 acc[RT][NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][tt] = zero4;
#if BF_DMA_ROWS
    fs_slice_gemm<0, 24, NTT, RT, false, PF>(wqt, wb, imgA, rdo, acc, BF_HOOK(imgA, A.dy2, FS_C, 0));
    BF_STAMP(4);
    u32x2 hp[RT][NTT];      // (read together: the compiler cannot move an LDS read over the image writes below)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) hp[rt][tt] = *(const u32x2*)(imgC + tt * 8192 + wro[rt]);
    // image C is free as soon as every wave holds its pre-activations: LayerNorm2's rows are requested NOW and land under this epilogue.
    // (Requested behind barrier 2 they sat in front of P2's weight fragments in the wave's in-order memory queue: stamps, P2's GEMM
    // 10.5 k cycles against 2.5 k for P1's.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();      // 1b
    dma_rows(A.xh2);
#else
    u32x2 hp[RT][NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) hp[rt][tt] = *(const u32x2*)(A.hpre + off[tt] + 16 * rt);      // in flight under the GEMM
    fs_slice_gemm<0, 24, NTT, RT, false, PF>(wqt, wb, imgA, rdo, acc, BF_HOOK(imgA, A.dy2, FS_C, 0));
    BF_STAMP(4);
#endif
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 h = bf_unpack4(hp[rt][tt]);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = acc[rt][tt][e] * bf_gelu_tanh_grad(h[e]);
        *(u32x2*)(imgB + tt * 8192 + wro[rt]) = bf_pack4(d);
      }
      rstd2[tt] = A.st2[2 * (long)(lv[tt] ? tokidx[tt] : 0) + 1];
    }
  }
  BF_BURST(imgB, A.dhpre, FS_C, 0);
  __syncthreads();      // 2
  BF_STAMP(5);

  // ================================ P2: dxh2 = W1'^T dhpre ; LayerNorm2 backward + skip -> dx1 (registers) ; dy1 -> image A and memory ====
  {
    f32x4 acc[RT][NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][tt] = zero4;
#if BF_DMA_ROWS
    fs_slice_gemm<1, 16, NTT, RT, false, PF>(wqt, wb, imgB, rdo, acc, BF_HOOK(imgB, A.dhpre, FS_C, 0));
    BF_STAMP(6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();      // 2b: LayerNorm2's rows are in image C
    u32x2 xh[RT][NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) xh[rt][tt] = *(const u32x2*)(imgC + tt * 8192 + wro[rt]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();      // 2c: every wave holds LayerNorm2's rows: LayerNorm1's are requested now and land under this epilogue
    dma_rows(A.xh1);
#else
    u32x2 xh[RT][NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) xh[rt][tt] = *(const u32x2*)(A.xh2 + off[tt] + 16 * rt);      // in flight under the GEMM
    fs_slice_gemm<1, 16, NTT, RT, false, PF>(wqt, wb, imgB, rdo, acc, BF_HOOK(imgB, A.dhpre, FS_C, 0));
    BF_STAMP(6);
#endif
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 xv = bf_unpack4(xh[rt][tt]);
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
    __syncthreads();      // 3 (image A's last readers, P1's GEMM, are behind barrier 2)
    // (the packed rows are unpacked AGAIN below: kept unpacked across the barrier, 24 more registers, the phase went to scratch)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) asm volatile("" : "+v"(xh[rt][tt]));
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const f32x4* sp = (const f32x4*)(stat + (16 * tt + l15) * NW * 8);
      const f32x4 a = sp[0], b = sp[1];
      const float m1 = ((a[0] + a[2]) + (b[0] + b[2])) * (1.0f / FS_C), m2 = ((a[1] + a[3]) + (b[1] + b[3])) * (1.0f / FS_C);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 xv = bf_unpack4(xh[rt][tt]);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = g[rt][tt][e] + rstd2[tt] * (acc[rt][tt][e] - m1 - xv[e] * m2);
        g[rt][tt] = d;      // dx1
        *(u32x2*)(imgA + tt * 8192 + wro[rt]) = bf_pack4(drop4(d, A.seed_out ^ smix, off[tt] + 16 * rt));
      }
      __builtin_amdgcn_sched_barrier(0);      // tile by tile: interleaved, the tiles' hash and LayerNorm temporaries push the phase into scratch
    }
    // dx1 waits in the OUTPUT buffer for the end of the launch (this wave's own rows, read back by itself: no other workgroup touches them).
    // Held in 48 registers through the recompute and the attention it pushed the launch into scratch: the compiler spilled half of it
    // anyway, and every scratch reload in between -- a vector-memory wait in an in-order queue -- drained the row stores and DMAs in flight.
#if BF_PARK_DX1
    slice_store(imgB + wave * (IMG / NW), A.dx, g, std::false_type{});      // image B: its readers (P2's GEMM) are behind barrier 3
#endif
  }
  // (Wo^T's first k-steps are requested HERE, not inside P2's GEMM: alive across LayerNorm2's backward, those 32 registers were what pushed
  // the phase into scratch -- and a scratch reload drains the wave's whole memory queue, row stores and DMAs included)
  fs_wring_prime<2, RT, PF>(wqt, wb);
  BF_BURST(imgA, A.dy1, FS_C, 0);
  __syncthreads();      // 4
  BF_STAMP(7);

  // ================================ P3: do = Wo^T dy1 -> operand fragments (registers) and image B ========================================
  {
    f32x4 acc[RT][NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][tt] = zero4;
    fs_slice_gemm<2, 24, NTT, RT, false, PF>(wqt, wb, imgA, rdo, acc, BF_HOOK(imgA, A.dy1, FS_C, 0));
    BF_STAMP(8);
    fs_wring_prime<0, RT, PF>(wqf, wb);      // the forward stream's q rows, under the epilogue and the barrier
    // image B (dhpre) died with P2's GEMM (barriers 3, 4): d_o in row-major form, for the plain and the transposing fragment reads
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) *(u32x2*)(imgB + tt * 8192 + wro[rt]) = bf_pack4(acc[rt][tt]);
  }
#if BF_DMA_ROWS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __syncthreads();      // 5: image A's readers (P3's GEMM) are done, LayerNorm1's rows are in image C
  BF_STAMP(9);

  // ================================ q | k | v recomputed from LayerNorm1's image with the forward's stream ===============================
  const float* const bias = lbias + 16 * RT * wave + 4 * kk;      // + 256 m + 16 rt
  u32x4 kf[HPW][NTT], vf[HPW][NTT];
  {
    f32x4 aq[RT][NTT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const f32x4 bq = *(const f32x4*)(bias + 16 * j);
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) aq[j][tt] = bq;
    }
    fs_slice_gemm<0, 24, NTT, RT, false, PF>(wqf, wb, imgC, rdo, aq);
    BF_STAMP(10);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int j = 0; j < RT; ++j) *(u32x2*)(imgA + tt * 8192 + wro[j]) = bf_pack4(aq[j][tt]);      // q' (log2(e) / sqrt(32) folded in), as the forward packed it
  }
  {
    f32x4 ak[RT][NTT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const f32x4 bk = *(const f32x4*)(bias + 256 + 16 * j);
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) ak[j][tt] = bk;
    }
    fs_slice_gemm<1, 24, NTT, RT, false, PF>(wqf, wb, imgC, rdo, ak);
    BF_STAMP(11);
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) kf[hh][tt] = pack8(ak[2 * hh][tt], ak[2 * hh + 1][tt]);
  }
  {
    f32x4 av[RT][NTT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const f32x4 bv = *(const f32x4*)(bias + 512 + 16 * j);
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) av[j][tt] = bv;
    }
    fs_slice_gemm<2, 24, NTT, RT, false, PF>(wqf, wb, imgC, rdo, av);
    BF_STAMP(12);
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) vf[hh][tt] = pack8(av[2 * hh][tt], av[2 * hh + 1][tt]);
  }
  __syncthreads();      // 6: image C's readers (the three GEMMs) are done
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {      // k in row-major form (this wave's columns)
      u32x2 u;
      u[0] = kf[rt >> 1][tt][2 * (rt & 1)];
      u[1] = kf[rt >> 1][tt][2 * (rt & 1) + 1];
      *(u32x2*)(imgC + tt * 8192 + wro[rt]) = u;
    }
  BF_STAMP(13);

  // ================================ attention backward: this wave's two heads, from its own columns of the images ========================
  {
    const float scale = 0.17677669529663687f, ln2 = 0.69314718055994531f;
    // TPS == 1: the block-diagonal (and causal) pattern inside a tile; bit r of allow1: key row 4 kk + r may be seen by query l15,
    // bit r of allow2: query row 4 kk + r sees key l15
    unsigned allow1 = 0xfu, allow2 = 0xfu;
    bool nomask = true;      // (TPS > 1: every key of the group's sequence is visible; constant-folded)
    const int si15 = (int)(((unsigned)l15 * A.magic) >> 16), pi15 = l15 - si15 * L;      // TPS == 1: sequence / position of slot l15 in its tile
    if constexpr (TPS == 1) {
      allow1 = allow2 = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * kk + r, sj = (int)(((unsigned)j * A.magic) >> 16), pj = j - sj * L;
        allow1 |= ((sj == si15 && (!A.causal || pj <= pi15)) ? 1u : 0u) << r;
        allow2 |= ((sj == si15 && (!A.causal || pi15 <= pj)) ? 1u : 0u) << r;
      }
      nomask = (L == 16) && !A.causal;
    }
    const unsigned long long sd_attn = A.seed_attn ^ smix;
    const unsigned aA = lds_addr(imgA), aB = lds_addr(imgB), aC = lds_addr(imgC);
    const int qq = l15 >> 2, pp = l15 & 3;
    // per-query statistics of pass 1 (max + log2 l, delta), read back in pass 2 as four consecutive queries per lane: the bias block of LDS
    // (dead since the v GEMM) holds [2][16 NTT] floats per wave
    float* const wst = lbias + wave * (2 * 16 * NTT);
    static_assert(2 * 16 * NTT * 4 * NW <= 3 * FS_C * 4, "row statistics fit the bias block");
    // LDS reads are issued in groups and waited for ONCE per group (every read of this phase used to be followed by its own
    // s_waitcnt lgkmcnt(0): ~70 exposed LDS round trips per head at L = 48, and ~100 ds_bpermute round trips for the statistics)
    auto rd64 = [&](unsigned addr) {
      u32x2 r;
      asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr) : "memory");
      return r;
    };
    auto wait0 = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto tie2 = [&](u32x2& v) { asm volatile("" : "+v"(v)); };      // keeps every use of v behind the wait above it (asm volatile statements keep their order)
    static_for<HPW>([&](auto hh_c) {
      constexpr int hh = decltype(hh_c)::value;
      const int head = HPW * wave + hh;
      // "row = token l15, k = dims" fragment of (image, tile): the packed-accumulator k order -- lane (l15, kk) holds dims 4 kk .. 4 kk + 3 of
      // the head's first 16 and of its second 16 -- so that it contracts against vf: two 8-byte reads
      const unsigned fo0 = (unsigned)(l15 * FS_ROW + (((2 * RT * wave + 4 * hh + (kk >> 1)) ^ l15) << 4) + (kk & 1) * 8);
      const unsigned fo1 = (unsigned)(l15 * FS_ROW + (((2 * RT * wave + 4 * hh + 2 + (kk >> 1)) ^ l15) << 4) + (kk & 1) * 8);
      // transposed fragment (rows = dims 16 dt .. + 15 of the head, k = tokens of tiles t0 | t1 in packed-pair order): lane 4 qq + pp of
      // 16-lane group kk supplies row 4 kk + qq, columns 4 pp .. 4 pp + 3 of the 16 x 16 block
      unsigned to[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int row = 4 * kk + qq, chunk = 2 * RT * wave + 4 * hh + 2 * dt + (pp >> 1);
        to[dt] = (unsigned)(row * FS_ROW + ((chunk ^ row) << 4) + (pp & 1) * 8);
      }
      // dropout mask rows: index = ((((seq0 + si) 8 + head) L + pi) L + key position  (block_sliced.hip); hb = the part that fits 32 bits
      const unsigned hb = (unsigned)(seq0 * 8 + head);
      u32x2 dqp[NTT][2];
      unsigned m4s[NTT][NK];      // pass 1's keep bits per (query tile, visible key tile): pass 2 fetches them across lanes instead of hashing again
      // ---- pass 1: queries in the columns ---------------------------------------------------------------------------------------------
      static_for<NTT>([&](auto qt_c) {
        constexpr int qt = decltype(qt_c)::value, k0 = (qt / TPS) * TPS, NP = (NK + 1) / 2;
        u32x2 q0 = rd64(aA + fo0 + qt * 8192u), q1 = rd64(aA + fo1 + qt * 8192u), g0 = rd64(aB + fo0 + qt * 8192u), g1 = rd64(aB + fo1 + qt * 8192u);
        u32x2 k0r[NK], k1r[NK];
#pragma unroll
        for (int j = 0; j < NK; ++j) {
          k0r[j] = rd64(aC + fo0 + (k0 + j) * 8192u);
          k1r[j] = rd64(aC + fo1 + (k0 + j) * 8192u);
        }
        wait0();
        tie2(q0); tie2(q1); tie2(g0); tie2(g1);
#pragma unroll
        for (int j = 0; j < NK; ++j) { tie2(k0r[j]); tie2(k1r[j]); }
        const u32x4 qfq = u32x4{q0[0], q0[1], q1[0], q1[1]}, gfq = u32x4{g0[0], g0[1], g1[0], g1[1]};
        f32x4 st[NK], dpt[NK];
#pragma unroll
        for (int j = 0; j < NK; ++j) {
          st[j] = mfma_bf16(u32x4{k0r[j][0], k0r[j][1], k1r[j][0], k1r[j][1]}, qfq, zero4);      // rows = keys 4 kk + r, column = query l15 (log2 domain)
          dpt[j] = mfma_bf16(vf[hh][k0 + j], gfq, zero4);
        }
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nomask || ((allow1 >> r) & 1u)) m = fmaxf(m, st[j][r]);
        m = rows_max(m);
        float sm = 0.0f;
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            st[j][r] = (nomask || ((allow1 >> r) & 1u)) ? __builtin_amdgcn_exp2f(st[j][r] - m) : 0.0f;
            sm += st[j][r];
          }
        sm = rows_sum(sm);
        const float il = sm > 0.0f ? __builtin_amdgcn_rcpf(sm) : 0.0f;
        // dropout on the probabilities: the forward's mask indices
        const int qs = 16 * qt + l15, si = (int)(((unsigned)qs * A.magic) >> 16), pi = qs - si * L;
        const unsigned long long mrow = ((unsigned long long)(hb + 8u * (unsigned)si) * (unsigned)L + (unsigned)pi) * (unsigned)L;
        float delta = 0.0f;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
          unsigned m4 = 0xfu;
          if constexpr (DROP) {
            const int jpos0 = 16 * (k0 + j) + 4 * kk - si * L;
            if ((L & 3) == 0) {
              m4 = dropout_keep4(sd_attn, mrow + (unsigned long long)(jpos0 < 0 ? 0 : jpos0), A.p);
            } else {
              m4 = 0u;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int jpos = jpos0 + r;
                m4 |= (dropout_keep(sd_attn, mrow + (unsigned long long)(jpos < 0 ? 0 : jpos), A.p) ? 1u : 0u) << r;
              }
            }
          }
          m4s[qt][j] = m4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = st[j][r] * il;
            float dp = dpt[j][r];
            if constexpr (DROP) dp = ((m4 >> r) & 1u) ? dp * ksc : 0.0f;      // d(dropped probability) -> d(probability)
            st[j][r] = pr;
            dpt[j][r] = dp;
            delta = fmaf(pr, dp, delta);
          }
        }
        delta = rows_sum(delta);
        if (kk == 0) {      // pass 2 rebuilds p = exp2(s - max) / l as exp2(s - c), c = max + log2(l)
          wst[16 * qt + l15] = sm > 0.0f ? m + __builtin_amdgcn_logf(sm) : INFINITY;
          wst[16 * NTT + 16 * qt + l15] = delta;
        }
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) dpt[j][r] = st[j][r] * (dpt[j][r] - delta) * scale;      // dS^T / sqrt(32)
        f32x4 dq[2] = {zero4, zero4};
        static_for<NP>([&](auto g_c) {      // K^T fragments pair by pair (every pair's fragments alive at once does not fit the register file)
          constexpr int g = decltype(g_c)::value, ja = 2 * g, jb = ja + 1;
          u32x2 ktl[2], kth[2];
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            ktl[dt] = bf_tr(aC + to[dt] + (unsigned)(k0 + ja) * 8192u);
            kth[dt] = bf_tr(aC + to[dt] + (unsigned)(k0 + (jb < NK ? jb : ja)) * 8192u);
          }
          const u32x4 pf = pack8(dpt[ja], jb < NK ? dpt[jb < NK ? jb : ja] : zero4);
          wait0();
          tie2(ktl[0]); tie2(ktl[1]); tie2(kth[0]); tie2(kth[1]);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) dq[dt] = mfma_bf16(u32x4{ktl[dt][0], ktl[dt][1], kth[dt][0], kth[dt][1]}, pf, dq[dt]);
        });
        dqp[qt][0] = bf_pack4(dq[0]);
        dqp[qt][1] = bf_pack4(dq[1]);
        __builtin_amdgcn_sched_barrier(0);      // tile after tile: interleaved (the mask hashes depend on indices only and float to the top), the
      });                                       // tiles' temporaries do not fit the register file
      // ---- pass 2: keys in the columns ------------------------------------------------------------------------------------------------
      u32x2 dkp[NTT][2], dvp[NTT][2];
      static_for<NTT>([&](auto jt_c) {
        constexpr int jt = decltype(jt_c)::value, k0 = (jt / TPS) * TPS, NP = (NK + 1) / 2;
        u32x2 kj0 = rd64(aC + fo0 + jt * 8192u), kj1 = rd64(aC + fo1 + jt * 8192u);
        u32x2 q0r[NK], q1r[NK], g0r[NK], g1r[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
          q0r[i] = rd64(aA + fo0 + (k0 + i) * 8192u); q1r[i] = rd64(aA + fo1 + (k0 + i) * 8192u);
          g0r[i] = rd64(aB + fo0 + (k0 + i) * 8192u); g1r[i] = rd64(aB + fo1 + (k0 + i) * 8192u);
        }
        wait0();
        tie2(kj0); tie2(kj1);
#pragma unroll
        for (int i = 0; i < NK; ++i) { tie2(q0r[i]); tie2(q1r[i]); tie2(g0r[i]); tie2(g1r[i]); }
        const u32x4 kfj = u32x4{kj0[0], kj0[1], kj1[0], kj1[1]};
        f32x4 sv[NK], dp[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
          sv[i] = mfma_bf16(u32x4{q0r[i][0], q0r[i][1], q1r[i][0], q1r[i][1]}, kfj, zero4);      // rows = queries 4 kk + r, column = key l15
          dp[i] = mfma_bf16(u32x4{g0r[i][0], g0r[i][1], g1r[i][0], g1r[i][1]}, vf[hh][jt], zero4);
        }
#pragma unroll
        for (int i = 0; i < NK; ++i) {
          // the statistics of queries 4 kk .. 4 kk + 3 of this query tile
          const f32x4 cqi = *(const f32x4*)(wst + 16 * (k0 + i) + 4 * kk), dqi = *(const f32x4*)(wst + 16 * NTT + 16 * (k0 + i) + 4 * kk);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool valid = nomask || ((allow2 >> r) & 1u);
            const float pr = valid ? __builtin_amdgcn_exp2f(sv[i][r] - cqi[r]) : 0.0f;
            float d = dp[i][r], pd = pr;
            if constexpr (DROP) {
              // pass 1 drew the keep bits of (query 4 kk + r of tile k0 + i, keys 4 (l15 >> 2) .. + 3 of this tile) in lane
              // (l15 = that query, kk = l15 >> 2 of THIS lane): one ds_bpermute instead of a 64-bit index and a hash per element
              // (544 quarter-rate integer multiplies per head at L = 48 before)
              const unsigned kb = (unsigned)__shfl((int)m4s[k0 + i][jt - k0], (4 * kk + r) | ((l15 >> 2) << 4));
              const bool keep = (kb >> (l15 & 3)) & 1u;
              pd = keep ? pr * ksc : 0.0f;
              d = keep ? d * ksc : 0.0f;
            }
            sv[i][r] = pd;                                  // dropped probabilities (dV)
            dp[i][r] = pr * (d - dqi[r]) * ln2;             // dS ln 2 (dK, against q')
          }
        }
        f32x4 dk[2] = {zero4, zero4}, dv[2] = {zero4, zero4};
        static_for<NP>([&](auto g_c) {      // Q^T and dO^T fragments for dK, dV, pair by pair
          constexpr int g = decltype(g_c)::value, ia = 2 * g, ib = ia + 1;
          u32x2 qtl[2], qth[2], gtl[2], gth[2];
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const unsigned ta = to[dt] + (unsigned)(k0 + ia) * 8192u, tb = to[dt] + (unsigned)(k0 + (ib < NK ? ib : ia)) * 8192u;
            qtl[dt] = bf_tr(aA + ta); qth[dt] = bf_tr(aA + tb);
            gtl[dt] = bf_tr(aB + ta); gth[dt] = bf_tr(aB + tb);
          }
          const u32x4 pfs = pack8(dp[ia], ib < NK ? dp[ib < NK ? ib : ia] : zero4);
          const u32x4 pfp = pack8(sv[ia], ib < NK ? sv[ib < NK ? ib : ia] : zero4);
          wait0();
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) { tie2(qtl[dt]); tie2(qth[dt]); tie2(gtl[dt]); tie2(gth[dt]); }
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            dk[dt] = mfma_bf16(u32x4{qtl[dt][0], qtl[dt][1], qth[dt][0], qth[dt][1]}, pfs, dk[dt]);
            dv[dt] = mfma_bf16(u32x4{gtl[dt][0], gtl[dt][1], gth[dt][0], gth[dt][1]}, pfp, dv[dt]);
          }
        });
        dkp[jt][0] = bf_pack4(dk[0]); dkp[jt][1] = bf_pack4(dk[1]);
        dvp[jt][0] = bf_pack4(dv[0]); dvp[jt][1] = bf_pack4(dv[1]);
        __builtin_amdgcn_sched_barrier(0);
      });
      // the head's results over its operands' columns (every read of them is behind us in this wave's in-order LDS queue)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          *(u32x2*)(imgC + tt * 8192 + wro[2 * hh + dt]) = dqp[tt][dt];
          *(u32x2*)(imgA + tt * 8192 + wro[2 * hh + dt]) = dkp[tt][dt];
          *(u32x2*)(imgB + tt * 8192 + wro[2 * hh + dt]) = dvp[tt][dt];
        }
      BF_STAMP(14 + hh);
      __builtin_amdgcn_sched_barrier(0);      // the two heads one after the other: interleaved, their fragment sets do not fit the register file
    });
  }
  fs_wring_prime<0, RT, PF>(wqh, wb);
  float rstd1[NTT];
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) rstd1[tt] = A.st1[2 * (long)(lv[tt] ? tokidx[tt] : 0) + 1];
#if BF_PARK_DX1
  f32x4 dx1raw[NTT][RT];      // dx1 back from the output buffer (row form), in flight under the closing GEMMs
  slice_load(A.dx, dx1raw);
#endif
#if !BF_HOOK_STORES
  // (issued per head, right behind each head's image writes, so that head 0's drain under head 1: measured, same box, no gain -- 8.94 ms
  // per train step either way; what the closing GEMMs wait for is not these stores' place in the queue)
  BF_BURST(imgC, A.dqkv, 3 * FS_C, 0);
  BF_BURST(imgA, A.dqkv, 3 * FS_C, FS_C);
  BF_BURST(imgB, A.dqkv, 3 * FS_C, 2 * FS_C);
#endif
#if !BF_DMA_ROWS
  u32x2 xh1r[RT][NTT];      // in flight under the closing GEMMs
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) xh1r[rt][tt] = *(const u32x2*)(A.xh1 + off[tt] + 16 * rt);
#endif
  __syncthreads();      // 7
  BF_STAMP(16);

  // ================================ dxh1 = [dq | dk | dv] W_in' ; LayerNorm1 backward + dx1 -> dx ==========================================
  {
    f32x4 acc[RT][NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][tt] = zero4;
    // dq | dk | dv rows -> memory (the in-projection's weight gradient reads them) under the three GEMMs that read the same images
    fs_slice_gemm<0, 24, NTT, RT, false, PF>(wqh, wb, imgC, rdo, acc, BF_HOOK(imgC, A.dqkv, 3 * FS_C, 0));
#if BF_DMA_ROWS
    __syncthreads();      // 7b: image C (dq) has no reader left: LayerNorm1's rows come back into it under the other two GEMMs (held in
    dma_rows(A.xh1);      // registers through the attention they pushed the launch into scratch)
#endif
    fs_slice_gemm<1, 24, NTT, RT, false, PF>(wqh, wb, imgA, rdo, acc, BF_HOOK(imgA, A.dqkv, 3 * FS_C, FS_C));
    fs_slice_gemm<2, 24, NTT, RT, false, PF>(wqh, wb, imgB, rdo, acc, BF_HOOK(imgB, A.dqkv, 3 * FS_C, 2 * FS_C));
#if BF_DMA_ROWS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();      // 8a
    u32x2 xh1r[RT][NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) xh1r[rt][tt] = *(const u32x2*)(imgC + tt * 8192 + wro[rt]);
#endif
    BF_STAMP(17);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 xv = bf_unpack4(xh1r[rt][tt]);
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
    __syncthreads();      // 8: the statistics; every image is dead behind it
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) asm volatile("" : "+v"(xh1r[rt][tt]));
    BF_STAMP(18);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      const f32x4* sp = (const f32x4*)(stat + (16 * tt + l15) * NW * 8);
      const f32x4 a = sp[0], b = sp[1];
      const float m1 = ((a[0] + a[2]) + (b[0] + b[2])) * (1.0f / FS_C), m2 = ((a[1] + a[3]) + (b[1] + b[3])) * (1.0f / FS_C);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 xv = bf_unpack4(xh1r[rt][tt]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[rt][tt][e] = rstd1[tt] * (acc[rt][tt][e] - m1 - xv[e] * m2);
      }
    }
    // tile by tile through a private staging piece (every image is dead behind barrier 8): dx1's rows -> accumulator layout, the sum back
    // to row form, out (write-through: the launch's last output, at its very end)
    char* const stg = imgA + wave * (IMG / NW);
    BF_RLANE;
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
#if BF_PARK_DX1
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int r = RPI * j + rrow;
        *(f32x4*)(stg + r * ROWB + ((rchunk ^ (r & (CPR - 1))) << 4)) = dx1raw[tt][j];
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int o = l15 * ROWB + (((4 * rt + kk) ^ (l15 & (CPR - 1))) << 4);
        const f32x4 d = *(const f32x4*)(stg + o) + acc[rt][tt];
        *(f32x4*)(stg + o) = d;      // (the same lane reads and rewrites its own 16 bytes)
      }
#else
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) *(f32x4*)(stg + l15 * ROWB + (((4 * rt + kk) ^ (l15 & (CPR - 1))) << 4)) = g[rt][tt] + acc[rt][tt];
#endif
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int r = RPI * j + rrow, t = tok_of(16 * tt + r);
        const f32x4 v = *(const f32x4*)(stg + r * ROWB + ((rchunk ^ (r & (CPR - 1))) << 4));
        if (t >= 0) st_wt16(A.dx + (long)t * FS_C + 16 * RT * wave + 4 * rchunk, v);
      }
    }
  }
  BF_STAMP(19);
#ifdef TANTE_ABLATE
  if (A.stamps && lane == 0) A.stamps[((long)blockIdx.x * 4 + wave) * 24 + 23] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int TPS, int NTT>
void bf_launch_d(const BfArgs& A, int nwg, hipStream_t s);
template <int TPS, int NTT, bool DROP>
void bf_launch(const BfArgs& A, int nwg, hipStream_t s) {
  constexpr int LDS = 3 * 16 * NTT * FS_ROW + 16 * NTT * 4 * 8 + 3 * FS_C * 4;
  static_assert(LDS <= 160 * 1024, "LDS per workgroup");
  static TantePerDevice attr;
  attr.once([&] { (void)hipFuncSetAttribute((const void*)block_bwd_fs_kernel<TPS, NTT, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
  hipLaunchKernelGGL((block_bwd_fs_kernel<TPS, NTT, DROP>), dim3((unsigned)nwg), dim3(256), LDS, s, A);
}
template <int TPS, int NTT>
void bf_launch_d(const BfArgs& A, int nwg, hipStream_t s) {
  if (A.p > 0.0f) bf_launch<TPS, NTT, true>(A, nwg, s);
  else bf_launch<TPS, NTT, false>(A, nwg, s);
}

// (TPS, NTT) for a sequence length, 0 = no instantiation (the three-launch path serves it)
bool bf_shape(int L, int causal, int& tps, int& ntt) {
  if (L >= 1 && 16 % L == 0) { tps = 1; ntt = 3; return true; }      // several sequences per tile (or one, L = 16); 48-token workgroups
  if (causal) return false;
  if (L == 32) { tps = 2; ntt = 2; return true; }
  if (L == 48) { tps = 3; ntt = 3; return true; }
  if (L == 64) { tps = 4; ntt = 4; return true; }
  return false;
}

}  // namespace

#ifdef TANTE_ABLATE
extern "C" void tante_bf_set_stamps(void* p) { g_bf_stamps = (unsigned long long*)p; }   // (workgroups * 4 waves * 24) u64, diagnostic builds only
#endif

extern "C" int tante_block_bwd_fused_supported(int C, int n_head, int hidden, int L, int causal) {
  int tps, ntt;
  return C == FS_C && n_head == 8 && hidden == FS_C && bf_shape(L, causal, tps, ntt);
}

extern "C" int tante_block_bwd_fused(const float* dout, const void* xh1, const float* st1, const void* hpre, const void* xh2, const float* st2,
                                     const void* tail_bwd_stream, const void* block_stream, const void* head_bwd_stream, int C, int n_head,
                                     int hidden, const TanteSeq* seq, int causal, float p_drop, uint64_t seed_attn, uint64_t seed_out,
                                     uint64_t seed_mlp, float* dx, void* dy2, void* dhpre, void* dy1, void* dqkv, void* stream) {
  if (!dout || !xh1 || !st1 || !hpre || !xh2 || !st2 || !tail_bwd_stream || !block_stream || !head_bwd_stream || !seq || !dx || !dy2 || !dhpre ||
      !dy1 || !dqkv)
    TANTE_FAIL(-1, "tante_block_bwd_fused: null pointer");
  int tps = 0, ntt = 0;
  if (C != FS_C || n_head != 8 || hidden != FS_C || !bf_shape(seq->L, causal, tps, ntt))
    TANTE_FAIL(-2, "tante_block_bwd_fused: unsupported shape C=%d heads=%d hidden=%d L=%d causal=%d", C, n_head, hidden, seq->L, causal);
  if (p_drop < 0.0f || p_drop >= 1.0f) TANTE_FAIL(-1, "tante_block_bwd_fused: dropout probability %f", (double)p_drop);
  if (seq->nseq <= 0 || seq->nseq >= (1 << 23)) TANTE_FAIL(-2, "tante_block_bwd_fused: %d sequences", seq->nseq);
  if ((((uintptr_t)dout | (uintptr_t)xh1 | (uintptr_t)hpre | (uintptr_t)xh2 | (uintptr_t)dx | (uintptr_t)dy2 | (uintptr_t)dhpre | (uintptr_t)dy1 |
        (uintptr_t)dqkv | (uintptr_t)tail_bwd_stream | (uintptr_t)block_stream | (uintptr_t)head_bwd_stream) & 15) ||
      (((uintptr_t)st1 | (uintptr_t)st2) & 7))
    TANTE_FAIL(-1, "tante_block_bwd_fused: 16-byte alignment");
  BfArgs A;
  A.dout = dout; A.xh1 = (const unsigned short*)xh1; A.hpre = (const unsigned short*)hpre; A.xh2 = (const unsigned short*)xh2;
  A.st1 = st1; A.st2 = st2;
  A.wt = (const char*)tail_bwd_stream;
  A.wf = (const char*)block_stream + tante_block_stream_bytes(C, hidden) - tante_fs_stream_bytes(C, hidden);      // the feature-sliced part of the block stream
  A.wh = (const char*)head_bwd_stream;
  A.dx = dx; A.dy2 = (unsigned short*)dy2; A.dhpre = (unsigned short*)dhpre; A.dy1 = (unsigned short*)dy1; A.dqkv = (unsigned short*)dqkv;
  A.sq = *seq; A.causal = causal;
  A.spw = 16 * ntt / seq->L;
  A.magic = (65536u + (unsigned)seq->L - 1u) / (unsigned)seq->L;
  A.stamps = nullptr;
#ifdef TANTE_ABLATE
  A.stamps = g_bf_stamps;
#endif
  A.p = p_drop; A.seed_attn = seed_attn; A.seed_out = seed_out; A.seed_mlp = seed_mlp; A.seed_mix = tante_seed_mix_ptr();
  const int nwg = (seq->nseq + A.spw - 1) / A.spw;
  hipStream_t s = (hipStream_t)stream;
  const int key = tps * 10 + ntt;
  switch (key) {
    case 13: bf_launch_d<1, 3>(A, nwg, s); break;
    case 22: bf_launch_d<2, 2>(A, nwg, s); break;
    case 33: bf_launch_d<3, 3>(A, nwg, s); break;
    case 44: bf_launch_d<4, 4>(A, nwg, s); break;
    default: TANTE_FAIL(-2, "tante_block_bwd_fused: no instantiation for L=%d", seq->L);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}
