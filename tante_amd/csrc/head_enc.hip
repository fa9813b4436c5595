// Token-local tail of a rollout call in ONE launch (bf16, C = 256):
//   derivative heads of every Taylor order -> Taylor sum -> prediction frame -> RE-ENCODING of that frame for the next call
//   (enc_dec_cnn.py:263-277 + tante.py:165-171 + enc_dec_cnn.py:217-229).
//
// With patch_scale 8 (three 2 x 2 stages, no overlap) the 8 x 8 x D pixel block a token's heads write is exactly the block the three
// encoder stages reduce back to that token, so the predicted frame never has to be read back: the frame is stored (it is the output) and
// re-encoded from the same registers.  The next call's stage-1 GEMM + enc23_kernel launches (and their 25 MB) disappear.
//
// Work split as in head_fused.hip: a workgroup owns 16 NWV tokens and ONE stage-1 pixel p (a 4 x 4 pixel quadrant of every token's
// block).  Head stages chain through MFMA accumulators; the encoder runs the other way through the same quadrant: stage 1 on the four
// 2 x 2 sub-blocks q, stage 2 on the quadrant (tap q), and stage 3 needs all four quadrants -- each workgroup writes its stage-2 output
// (bf16 operand fragments, 32 KiB) through to memory, and the LAST of a group's four workgroups to arrive (one agent-scope counter per
// group, no spinning) reads the four back and runs stage 3 over the whole K = 512 as ONE accumulation chain in the fixed tap order
// 0..3: deterministic whatever the arrival order.  (The first form exchanged fp32 stage-3 partials, 128 KiB per workgroup: 67 MB written
// and read back by the whole chip at the same moment, 13 us of the launch; the operand fragments are a quarter of that and three of
// four workgroups skip stage 3.)
//
// Pipeline (the round-3 kernel waited three times for 142 KiB of weights and for its token rows, and its compiler-scheduled LDS reads
// waited lgkmcnt(0) -- a full LDS round trip per MFMA -- whenever an LDS-DMA was in flight): every GEMM stage reads its weight
// fragments through the explicit read ring of fused_common.hip.h (counted lgkmcnt), and everything that comes from memory for order
// k + 1 is requested while order k computes:  W1[k+1] and the token rows right after order k's stage 1 (the W1 tile is free then; the
// rows wait in registers), W3[k+1] after its stage 3; only W2[k+1] (it shares its LDS region with the row staging) arrives under
// stage 1 of its own order.  The encoder's weights ride the same slots: W1e is resident, W2e lands on W1 during the last order's
// stages 2 + 3, tap 0 of W3e on W2 during encoder stages 1 + 2; the reducing workgroup streams taps 1..3 through the two regions in turn.
#include "fused_common.hip.h"
#include <stdlib.h>

#if defined(HE_EXP_GELU_ID) && !defined(TANTE_ABLATE)
#error "HE_EXP_GELU_ID is a timing experiment (wrong results on purpose): build it with -DTANTE_ABLATE, never into the product library"
#endif

namespace {

constexpr int HE_HB = 1024;                          // bias block of a tile (one LDS-DMA pass)
constexpr int HE_T3 = 64 * 8 * 16 + HE_HB;           //  9 216  head W3 (64 rows x K 64)        | bias3
constexpr int HE_T1 = 128 * 32 * 16 + HE_HB;         // 66 560  head W1 pixel tile (128 x K 256) | bias1
constexpr int HE_T2 = 64 * 16 * 16 + HE_HB;          // 17 408  head W2 sub-pixel tile (64 x K 128) | bias2
constexpr int HE_E1 = 64 * 8 * 16 + HE_HB;           //  9 216  encoder W1e (64 x K 64: 4 taps x D <= 16 channels) | bias1e
constexpr int HE_E2 = 128 * 32 * 16 + HE_HB;         // 66 560  encoder W2e (128 x K 256)        | bias2e
constexpr int HE_E3 = 256 * 16 * 16;                 // 65 536  encoder W3e tap slice (256 x K 128)
constexpr int HE_LDS = HE_T3 + HE_T1 + 4 * HE_T2 + HE_E1;      // 154 624
constexpr long HE_ENC_BYTES = (long)HE_E1 + HE_E2 + 4L * HE_E3 + HE_HB;
static_assert(HE_E2 <= HE_T1 && HE_E3 <= 4 * HE_T2, "the encoder's tiles ride the head's LDS slots");

struct HeArgs {
  const float *xk0, *xk1, *xk2, *xk3;      // order k's residual stream (rows addressed by a_*)
  const char *wk0, *wk1, *wk2, *wk3;       // order k's head stream (tante_pack_head)
  float ck0, ck1, ck2, ck3;                // Taylor coefficients
  int n_ord;
  long a_s1, a_s0, a_off; int a_n0;        // row r = (img, hp, wp) -> (r / a_n0) * a_s1 + (r % a_n0) * a_s0 + a_off
  int n_img, Hp, Wp, D;
  float* out; long out_bstride;
  const float* last; long last_bstride;
  const char* we;                          // encoder stream (tante_pack_head_enc)
  float* z;                                // (rows, 256) fp32: the new frame's encoding before FiLM
  void* part;                              // (groups, 4 pixels, 4 TT fragments, threads) 16-byte words: stage 2's output as stage 3 reads it
  int* cnt;                                // (groups, 4) {arrivals, -, -, -}: zero between launches
  int groups;
  unsigned long long* stamps;              // -DTANTE_ABLATE builds only (tools/head_enc_stamps.py), else null
};

#ifdef TANTE_ABLATE
unsigned long long* g_he_stamps = nullptr;
#define HE_STAMP(k)                                                                          \
  do {                                                                                       \
    if (A.stamps) {                                                                          \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                            \
      if (lane == 0) A.stamps[((long)blockIdx.x * 8 + wave) * 40 + (k)] = t_;                \
    }                                                                                        \
  } while (0)
#else
#define HE_STAMP(k)
#endif

template <int NWV>
__device__ __forceinline__ void he_glds(const char* __restrict__ g, char* l, int bytes, int tid) {   // 1 KiB per wave pass
  const int wave = tid >> 6, lane = tid & 63;
  for (int off = wave * 1024; off < bytes; off += NWV * 1024)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off + lane * 16),
                                     (__attribute__((address_space(3))) void*)(l + off), 16, 0, 0);
}
#ifdef HE_EXP_GELU_ID      // timing experiment only (wrong results): the whole price of the GELU arithmetic (round-5 verdict item 4)
__device__ __forceinline__ f32x4 he_gelu(const f32x4& v) { return v; }
#else
__device__ __forceinline__ f32x4 he_gelu(const f32x4& v) { return gelu_poly4<false>(v); }
#endif
// agent-scope (sc1) 16-byte accesses to the hand-off buffer: the load misses this CU's L1, the store is written through (and waits the
// two states a VALU write to the data registers of a >64-bit store needs behind it: common.hip.h, st_wt16)
__device__ __forceinline__ u32x4 he_ld_agent(const u32x4* p) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ void he_st_agent(u32x4* p, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// Pin a packed fragment / an accumulator where the source computes it.  MFMAs and GELUs are pure values to the optimiser: without a use
// in front of the next barrier it sinks them below it (down to the block of their first real use) while the asm-volatile fragment reads
// stay put -- in the last order's copy of the body all 64 stage-1 fragments were live at once, ~600 bytes of scratch per lane.
__device__ __forceinline__ void he_pin(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void he_pin(f32x4& v) { asm volatile("" : "+v"(v)); }

// NDT = 16-row tiles of stage 3 that hold real channels (D <= 4 NDT): the Taylor accumulators, the frame operands and the stage-3 MFMAs of the
// dead tiles do not exist.  TT = 16-token tiles per wave: every weight fragment read from LDS feeds TT MFMAs.  In-kernel stamps of the
// TT = 1 form (8 waves x 16 tokens) showed stage 1 and stages 2 + 3 running at 60 - 80 % of the CU's LDS read rate (a ds_read_b128 is
// 8 cycles of the LDS pipe, the MFMA it feeds 4 cycles of the CU's four matrix pipes): 4 waves x 32 tokens halve the LDS traffic per token.
template <int NWV, int TT, bool ENC, int NDT>
__global__ __launch_bounds__(NWV * 64, 1) void head_enc_kernel(const HeArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifndef HE_RING
#define HE_RING 4
#endif
  constexpr int RD = TT == 2 ? HE_RING : 4;      // fragments in flight per wave in the LDS read ring
  char* w3s = smem;
  char* w1s = smem + HE_T3;
  char* w2s = w1s + HE_T1;
  char* w1es = w2s + 4 * HE_T2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  const int p = (blockIdx.x & 31) >> 3;                          // stage-1 pixel (kh, kw) = (p >> 1, p & 1)
  const int grp = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7);      // the four pixels of a group share blockIdx & 7 (one XCD)
  if (grp >= A.groups) return;
  HE_STAMP(0);

  // ---- rows: the wave's TT tiles of 16 consecutive tokens.  Hp Wp and a_n0 are multiples of 16 (checked by the launcher), so a tile never
  // straddles an image or an addressing block: one base offset + j a_s0, and a tile is live or dead as a whole (wave-uniform) ---------
  const int HW = A.Hp * A.Wp;
  const unsigned n_rows = (unsigned)A.n_img * (unsigned)HW;
  const unsigned row0 = (unsigned)__builtin_amdgcn_readfirstlane((grp * NWV + wave) * (TT * 16));
  bool live[TT];
  long row_base[TT];
  int img[TT], hp[TT], wp[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    live[tt] = row0 + 16u * tt < n_rows;
    const unsigned rowl = live[tt] ? row0 + 16u * tt : 0u;                     // dead tiles read tile 0 (valid memory) and store nothing
    const unsigned aq0 = rowl / (unsigned)A.a_n0, ar0 = rowl - aq0 * (unsigned)A.a_n0;
    row_base[tt] = (long)aq0 * A.a_s1 + (long)ar0 * A.a_s0 + A.a_off;
    img[tt] = (int)(rowl / (unsigned)HW);
    const int hw = (int)(rowl - (unsigned)img[tt] * (unsigned)HW) + l15;
    hp[tt] = (int)(((float)hw + 0.5f) * __builtin_amdgcn_rcpf((float)A.Wp));   // exact: hw < 2^22
    wp[tt] = hw - hp[tt] * A.Wp;
  }

  auto xk_of = [&](int o) { return o == 0 ? A.xk0 : o == 1 ? A.xk1 : o == 2 ? A.xk2 : A.xk3; };
  auto wk_of = [&](int o) { return o == 0 ? A.wk0 : o == 1 ? A.wk1 : o == 2 ? A.wk2 : A.wk3; };
  auto ck_of = [&](int o) { return o == 0 ? A.ck0 : o == 1 ? A.ck1 : o == 2 ? A.ck2 : A.ck3; };

  // The wave's token rows in ROW form (one instruction = the 1 KiB of one token), requested one order ahead: order k + 1's rows are
  // asked for at barrier (A) of order k, arrive under its stage 1, and wait as bf16 pairs through its stages 2 + 3.
  f32x4 xraw[TT][16];
  u32x2 xb[TT][16];
  auto load_rows = [&](const float* Xp) {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const float* src = Xp + row_base[tt] + 4 * lane;
#pragma unroll
      for (int j = 0; j < 16; ++j) xraw[tt][j] = *(const f32x4*)(src + (long)j * A.a_s0);
    }
  };
  auto pack_rows = [&]() {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        xb[tt][j][0] = pack_bf16x2(xraw[tt][j][0], xraw[tt][j][1]);
        xb[tt][j][1] = pack_bf16x2(xraw[tt][j][2], xraw[tt][j][3]);
      }
  };

  // ---- prologue: order 0's W1[p], W3 and rows (and the resident encoder stage-1 tile) ---------------------------------------------
  {
    const char* W0 = wk_of(0);
    he_glds<NWV>(W0 + HE_T3 + (long)p * HE_T1, w1s, HE_T1, tid);
    he_glds<NWV>(W0, w3s, HE_T3, tid);
    if constexpr (ENC) he_glds<NWV>(A.we, w1es, HE_E1, tid);
    load_rows(xk_of(0));
    pack_rows();
  }
  HE_STAMP(1);

  f32x4 dsum[TT][4][NDT];            // sum over the orders of coefficient x derivative, per sub-pixel q and channel tile ns
  f32x4 pre[TT][2][NDT][2];          // the last input frame's values (fetched during the last order)
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ns = 0; ns < NDT; ++ns) dsum[tt][q][ns] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int Wout = A.Wp * 8;

  auto run_order = [&](auto is_last_c, const int ord) {
    constexpr bool IS_LAST = decltype(is_last_c)::value;
    const float cord = ck_of(ord);
    const char* Wp = wk_of(ord);
    // Every LDS address of the order's body is derived from a lane id the optimiser cannot see through: loop-invariant code motion would
    // otherwise hoist ~50 address registers (the staging writes and reads, the fragment bases) out of the order loop and keep them
    // live -- or spilled -- through all of it.  Recomputing them is two or three VALU instructions each.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int kq = ln >> 4, lr = ln & 15;
    // ---- bf16 rows -> a private 8 KiB piece of the W2 region per tile -> B-operand fragments (k-permuted = accumulator order) --------
    u32x4 xf[TT][8];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      char* xs = w2s + (wave * TT + tt) * 8192;          // 16 rows x 512 B; 8-byte chunk c of row j at chunk c ^ (2 j): conflict-free both ways
#pragma unroll
      for (int j = 0; j < 16; ++j) *(u32x2*)(xs + j * 512 + ((ln ^ ((2 * j) & 63)) << 3)) = xb[tt][j];
#pragma unroll
      for (int b = 0; b < 8; ++b) {          // k-block b: features 32 b + 4 kk .. + 3 and 32 b + 16 + 4 kk .. + 3
        const u32x2 lo = *(const u32x2*)(xs + lr * 512 + (((8 * b + kq) ^ ((2 * lr) & 63)) << 3));
        const u32x2 hi = *(const u32x2*)(xs + lr * 512 + (((8 * b + 4 + kq) ^ ((2 * lr) & 63)) << 3));
        xf[tt][b] = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
    }
    unsigned a1[8], a2[4], a3[2];
#pragma unroll
    for (int b = 0; b < 8; ++b) a1[b] = lds_addr(w1s + lr * 512 + (swz_chunk(lr, b * 4 + kq, 32) << 4));     // W1: 512 B rows
#pragma unroll
    for (int b = 0; b < 4; ++b) a2[b] = lds_addr(w2s + lr * 256 + (swz_chunk(lr, b * 4 + kq, 16) << 4));     // W2 tiles: 256 B rows
#pragma unroll
    for (int b = 0; b < 2; ++b) a3[b] = lds_addr(w3s + lr * 128 + (swz_chunk(lr, b * 4 + kq, 8) << 4));      // W3: 128 B rows
    // this wave's pieces of W1 / W3 have landed (order 0: requested before its rows; later orders: long ago)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();      // (A) every wave has its fragments: the W2 tiles may land on the staging pieces
    HE_STAMP(2 + 6 * ord);
    he_glds<NWV>(Wp + HE_T3 + 4L * HE_T1, w2s, 4 * HE_T2, tid);
    if constexpr (!IS_LAST) load_rows(xk_of(ord + 1));             // in flight during stage 1
    // ---- stage 1: pixel p, two halves of its 128 channels; h1[kb] = B-operand k-blocks for stage 2 ------------------------------------
    u32x4 h1[TT][4];
    {
      const float* bias1 = (const float*)(w1s + 128 * 32 * 16);
      static_for<2>([&](auto hc) {
        constexpr int hh = decltype(hc)::value;
        f32x4 acc[TT][4];
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
          const f32x4 b = *(const f32x4*)(bias1 + (4 * hh + ns) * 16 + kq * 4);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) acc[tt][ns] = b;
        }
        mfma_stream<32, RD>([&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<(4 * hh + i % 4) * 8192>{a1[i / 4]}; },
                           [&](auto ic, const u32x4& wf) {
                             constexpr int i = decltype(ic)::value;
#pragma unroll
                             for (int tt = 0; tt < TT; ++tt) acc[tt][i % 4] = mfma_bf16(wf, xf[tt][i / 4], acc[tt][i % 4]);
                           });
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          h1[tt][2 * hh] = pack8(he_gelu(acc[tt][0]), he_gelu(acc[tt][1]));
          h1[tt][2 * hh + 1] = pack8(he_gelu(acc[tt][2]), he_gelu(acc[tt][3]));
          he_pin(h1[tt][2 * hh]); he_pin(h1[tt][2 * hh + 1]);
        }
      });
    }
    HE_STAMP(3 + 6 * ord);
    if constexpr (!IS_LAST) pack_rows();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HE_STAMP(4 + 6 * ord);
    __syncthreads();      // (B) the W2 tiles are complete; every wave is done with W1
    HE_STAMP(5 + 6 * ord);
    if constexpr (!IS_LAST) he_glds<NWV>(wk_of(ord + 1) + HE_T3 + (long)p * HE_T1, w1s, HE_T1, tid);     // next order's W1: lands under stages 2 + 3
    else if constexpr (ENC) he_glds<NWV>(A.we + HE_E1, w1s, HE_E2, tid);
    if constexpr (IS_LAST) {
      // the last input frame's pixels of this quadrant: in flight during stages 2 + 3.  Channels past D read channel D - 1 and are zeroed.
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        const float* src = A.last + (long)img[tt] * A.last_bstride;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ns = 0; ns < NDT; ++ns) {
            const int ch = 4 * ns + kq;
            const int y0 = hp[tt] * 8 + (p >> 1) * 4 + j * 2, x0 = wp[tt] * 8 + (p & 1) * 4;
            const float* lp = src + ((long)(ch < A.D ? ch : A.D - 1) * (A.Hp * 8) + y0) * Wout + x0;
            const f32x4 r0 = *(const f32x4*)lp, r1 = *(const f32x4*)(lp + Wout);
            const bool ok = ch < A.D;
            pre[tt][j][ns][0] = ok ? r0 : f32x4{0.f, 0.f, 0.f, 0.f};
            pre[tt][j][ns][1] = ok ? r1 : f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
    }
    // ---- stages 2 + 3: sub-pixel tile q of W2, then W3 ---------------------------------------------------------------------------
    const float* bias3 = (const float*)(w3s + 64 * 8 * 16);
    static_for<4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;           // sub-pixel (kh2, kw2) = (q >> 1, q & 1)
      const float* bias2 = (const float*)(w2s + q * HE_T2 + 64 * 16 * 16);
      f32x4 acc2[TT][4];
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) {
        const f32x4 b = *(const f32x4*)(bias2 + ns * 16 + kq * 4);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc2[tt][ns] = b;
      }
      mfma_stream<16, RD>([&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<q * HE_T2 + (i % 4) * 4096>{a2[i / 4]}; },
                         [&](auto ic, const u32x4& wf) {
                           constexpr int i = decltype(ic)::value;
#pragma unroll
                           for (int tt = 0; tt < TT; ++tt) acc2[tt][i % 4] = mfma_bf16(wf, h1[tt][i / 4], acc2[tt][i % 4]);
                         });
      u32x4 h2[TT][2];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int b = 0; b < 2; ++b) { h2[tt][b] = pack8(he_gelu(acc2[tt][2 * b]), he_gelu(acc2[tt][2 * b + 1])); he_pin(h2[tt][b]); }
      // stage 3: rows n3 = (co, kh3, kw3) = 16 ns + 4 kk + r -> channel co = 4 ns + kk, r = (kh3, kw3)
      f32x4 d[TT][NDT];
#pragma unroll
      for (int ns = 0; ns < NDT; ++ns) {
        const f32x4 b = *(const f32x4*)(bias3 + ns * 16 + kq * 4);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) d[tt][ns] = b;
      }
      mfma_stream<2 * NDT, RD>([&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<(i % NDT) * 2048>{a3[i / NDT]}; },
                              [&](auto ic, const u32x4& wf) {
                                constexpr int i = decltype(ic)::value;
#pragma unroll
                                for (int tt = 0; tt < TT; ++tt) d[tt][i % NDT] = mfma_bf16(wf, h2[tt][i / NDT], d[tt][i % NDT]);
                              });
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int ns = 0; ns < NDT; ++ns) { dsum[tt][q][ns] += d[tt][ns] * cord; he_pin(dsum[tt][q][ns]); }
    });
    HE_STAMP(6 + 6 * ord);
    if constexpr (!IS_LAST) {
      __syncthreads();    // (C) every wave is done with this order's W2 tiles and W3
      HE_STAMP(7 + 6 * ord);
      he_glds<NWV>(wk_of(ord + 1), w3s, HE_T3, tid);
    }
  };
  for (int ord = 0; ord + 1 < A.n_ord; ++ord) run_order(std::false_type{}, ord);
  run_order(std::true_type{}, A.n_ord - 1);

  if constexpr (ENC) {
    // W2e (requested after the last order's stage 1) is complete for this wave; after the barrier for all -- and every wave is done with W2 / W3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();      // (C')
    HE_STAMP(26);
    he_glds<NWV>(A.we + HE_E1 + HE_E2, w2s, HE_E3, tid);      // tap 0 of W3e for whichever workgroup reduces: lands under the frame stores and stages 1 + 2
  }
  // ---- the frame: out = last + sum_k c_k d_k, in the pair layout of the stores (rows y0, y0 + 1 x 4 pixels) --------------------------
  f32x4 fv[TT][4][NDT];              // [q][ns][r]: pixel r = (kh3, kw3) of sub-pixel q, channel 4 ns + kk
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ns = 0; ns < NDT; ++ns) {
        const f32x4 dl_ = dsum[tt][2 * j][ns], dr_ = dsum[tt][2 * j + 1][ns];
        const f32x4 t0 = pre[tt][j][ns][0] + f32x4{dl_[0], dl_[1], dr_[0], dr_[1]};
        const f32x4 t1 = pre[tt][j][ns][1] + f32x4{dl_[2], dl_[3], dr_[2], dr_[3]};
        if (live[tt] && 4 * ns + kk < A.D) {
          const int y0 = hp[tt] * 8 + (p >> 1) * 4 + j * 2, x0 = wp[tt] * 8 + (p & 1) * 4;
          float* o0 = A.out + (long)img[tt] * A.out_bstride + ((long)(4 * ns + kk) * (A.Hp * 8) + y0) * Wout + x0;
          *(f32x4*)o0 = t0;
          *(f32x4*)(o0 + Wout) = t1;
        }
        fv[tt][2 * j][ns] = f32x4{t0[0], t0[1], t1[0], t1[1]};
        fv[tt][2 * j + 1][ns] = f32x4{t0[2], t0[3], t1[2], t1[3]};
      }
  HE_STAMP(27);
  if constexpr (!ENC) return;
  if constexpr (ENC) {
    unsigned a1[8], a2[4], ae1[2];       // W2e sits in the W1 tile (512 B rows), the W3e slice on the W2 tiles (256 B rows)
#pragma unroll
    for (int b = 0; b < 8; ++b) a1[b] = lds_addr(w1s + l15 * 512 + (swz_chunk(l15, b * 4 + kk, 32) << 4));
#pragma unroll
    for (int b = 0; b < 4; ++b) a2[b] = lds_addr(w2s + l15 * 256 + (swz_chunk(l15, b * 4 + kk, 16) << 4));
#pragma unroll
    for (int b = 0; b < 2; ++b) ae1[b] = lds_addr(w1es + l15 * 128 + (swz_chunk(l15, b * 4 + kk, 8) << 4));
    // ---- encoder stage 1: the 2 x 2 x D pixels of sub-block q -> 64 channels (k = (ci, kh, kw): lane kk holds ci = kk + 4 ns) ---------
    u32x4 h1e[TT][4][2];
    {
      const float* bias1e = (const float*)(w1es + 64 * 8 * 16);
      constexpr int KB1 = NDT > 2 ? 2 : 1;          // k-blocks that hold real channels (ci < 8, ci < 16)
      static_for<4>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 xin[TT][2];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          xin[tt][0] = pack8(fv[tt][q][0], NDT > 1 ? fv[tt][q][NDT > 1 ? 1 : 0] : zero);
          xin[tt][1] = NDT > 2 ? pack8(fv[tt][q][NDT > 2 ? 2 : 0], NDT > 3 ? fv[tt][q][NDT > 3 ? 3 : 0] : zero) : u32x4{0u, 0u, 0u, 0u};
        }
        f32x4 acc[TT][4];
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
          const f32x4 b = *(const f32x4*)(bias1e + ns * 16 + kk * 4);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) acc[tt][ns] = b;
        }
        mfma_stream<4 * KB1, RD>([&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<(i % 4) * 2048>{ae1[i / 4]}; },
                                [&](auto ic, const u32x4& wf) {
                                  constexpr int i = decltype(ic)::value;
#pragma unroll
                                  for (int tt = 0; tt < TT; ++tt) acc[tt][i % 4] = mfma_bf16(wf, xin[tt][i / 4], acc[tt][i % 4]);
                                });
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          h1e[tt][q][0] = pack8(he_gelu(acc[tt][0]), he_gelu(acc[tt][1]));
          h1e[tt][q][1] = pack8(he_gelu(acc[tt][2]), he_gelu(acc[tt][3]));
          he_pin(h1e[tt][q][0]); he_pin(h1e[tt][q][1]);
        }
      });
    }
    HE_STAMP(28);
    // ---- encoder stage 2: position p, taps q -> 128 channels, two halves -----------------------------------------------------------
    u32x4 h2e[TT][4];
    {
      const float* bias2e = (const float*)(w1s + 128 * 32 * 16);
      static_for<2>([&](auto hc) {
        constexpr int hh = decltype(hc)::value;
        f32x4 acc[TT][4];
#pragma unroll
        for (int ns = 0; ns < 4; ++ns) {
          const f32x4 b = *(const f32x4*)(bias2e + (4 * hh + ns) * 16 + kk * 4);
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) acc[tt][ns] = b;
        }
        mfma_stream<32, RD>([&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<(4 * hh + i % 4) * 8192>{a1[i / 4]}; },
                           [&](auto ic, const u32x4& wf) {
                             constexpr int i = decltype(ic)::value, kb = i / 4;
#pragma unroll
                             for (int tt = 0; tt < TT; ++tt) acc[tt][i % 4] = mfma_bf16(wf, h1e[tt][kb >> 1][kb & 1], acc[tt][i % 4]);
                           });
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          h2e[tt][2 * hh] = pack8(he_gelu(acc[tt][0]), he_gelu(acc[tt][1]));
          h2e[tt][2 * hh + 1] = pack8(he_gelu(acc[tt][2]), he_gelu(acc[tt][3]));
          he_pin(h2e[tt][2 * hh]); he_pin(h2e[tt][2 * hh + 1]);
        }
      });
    }
    HE_STAMP(29);
    // ---- hand-off: this pixel's stage-2 output, fragment-wise (every store and every load is 1 KiB contiguous per wave), written
    // through to memory with agent scope (sc1) as MI355X_MICROARCH.md's hand-off table asks of both sides -----------------------------
    u32x4* hslot = (u32x4*)A.part + (long)grp * 4 * (4 * TT * NWV * 64) + tid;
    constexpr long HSTRIDE = 4L * TT * NWV * 64;           // 16-byte words between the slots of consecutive pixels
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int b = 0; b < 4; ++b) he_st_agent(hslot + p * HSTRIDE + (tt * 4 + b) * (NWV * 64), h2e[tt][b]);
    // ---- arrival: the fragments are in memory (write-through stores, acknowledged) before the counter moves; the barrier also says every
    // wave is done with W2e on the W1 tile --------------------------------------------------------------------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HE_STAMP(30);
    __syncthreads();
    int* flag = (int*)w3s;               // W3 is dead: the arrival ticket
    if (tid == 0) *(volatile int*)flag = __hip_atomic_fetch_add(A.cnt + 4 * grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    HE_STAMP(31);
    if (*(volatile int*)flag != 3) return;
    // ---- the last of the group's four workgroups: encoder stage 3 over K = 512, taps in the order 0..3 whatever the arrival order was.
    // Tap 0 has been on the W2 region since (C'); the taps alternate between the two big regions, each requested as soon as every wave
    // is done with the slice before it.
    if (tid == 0) A.cnt[4 * grp] = 0;    // ready for the next launch
    constexpr int TAPI = HE_E3 / (NWV * 1024);       // LDS-DMA instructions per wave and tap slice
    u32x4 hs[4][TT][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int b = 0; b < 4; ++b) hs[t][tt][b] = he_ld_agent(hslot + t * HSTRIDE + (tt * 4 + b) * (NWV * 64));
    he_glds<NWV>(A.we + HE_E1 + HE_E2 + 1L * HE_E3, w1s, HE_E3, tid);
    unsigned a3e[2][4];                  // tap slices: 256 B rows on the W2 region (even taps) / the W1 tile (odd taps)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      a3e[0][b] = a2[b];
      a3e[1][b] = lds_addr(w1s + l15 * 256 + (swz_chunk(l15, b * 4 + kk, 16) << 4));
    }
    f32x4 acc3[TT][16];
    {
      const float* bias3e = (const float*)(A.we + HE_E1 + HE_E2 + 4L * HE_E3);
#pragma unroll
      for (int ns = 0; ns < 16; ++ns) {
        const f32x4 b = *(const f32x4*)(bias3e + ns * 16 + kk * 4);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc3[tt][ns] = b;
      }
    }
    static_for<4>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      // tap t's pieces of this wave (and the fragments) have landed -- tap t + 1 may still be in flight behind them
      if constexpr (t == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TAPI) : "memory");
      else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                 // tap t is complete, and every wave is done with tap t - 1's region
        if constexpr (t + 1 < 4) he_glds<NWV>(A.we + HE_E1 + HE_E2 + (long)(t + 1) * HE_E3, (t + 1) % 2 ? w1s : w2s, HE_E3, tid);
      }
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int b = 0; b < 4; ++b) he_pin(hs[t][tt][b]);          // (asm-volatile loads: the optimiser must see them defined here)
      static_for<2>([&](auto hc) {
        constexpr int hh = decltype(hc)::value;
        mfma_stream<32, RD>([&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<(8 * hh + i % 8) * 4096>{a3e[t % 2][i / 8]}; },
                           [&](auto ic, const u32x4& wf) {
                             constexpr int i = decltype(ic)::value;
#pragma unroll
                             for (int tt = 0; tt < TT; ++tt) acc3[tt][8 * hh + i % 8] = mfma_bf16(wf, hs[t][tt][i / 8], acc3[tt][8 * hh + i % 8]);
                           });
      });
    });
    HE_STAMP(33);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
      if (live[tt]) {
        float* zrow = A.z + (long)(row0 + 16u * tt + (unsigned)l15) * 256 + kk * 4;
#pragma unroll
        for (int ns = 0; ns < 16; ++ns) *(f32x4*)(zrow + ns * 16) = acc3[tt][ns];
      }
    HE_STAMP(34);
  }
}

// ---- encoder stream packing: [W1e | b1e][W2e | b2e][W3e tap 0..3][b3e] ------------------------------------------------------------------
// k orders follow the registers of the kernel: stage 1 reads the frame block as (ci = kk + 4 ns, r = (kh, kw)); stages 2 and 3 read GELU'd
// accumulator tiles (position 8 kk + e of a 32-block holds feature 16 (e >> 2) + 4 kk + (e & 3)).
__device__ __forceinline__ int he_kperm(int pos) {
  const int blk = pos >> 5, qq = pos & 31, kq = qq >> 3, dt = (qq >> 2) & 1, r = qq & 3;
  return blk * 32 + dt * 16 + kq * 4 + r;
}
__global__ void pack_head_enc_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                     const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3, int D,
                                     char* __restrict__ dst) {
  const int t = blockIdx.x;     // 0: W1e, 1: W2e, 2..5: W3e taps, 6: b3e
  if (t == 0) {                 // conv1 (64, D, 2, 2): row n, position 32 kb + 8 kk + e -> ci = 8 kb + kk + 4 (e >> 2), (kh, kw) = e & 3
    for (int idx = threadIdx.x; idx < 64 * 8; idx += blockDim.x) {
      const int n = idx / 8, c = idx % 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int pos = c * 8 + e, kb = pos >> 5, kq = (pos & 31) >> 3, ci = 8 * kb + kq + 4 * (e >> 2), r = e & 3;
        v[e] = ci < D ? w1[((long)n * D + ci) * 4 + r] : 0.f;
      }
      u32x4 o;
      o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
      *((u32x4*)dst + (long)n * 8 + swz_chunk(n, c, 8)) = o;
    }
    float* bias = (float*)(dst + 64 * 8 * 16);
    for (int r = threadIdx.x; r < 256; r += blockDim.x) bias[r] = r < 64 ? b1[r] : 0.f;
  } else if (t == 1) {          // conv2 (128, 64, 2, 2): k-block kb = 2 tap + half, channels k-permuted inside the tap
    char* base = dst + HE_E1;
    for (int idx = threadIdx.x; idx < 128 * 32; idx += blockDim.x) {
      const int n = idx / 32, c = idx % 32;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int pos = c * 8 + e, tap = pos >> 6, ci = he_kperm(pos & 63);
        v[e] = w2[(((long)n * 64 + ci) * 2 + (tap >> 1)) * 2 + (tap & 1)];
      }
      u32x4 o;
      o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
      *((u32x4*)base + (long)n * 32 + swz_chunk(n, c, 32)) = o;
    }
    float* bias = (float*)(base + 128 * 32 * 16);
    for (int r = threadIdx.x; r < 256; r += blockDim.x) bias[r] = r < 128 ? b2[r] : 0.f;
  } else if (t <= 5) {          // conv3 (256, 128, 2, 2), tap = t - 2: rows n, K = 128 channels k-permuted
    const int tap = t - 2;
    char* base = dst + HE_E1 + HE_E2 + (long)tap * HE_E3;
    for (int idx = threadIdx.x; idx < 256 * 16; idx += blockDim.x) {
      const int n = idx / 16, c = idx % 16;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c2 = he_kperm(c * 8 + e);
        v[e] = w3[(((long)n * 128 + c2) * 2 + (tap >> 1)) * 2 + (tap & 1)];
      }
      u32x4 o;
      o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
      *((u32x4*)base + (long)n * 16 + swz_chunk(n, c, 16)) = o;
    }
  } else {
    float* bias = (float*)(dst + HE_E1 + HE_E2 + 4L * HE_E3);
    for (int r = threadIdx.x; r < 256; r += blockDim.x) bias[r] = b3[r];
  }
}

template <int NWV, int TT, bool ENC, int NDT>
void launch_head_enc_k(const HeArgs& A, hipStream_t s) {
  static TantePerDevice attr;
  attr.once([&] {
    (void)hipFuncSetAttribute((const void*)head_enc_kernel<NWV, TT, ENC, NDT>, hipFuncAttributeMaxDynamicSharedMemorySize, HE_LDS);
  });
  const unsigned grid = (unsigned)((A.groups + 7) / 8) * 32;   // 8 token groups x 4 pixels per 32 consecutive workgroups
  hipLaunchKernelGGL((head_enc_kernel<NWV, TT, ENC, NDT>), dim3(grid), dim3(NWV * 64), HE_LDS, s, A);
}
template <int NWV, int TT, bool ENC>
void launch_head_enc_nw(const HeArgs& A, hipStream_t s) {
  switch ((A.D + 3) / 4) {
    case 1: launch_head_enc_k<NWV, TT, ENC, 1>(A, s); break;
    case 2: launch_head_enc_k<NWV, TT, ENC, 2>(A, s); break;
    case 3: launch_head_enc_k<NWV, TT, ENC, 3>(A, s); break;
    default: launch_head_enc_k<NWV, TT, ENC, 4>(A, s); break;
  }
}

int he_group_tokens(long rows) {      // 128-token groups (8 waves) once they still give every CU a workgroup; 64-token groups for small batches
  const int force = tante_opt("TANTE_HEAD_WAVES", 0);
  return (force ? force == 8 : rows >= 128 * 56) ? 128 : 64;
}

}  // namespace

#ifdef TANTE_ABLATE
extern "C" void tante_head_enc_set_stamps(unsigned long long* p) { g_he_stamps = p; }
#endif

extern "C" int tante_head_enc_supported(int C, int D) { return C == 256 && D >= 1 && D <= 16; }

extern "C" int64_t tante_head_enc_stream_bytes(int C) { return C == 256 ? HE_ENC_BYTES : 0; }

/* workspace of tante_head_enc_fused for `rows` tokens: the four pixels' stage-2 fragments of every token group (128 bf16 per token and
 * pixel), then one arrival counter per token group */
static long he_frag_bytes(long groups, long gt) { return groups * 4 * gt * 128 * 2; }
extern "C" int64_t tante_head_enc_ws_bytes(int64_t rows) {
  if (rows <= 0) return 0;
  const long gt = he_group_tokens(rows), groups = (rows + gt - 1) / gt;
  return he_frag_bytes(groups, gt) + ((groups * 16 + 255) / 256) * 256;
}

extern "C" int tante_pack_head_enc(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int C, int D,
                                   void* enc_stream, void* stream) {
  if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !enc_stream) TANTE_FAIL(-1, "tante_pack_head_enc: null pointer");
  if (!tante_head_enc_supported(C, D)) TANTE_FAIL(-2, "tante_pack_head_enc: unsupported C=%d D=%d", C, D);
  hipLaunchKernelGGL(pack_head_enc_kernel, dim3(7), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, w3, b3, D, (char*)enc_stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_head_enc_fused(int n_ord, const float* const* rows, const void* const* head_streams, const float* coefs, int32_t a_n0,
                                    int64_t a_s1, int64_t a_s0, int64_t a_off, int n_img, int Hp, int Wp, int C, int D, float* out,
                                    int64_t out_bstride, const float* last, int64_t last_bstride, const void* enc_stream, float* z,
                                    void* ws, int64_t ws_bytes, void* stream) {
  if (!rows || !head_streams || !coefs || !out || !last) TANTE_FAIL(-1, "tante_head_enc_fused: null pointer");
  if (n_ord < 1 || n_ord > 4) TANTE_FAIL(-2, "tante_head_enc_fused: 1 .. 4 orders");
  if (!tante_head_enc_supported(C, D)) TANTE_FAIL(-2, "tante_head_enc_fused: unsupported C=%d D=%d", C, D);
  if (a_n0 <= 0 || n_img <= 0 || Hp <= 0 || Wp <= 0) TANTE_FAIL(-1, "tante_head_enc_fused: bad shape");
  if (a_s1 % 4 || a_s0 % 4 || a_off % 4 || out_bstride % 4 || last_bstride % 4 || ((uintptr_t)out % 16) || ((uintptr_t)last % 16))
    TANTE_FAIL(-1, "tante_head_enc_fused: alignment");
  for (int k = 0; k < n_ord; ++k)
    if (!rows[k] || !head_streams[k] || ((uintptr_t)rows[k] % 16) || ((uintptr_t)head_streams[k] % 16))
      TANTE_FAIL(-1, "tante_head_enc_fused: order %d: null or misaligned rows / stream", k);
  const bool enc = enc_stream != nullptr;
  const long n_rows = (long)n_img * Hp * Wp;
  if (enc) {
    if (!z || !ws) TANTE_FAIL(-1, "tante_head_enc_fused: the encoding needs z and a workspace");
    if (((uintptr_t)enc_stream % 16) || ((uintptr_t)z % 16) || ((uintptr_t)ws % 16)) TANTE_FAIL(-1, "tante_head_enc_fused: alignment (encoder)");
    if (ws_bytes < tante_head_enc_ws_bytes(n_rows)) TANTE_FAIL(-1, "tante_head_enc_fused: workspace of %lld bytes, %lld needed", (long long)ws_bytes,
                                                               (long long)tante_head_enc_ws_bytes(n_rows));
  }
  HeArgs A;
  const float* xs[4] = {nullptr, nullptr, nullptr, nullptr};
  const char* wsv[4] = {nullptr, nullptr, nullptr, nullptr};
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < n_ord; ++k) { xs[k] = rows[k]; wsv[k] = (const char*)head_streams[k]; cs[k] = coefs[k]; }
  A.xk0 = xs[0]; A.xk1 = xs[1]; A.xk2 = xs[2]; A.xk3 = xs[3];
  A.wk0 = wsv[0]; A.wk1 = wsv[1]; A.wk2 = wsv[2]; A.wk3 = wsv[3];
  A.ck0 = cs[0]; A.ck1 = cs[1]; A.ck2 = cs[2]; A.ck3 = cs[3];
  A.n_ord = n_ord;
  A.a_n0 = a_n0; A.a_s1 = a_s1; A.a_s0 = a_s0; A.a_off = a_off;
  A.n_img = n_img; A.Hp = Hp; A.Wp = Wp; A.D = D;
  A.out = out; A.out_bstride = out_bstride; A.last = last; A.last_bstride = last_bstride;
  const int gt = he_group_tokens(n_rows);
  A.groups = (int)((n_rows + gt - 1) / gt);
  A.we = (const char*)enc_stream; A.z = z;
  A.part = ws;
  A.cnt = enc ? (int*)((char*)ws + he_frag_bytes(A.groups, gt)) : nullptr;
#ifdef TANTE_ABLATE
  A.stamps = g_he_stamps;
#else
  A.stamps = nullptr;
#endif
  hipStream_t s = (hipStream_t)stream;
  // 128-token groups run as 8 waves x 1 tile.  The 4 waves x 2 tiles form (TT = 2: every LDS weight fragment feeds two MFMAs, half the LDS
  // traffic) was built and measured SLOWER, 78 against 67 us: the kernel is bound by instruction issue (per order and wave ~2 000
  // instructions for 152 MFMAs; GELU alone is 8 VALU instructions per element and packed fp32 math does not run in an MFMA's shadow),
  // not by LDS bandwidth or latency (ring depths 4 / 8 / 12: +-0), so one wave per SIMD doing twice the work gains nothing.  Kept as a
  // template parameter, not instantiated.
  if (gt == 128) { if (enc) launch_head_enc_nw<8, 1, true>(A, s); else launch_head_enc_nw<8, 1, false>(A, s); }
  else { if (enc) launch_head_enc_nw<4, 1, true>(A, s); else launch_head_enc_nw<4, 1, false>(A, s); }
  TANTE_CHECK_LAUNCH();
  return 0;
}
