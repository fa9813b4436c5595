// Fused derivative head (bf16):  tokens of the last time slot -> 3 x [ConvTranspose2d k = s = 2 (+GELU)] -> Taylor accumulation
// into the prediction frames, ONE launch per Taylor order  (enc_dec_cnn.py:263-277 + tante.py:165-171).
//
// A token's 8 x 8 x D output block depends on that token alone: stage 1 makes its 2 x 2 pixels (C/2 channels each), stage 2 the
// 2 x 2 sub-pixels of each (C/4 channels), stage 3 the 2 x 2 x D values of each.  A wave owns 16 tokens; every intermediate is
// an MFMA accumulator tile that is re-packed (bf16) into the next stage's B operand in registers (k order = accumulator order,
// weights packed to match -- the same chaining as the fused block kernel).  The weights stream through LDS once per workgroup:
// W3 (resident), W1 in 4 pixel tiles, W2 in 4 sub-pixel tiles.  The epilogue adds  coef_i * derivative  straight into the
// output frames (out_i = last + sum_k coef_ik d_k: the first order starts from the last input frame, later orders accumulate),
// so neither the three intermediate images nor the derivative fields nor a separate Taylor pass exist.
#include "common.hip.h"
#include <stdlib.h>
#include <utility>

namespace {

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ u32x4 hpack8(const f32x4& a, const f32x4& b) {
  u32x4 f;
  f[0] = pack_bf16x2(a[0], a[1]); f[1] = pack_bf16x2(a[2], a[3]); f[2] = pack_bf16x2(b[0], b[1]); f[3] = pack_bf16x2(b[2], b[3]);
  return f;
}
__device__ __forceinline__ f32x4 hmfma(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 gelu4(const f32x4& v) {
  return gelu_poly4<false>(v);
}
template <int NWV>
__device__ __forceinline__ void hglds(const char* __restrict__ g, char* l, int bytes, int tid) {   // 1 KiB per wave pass
  const int wave = tid >> 6, lane = tid & 63;
  for (int off = wave * 1024; off < bytes; off += NWV * 1024)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off + lane * 16),
                                     (__attribute__((address_space(3))) void*)(l + off), 16, 0, 0);
}

struct HeadArgs {
  const float* x;        // token stream
  long a_s1, a_s0, a_off; int a_n0;   // row r = (img, hp, wp) -> (r / a_n0) * a_s1 + (r % a_n0) * a_s0 + a_off
  int n_img, Hp, Wp, D;
  const char* w;         // packed stream: [W3 | bias3] [W1 tile p | bias1 p] x4 [W2 tile q | bias2 q] x4
  float* out; long out_bstride; int n_out;
  const float* last; long last_bstride;   // NULL: accumulate into out; else out_i = last + coef_i * d
  float coef[8];
  // MULTI form (fused_head_kernel<..., true>, one prediction frame): every Taylor order in ONE launch -- order k reads its last-slot token
  // rows from xk[k] (the rows as backbone k left them: dense (rows, C) copies for all but the last order) and its weight stream from wk[k];
  // the derivatives are summed in registers, coefficient ck[k] each, and the frame is read and written ONCE.
  int n_ord;
  const float *xk0, *xk1, *xk2, *xk3;
  const char *wk0, *wk1, *wk2, *wk3;
  float ck0, ck1, ck2, ck3;
  long m_s1, m_s0, m_off; int m_n0;   // row addressing of xk0 .. xk[n_ord - 2] (the copies); the last order uses a_*
  int groups;            // token groups of 16 NWV
  int debug;             // TANTE_HEAD_DEBUG ablation bits (timing only, results are wrong): 1 no epilogue memory, 2 no GELU, 4 no weight stream
  unsigned long long* stamps;   // -DTANTE_ABLATE builds only (tools/head_stamps.py), else null
};

#ifdef TANTE_ABLATE
unsigned long long* g_head_stamps = nullptr;
#define HEAD_STAMP(k)                                                                       \
  do {                                                                                      \
    if (A.stamps) {                                                                         \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                           \
      if (lane == 0) A.stamps[((long)blockIdx.x * 8 + wave) * 12 + (k)] = t_;               \
    }                                                                                       \
  } while (0)
#else
#define HEAD_STAMP(k)
#endif

constexpr int HB = 1024;   // bias block bytes per tile (one LDS-DMA pass)

// sizes for C = 32 * CB:  stage 1: K = C, 4 pixel tiles of C/2 rows;  stage 2: K = C/2, 4 tiles of C/4 rows;  stage 3: K = C/4, 64 rows
template <int CB>
struct HeadGeom {
  static constexpr int C = 32 * CB, C1 = C / 2, C2 = C / 4;
  static constexpr int CPR1 = CB * 4, CPR2 = CB * 2, CPR3 = CB;          // 16-byte chunks per row (K / 8)
  static constexpr int NS1 = C1 / 16, NS2 = C2 / 16;                     // 16-row sub-tiles
  static constexpr int KB2 = C1 / 32, KB3 = C2 / 32;                     // k-blocks of stages 2, 3
  static constexpr int T1 = C1 * CPR1 * 16 + HB, T2 = C2 * CPR2 * 16 + HB, T3 = 64 * CPR3 * 16 + HB;
  static constexpr int XS = 8 * 16 * CB * 64;                            // token-row staging of 8 waves (bf16), overlaid on the W2 tiles
  static constexpr int LDS = T3 + T1 + (4 * T2 > XS ? 4 * T2 : XS);      // C = 256: 9 + 65 + 4 x 17 = 142 KiB
};

// Work split: a workgroup owns 16 NWV tokens AND one stage-1 pixel p; everything downstream of that pixel (its 4 sub-pixels, their
// 2 x 2 x D values) depends on nothing else, so the four pixel workgroups of a token group never talk.  The workgroup's WHOLE weight
// stream -- W3, W1[p], the four W2 tiles: 142 KiB at C = 256 -- is resident in LDS: every LDS-DMA pass, the token rows and the
// epilogue's read-modify-write operands are issued in the first few hundred cycles, ONE wait + barrier follows, and the three stages
// then run without another barrier.  (Round 1 streamed the tiles through a two-slot ring with a wait + barrier per tile: seven
// L2 round trips in a row, each ~2.5 us because the compute between them is 16 - 64 MFMAs -- the kernel sat at 22 us for 54 MB of
// traffic, waves parked 59 % of the time.)  One workgroup per CU; blockIdx -> (group, p) keeps the four pixels of a group on one
// XCD (their output sectors interleave, the L2 merges them).
template <int CB, int NWV, bool MULTI = false>
__global__ __launch_bounds__(NWV * 64, 1) void fused_head_kernel(const HeadArgs A) {
  using G = HeadGeom<CB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [W3 | bias3][W1 p | bias1 p][W2 q | bias2 q] x 4
  char* w3s = smem;
  char* w1s = smem + G::T3;
  char* w2s = smem + G::T3 + G::T1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  const int p = (blockIdx.x & 31) >> 3;                          // stage-1 pixel (kh, kw) = (p >> 1, p & 1)
  const int grp = (blockIdx.x >> 5) * 8 + (blockIdx.x & 7);
  if (grp >= A.groups) return;
  HEAD_STAMP(0);
  f32x4 dsum[MULTI ? 4 : 1][4];      // MULTI: sum over the orders of coefficient x derivative, per sub-pixel q and channel tile ns
  f32x4 pre[2][4][2];                // the frame values the epilogue adds to (fetched during the first order)
  if constexpr (MULTI) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ns = 0; ns < 4; ++ns) dsum[q][ns] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
 // One order.  The LAST order's pass is a second copy of this body (IS_LAST): only there are the frame values the epilogue adds to
 // fetched -- 64 registers that would otherwise be live through every order's pass.
 auto run_order = [&](auto is_last_c, const int ord) {
  constexpr bool IS_LAST = decltype(is_last_c)::value;
  // this order's token rows, weight stream, coefficient and row addressing (named fields: a run-time index into the kernel arguments
  // would send the struct to scratch)
  const bool last_ord = !MULTI || IS_LAST;
  const float* Xp = !MULTI ? A.x : (ord == 0 ? A.xk0 : ord == 1 ? A.xk1 : ord == 2 ? A.xk2 : A.xk3);
  const char* Wp = !MULTI ? A.w : (ord == 0 ? A.wk0 : ord == 1 ? A.wk1 : ord == 2 ? A.wk2 : A.wk3);
  const float cord = !MULTI ? 1.0f : (ord == 0 ? A.ck0 : ord == 1 ? A.ck1 : ord == 2 ? A.ck2 : A.ck3);
  const long r_s1 = last_ord ? A.a_s1 : A.m_s1, r_s0 = last_ord ? A.a_s0 : A.m_s0, r_off = last_ord ? A.a_off : A.m_off;
  const int r_n0 = last_ord ? A.a_n0 : A.m_n0;
  if (MULTI && ord) __syncthreads();      // every wave is done with the previous order's weights before the DMA overwrites them
  if (!(A.debug & 4)) {
    hglds<NWV>(Wp + G::T3 + (long)p * G::T1, w1s, G::T1, tid);
    hglds<NWV>(Wp, w3s, G::T3, tid);
  }
  HEAD_STAMP(10);

  // Row -> (image, hp, wp) and row -> element offset WITHOUT per-lane integer division: the wave's 16 rows are consecutive, so the three
  // quotients are taken once on the wave-uniform first row and stepped (64-bit divisions by run-time values, one set per lane and load,
  // were 6 k of this kernel's 35 k clocks per wave: in-kernel stamps).
  const int HW = A.Hp * A.Wp;
  const unsigned n_rows = (unsigned)A.n_img * (unsigned)HW;
  const unsigned row0 = (unsigned)__builtin_amdgcn_readfirstlane((grp * NWV + wave) * 16);
  const unsigned img0 = row0 / (unsigned)HW, hw0 = row0 - img0 * (unsigned)HW;
  const unsigned aq0 = row0 / (unsigned)r_n0, ar0 = row0 - aq0 * (unsigned)r_n0;
  auto row_img_hw = [&](int j, int& im, int& hwv) {          // row0 + j -> (image, position in image)
    im = (int)img0; hwv = (int)hw0 + j;
    while (hwv >= HW) { hwv -= HW; ++im; }
  };
  auto row_offset = [&](int j) {                              // row0 + j -> element offset of its token row
    long q = aq0; int rem = (int)ar0 + j;
    while (rem >= r_n0) { rem -= r_n0; ++q; }
    return q * r_s1 + (long)rem * r_s0 + r_off;
  };
  const bool live = row0 + (unsigned)l15 < n_rows;
  int img, hw;
  row_img_hw(live ? l15 : 0, img, hw);
  if (!live) { img = 0; hw = 0; }
  const int hp = (int)(((float)hw + 0.5f) * __builtin_amdgcn_rcpf((float)A.Wp)), wp = hw - hp * A.Wp;   // exact: hw < 2^22
  // ---- the wave's 16 token rows: ROW-form loads (one instruction = the 1 KiB of one token; in B-operand layout an instruction touched
  // 16 rows x 64 B and the memory pipe worked it off lane by lane: 63 clocks against 17, tools/ubench/ta_cost.hip -- the token rows
  // alone held the pipe for 10 k of the wave's 35 k clocks), packed to bf16 into a private 8 KiB piece of the W2 region (whose LDS-DMA
  // is issued behind a barrier below), read back as B-operand fragments: 8-byte chunk c of row j at chunk c ^ (2 j), conflict-free
  // for the ds_write_b64 (lane = chunk) and the ds_read_b64 (32 lanes = 16 rows x 2 neighbouring chunks) alike.
  u32x4 xf[CB];   // the token as B-operand k-blocks, accumulator (k-permuted) order
  {
    static_assert(CB == 8 || CB == 4, "a token row is CB * 128 bytes: one or half a 64-lane instruction");
    constexpr int LPR = CB * 8;                       // lanes per token row (16 bytes each)
    constexpr int RPI = 64 / LPR;                     // rows per instruction
    f32x4 xraw[16 / RPI];
#pragma unroll
    for (int j = 0; j < 16 / RPI; ++j) {
      const int rj = RPI * j + lane / LPR;
      const long eo = row0 + (unsigned)rj < n_rows ? row_offset(rj) : r_off;      // dead rows read row 0 of the view (valid memory)
      xraw[j] = *(const f32x4*)(Xp + eo + 4 * (lane % LPR));
    }
    HEAD_STAMP(11);
    char* xs = w2s + wave * (16 * CB * 64);           // 16 rows x (32 CB) bf16
#pragma unroll
    for (int j = 0; j < 16 / RPI; ++j) {
      const int rj = RPI * j + lane / LPR, c = lane % LPR;
      u32x2 u;
      u[0] = pack_bf16x2(xraw[j][0], xraw[j][1]); u[1] = pack_bf16x2(xraw[j][2], xraw[j][3]);
      *(u32x2*)(xs + rj * (CB * 64) + ((c ^ ((2 * rj) & (LPR - 1))) << 3)) = u;
    }
#pragma unroll
    for (int b = 0; b < CB; ++b) {      // k-block b: features 32 b + 4 kk .. + 3 and 32 b + 16 + 4 kk .. + 3  = 8-byte chunks 8 b + kk, 8 b + 4 + kk
      const u32x2 lo = *(const u32x2*)(xs + l15 * (CB * 64) + (((8 * b + kk) ^ ((2 * l15) & (LPR - 1))) << 3));
      const u32x2 hi = *(const u32x2*)(xs + l15 * (CB * 64) + (((8 * b + 4 + kk) ^ ((2 * l15) & (LPR - 1))) << 3));
      xf[b] = live ? u32x4{lo[0], lo[1], hi[0], hi[1]} : u32x4{0u, 0u, 0u, 0u};
    }
  }
  __syncthreads();      // every wave has its fragments: the W2 tiles may land on the staging pieces
  if (!(A.debug & 4)) hglds<NWV>(Wp + G::T3 + 4L * G::T1, w2s, 4 * G::T2, tid);
  int xo1[CB], xo2[G::KB2], xo3[G::KB3];
#pragma unroll
  for (int b = 0; b < CB; ++b) xo1[b] = swz_chunk(l15, b * 4 + kk, G::CPR1) << 4;
#pragma unroll
  for (int b = 0; b < G::KB2; ++b) xo2[b] = swz_chunk(l15, b * 4 + kk, G::CPR2) << 4;
#pragma unroll
  for (int b = 0; b < G::KB3; ++b) xo3[b] = swz_chunk(l15, b * 4 + kk, G::CPR3) << 4;

  // epilogue addressing: the values the epilogue adds to (frame 0) are `last` for the first order and `out` itself for the later
  // ones; both sub-pixel pairs' operands are fetched here, with everything else that comes from memory.
  const long frame = (long)A.D * (A.Hp * 8) * (A.Wp * 8);
  const int Wout = A.Wp * 8;
  const float* rmw_src = A.last ? A.last + (long)img * A.last_bstride : A.out + (long)img * A.out_bstride;
  auto pix_of = [&](int q, int ns) {
    const int y0 = hp * 8 + (p >> 1) * 4 + (q >> 1) * 2, x0 = wp * 8 + (p & 1) * 4 + (q & 1) * 2;
    return ((long)(4 * ns + kk) * (A.Hp * 8) + y0) * Wout + x0;
  };
  // The epilogue works on sub-pixel PAIRS (q = 2 j, 2 j + 1 are horizontal neighbours): 4 pixels = 16 bytes per row, so every
  // read-modify-write instruction moves 16 bytes per lane instead of 8 (half the memory instructions of the kernel's epilogue).
  if constexpr (!MULTI || IS_LAST)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ns = 0; ns < 4; ++ns)
      if (4 * ns < A.D && live && 4 * ns + kk < A.D && !(A.debug & 1)) {
        const float* lp = rmw_src + pix_of(2 * j, ns);
        pre[j][ns][0] = *(const f32x4*)lp;
        pre[j][ns][1] = *(const f32x4*)(lp + Wout);
      }

  HEAD_STAMP(1);
  // W1 and W3 are complete here: every wave waited for its token rows, which it requested AFTER its share of the two tiles, before the
  // barrier above.  Stage 1 runs while the W2 tiles and the read-modify-write operands arrive.

  // ---- stage 1: pixel p; h1[kb] = B-operand k-blocks (C/2 channels) for stage 2 ------------------------------------------------
  u32x4 h1[G::KB2];
  {
    const float* bias1 = (const float*)(w1s + G::C1 * G::CPR1 * 16);
    f32x4 acc[G::NS1];
#pragma unroll
    for (int ns = 0; ns < G::NS1; ++ns) acc[ns] = *(const f32x4*)(bias1 + ns * 16 + kk * 4);
#pragma unroll
    for (int b = 0; b < CB; ++b)
#pragma unroll
      for (int ns = 0; ns < G::NS1; ++ns) acc[ns] = hmfma(*(const u32x4*)(w1s + (ns * 16 + l15) * G::CPR1 * 16 + xo1[b]), xf[b], acc[ns]);
#pragma unroll
    for (int b = 0; b < G::KB2; ++b)
      h1[b] = (A.debug & 2) ? hpack8(acc[2 * b], acc[2 * b + 1]) : hpack8(gelu4(acc[2 * b]), gelu4(acc[2 * b + 1]));
  }

  HEAD_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  HEAD_STAMP(3);
  __syncthreads();
  HEAD_STAMP(4);
  // ---- stages 2 + 3: sub-pixel tile q of W2, W3 ----------------------------------------------------------------------------
  const float* bias3 = (const float*)(w3s + 64 * G::CPR3 * 16);
  f32x4 dl[4];
  sfor<4>([&](auto qc) {
    constexpr int q = decltype(qc)::value;           // sub-pixel (kh2, kw2) = (q >> 1, q & 1)
    const char* wt = w2s + q * G::T2;
    const float* bias2 = (const float*)(wt + G::C2 * G::CPR2 * 16);
    f32x4 acc2[G::NS2];
#pragma unroll
    for (int ns = 0; ns < G::NS2; ++ns) acc2[ns] = *(const f32x4*)(bias2 + ns * 16 + kk * 4);
#pragma unroll
    for (int b = 0; b < G::KB2; ++b)
#pragma unroll
      for (int ns = 0; ns < G::NS2; ++ns) acc2[ns] = hmfma(*(const u32x4*)(wt + (ns * 16 + l15) * G::CPR2 * 16 + xo2[b]), h1[b], acc2[ns]);
    u32x4 h2[G::KB3];
#pragma unroll
    for (int b = 0; b < G::KB3; ++b) h2[b] = (A.debug & 2) ? hpack8(acc2[2 * b], acc2[2 * b + 1]) : hpack8(gelu4(acc2[2 * b]), gelu4(acc2[2 * b + 1]));
    // stage 3: rows n3 = (co, kh3, kw3) = 16 ns + 4 kk + r  ->  channel co = 4 ns + kk, r = (kh3, kw3)
#pragma unroll
    for (int ns = 0; ns < 4; ++ns) {
      if (4 * ns < A.D) {   // uniform: this 16-row tile holds real channels
        f32x4 d = *(const f32x4*)(bias3 + ns * 16 + kk * 4);
#pragma unroll
        for (int b = 0; b < G::KB3; ++b) d = hmfma(*(const u32x4*)(w3s + (ns * 16 + l15) * G::CPR3 * 16 + xo3[b]), h2[b], d);
        if ((A.debug & 1) && d[0] != 1.2345f) continue;
        if constexpr (MULTI) {
          dsum[q][ns] += d * cord;                     // the frame is touched once, after the last order (below)
        } else if constexpr ((q & 1) == 0) {
          dl[ns] = d;                                  // left half of the pair: kept until its right neighbour exists
        } else if (live && 4 * ns + kk < A.D) {
          const long pix = pix_of(q - 1, ns);
          float* o0 = A.out + (long)img * A.out_bstride + pix;
          const float c0 = A.coef[0];
          const f32x4 t0 = f32x4{dl[ns][0], dl[ns][1], d[0], d[1]}, t1 = f32x4{dl[ns][2], dl[ns][3], d[2], d[3]};   // rows y0, y0 + 1
          *(f32x4*)o0 = pre[q >> 1][ns][0] + t0 * c0;
          *(f32x4*)(o0 + Wout) = pre[q >> 1][ns][1] + t1 * c0;
#pragma unroll
          for (int i = 1; i < 8; ++i) {   // further output frames (output_length > 1 / adaptive dt): plain read-modify-write
            if (i >= A.n_out) break;
            float* o = o0 + (long)i * frame;
            const float* bp = A.last ? A.last + (long)img * A.last_bstride + pix : o;
            const f32x4 r0 = *(const f32x4*)bp, r1 = *(const f32x4*)(bp + Wout);
            const float c = A.coef[i];
            *(f32x4*)o = r0 + t0 * c;
            *(f32x4*)(o + Wout) = r1 + t1 * c;
          }
        }
      }
    }
    HEAD_STAMP(5 + q);
  });
 };   // run_order
 if constexpr (MULTI) {
   for (int ord = 0; ord + 1 < A.n_ord; ++ord) run_order(std::false_type{}, ord);
   run_order(std::true_type{}, A.n_ord - 1);
 } else {
   run_order(std::true_type{}, 0);
 }
  if constexpr (MULTI) {
    const float* bias_unused = nullptr; (void)bias_unused;
    const long frame = (long)A.D * (A.Hp * 8) * (A.Wp * 8); (void)frame;
    const int Wout = A.Wp * 8;
    const int HW = A.Hp * A.Wp;
    const unsigned n_rows = (unsigned)A.n_img * (unsigned)HW;
    const unsigned row0 = (unsigned)__builtin_amdgcn_readfirstlane((grp * NWV + wave) * 16);
    const bool live = row0 + (unsigned)l15 < n_rows;
    unsigned im = row0 / (unsigned)HW; int hwv = (int)(row0 - im * (unsigned)HW) + (live ? l15 : 0);
    while (hwv >= HW) { hwv -= HW; ++im; }
    if (!live) { im = 0; hwv = 0; }
    const int hp = (int)(((float)hwv + 0.5f) * __builtin_amdgcn_rcpf((float)A.Wp)), wp = hwv - hp * A.Wp;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ns = 0; ns < 4; ++ns)
        if (4 * ns < A.D && live && 4 * ns + kk < A.D) {
          const int y0 = hp * 8 + (p >> 1) * 4 + j * 2, x0 = wp * 8 + (p & 1) * 4;
          float* o0 = A.out + (long)im * A.out_bstride + ((long)(4 * ns + kk) * (A.Hp * 8) + y0) * Wout + x0;
          const f32x4 dl_ = dsum[2 * j][ns], dr_ = dsum[2 * j + 1][ns];
          *(f32x4*)o0 = pre[j][ns][0] + f32x4{dl_[0], dl_[1], dr_[0], dr_[1]};
          *(f32x4*)(o0 + Wout) = pre[j][ns][1] + f32x4{dl_[2], dl_[3], dr_[2], dr_[3]};
        }
  }
#ifdef TANTE_ABLATE
  if (A.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HEAD_STAMP(9);
  }
#endif
}

// ---- stream packing: one thread block per tile -------------------------------------------------------------------------------
__device__ __forceinline__ int hkperm(int p) {   // position p of a k-permuted row holds source feature c
  const int blk = p >> 5, qq = p & 31, kk = qq >> 3, dt = (qq >> 2) & 1, r = qq & 3;
  return blk * 32 + dt * 16 + kk * 4 + r;
}

// ConvTranspose2d weights (Cin, Cout, 2, 2) of the three stages -> [W3][W1 x4][W2 x4] tiles, rows k-permuted
__global__ void pack_head_stream_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                        const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3, int C, int D,
                                        char* __restrict__ dst) {
  const int C1 = C / 2, C2 = C / 4;
  const int cpr1 = C / 8, cpr2 = C1 / 8, cpr3 = C2 / 8;
  const long T1 = (long)C1 * cpr1 * 16 + HB, T2 = (long)C2 * cpr2 * 16 + HB, T3 = 64L * cpr3 * 16 + HB;
  const int t = blockIdx.x;   // 0: W3, 1..4: W1 pixel tiles, 5..8: W2 sub-pixel tiles
  const float* w; const float* b; int rows, cpr, K, Cout, pix; char* base;
  if (t == 0) { w = w3; b = b3; rows = 64; cpr = cpr3; K = C2; Cout = D; pix = -1; base = dst; }
  else if (t <= 4) { w = w1; b = b1; rows = C1; cpr = cpr1; K = C; Cout = C1; pix = t - 1; base = dst + T3 + (long)(t - 1) * T1; }
  else { w = w2; b = b2; rows = C2; cpr = cpr2; K = C1; Cout = C2; pix = t - 5; base = dst + T3 + 4 * T1 + (long)(t - 5) * T2; }
  for (int idx = threadIdx.x; idx < rows * cpr; idx += blockDim.x) {
    const int r = idx / cpr, c = idx % cpr;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = hkperm(c * 8 + e);   // input channel ci
      float xv = 0.f;
      if (pix >= 0) {                    // row = output channel co of pixel (kh, kw) = pix
        xv = w[(((long)k * Cout + r) * 2 + (pix >> 1)) * 2 + (pix & 1)];
      } else if (r < 4 * D) {            // stage 3: row n3 = (co, kh3, kw3)
        xv = w[(long)k * (D * 4) + r];
      }
      v[e] = xv;
    }
    u32x4 o;
    o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
    *((u32x4*)base + (long)r * cpr + swz_chunk(r, c, cpr)) = o;
  }
  float* bias = (float*)(base + (long)rows * cpr * 16);
  for (int r = threadIdx.x; r < 256; r += blockDim.x) {
    float v = 0.f;
    if (pix >= 0) { if (r < rows) v = b[r]; }
    else if (r < 4 * D) v = b[r / 4];
    bias[r] = v;
  }
}

template <int CB, int NWV, bool MULTI>
void launch_head_nw(HeadArgs A, hipStream_t s) {
  using G = HeadGeom<CB>;
  static TantePerDevice attr;
  attr.once([&] {
    (void)hipFuncSetAttribute((const void*)fused_head_kernel<CB, NWV, MULTI>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
  });
  const long rows = (long)A.n_img * A.Hp * A.Wp;
  A.groups = (int)((rows + NWV * 16 - 1) / (NWV * 16));
  const unsigned grid = (unsigned)((A.groups + 7) / 8) * 32;   // 8 token groups x 4 pixels per 32 consecutive workgroups
  hipLaunchKernelGGL((fused_head_kernel<CB, NWV, MULTI>), dim3(grid), dim3(NWV * 64), G::LDS, s, A);
}

template <int CB, bool MULTI = false>
void launch_head(HeadArgs A, hipStream_t s) {
  // 128-token groups (8 waves) once they still give every CU a workgroup; 64-token groups for small batches
  const long rows = (long)A.n_img * A.Hp * A.Wp;
  const int force = tante_opt("TANTE_HEAD_WAVES", 0);
  const bool wide = force ? force == 8 : rows >= 128 * 56;
  if (wide) launch_head_nw<CB, 8, MULTI>(A, s);
  else launch_head_nw<CB, 4, MULTI>(A, s);
}

}  // namespace

#ifdef TANTE_ABLATE
extern "C" void tante_head_set_stamps(unsigned long long* p) { g_head_stamps = p; }
#endif

extern "C" int tante_head_fused_supported(int C, int D) { return (C == 128 || C == 256) && D >= 1 && D <= 16; }

extern "C" int64_t tante_head_stream_bytes(int C) {
  const long C1 = C / 2, C2 = C / 4;
  return (64L * (C2 / 8) * 16 + HB) + 4 * (C1 * (long)(C / 8) * 16 + HB) + 4 * (C2 * (C1 / 8) * 16 + HB);   // HB = 1024
}

extern "C" int tante_pack_head(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int C, int D,
                               void* head_stream, void* stream) {
  if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !head_stream) TANTE_FAIL(-1, "tante_pack_head: null pointer");
  if (!tante_head_fused_supported(C, D)) TANTE_FAIL(-2, "tante_pack_head: unsupported C=%d D=%d", C, D);
  hipLaunchKernelGGL(pack_head_stream_kernel, dim3(9), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, w3, b3, C, D, (char*)head_stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_head_fused(const float* x, int32_t a_n0, int64_t a_s1, int64_t a_s0, int64_t a_off, int n_img, int Hp, int Wp, int C, int D,
                                const void* head_stream, float* out, int64_t out_bstride, int n_out, const float* coefs, const float* last,
                                int64_t last_bstride, void* stream) {
  if (!x || !head_stream || !out || !coefs) TANTE_FAIL(-1, "tante_head_fused: null pointer");
  if (!tante_head_fused_supported(C, D)) TANTE_FAIL(-2, "tante_head_fused: unsupported C=%d D=%d", C, D);
  if (n_out < 1 || n_out > 8 || a_n0 <= 0 || n_img <= 0 || Hp <= 0 || Wp <= 0) TANTE_FAIL(-1, "tante_head_fused: bad shape");
  if (a_s1 % 4 || a_s0 % 4 || a_off % 4 || out_bstride % 2 || last_bstride % 2 || ((uintptr_t)x % 16) || ((uintptr_t)out % 8) ||
      (last && ((uintptr_t)last % 8)))
    TANTE_FAIL(-1, "tante_head_fused: alignment");
  HeadArgs A;
  A.x = x; A.a_n0 = a_n0; A.a_s1 = a_s1; A.a_s0 = a_s0; A.a_off = a_off;
  A.n_img = n_img; A.Hp = Hp; A.Wp = Wp; A.D = D;
  A.w = (const char*)head_stream; A.out = out; A.out_bstride = out_bstride; A.n_out = n_out;
  A.last = last; A.last_bstride = last_bstride;
  for (int i = 0; i < 8; ++i) A.coef[i] = i < n_out ? coefs[i] : 0.f;
  A.n_ord = 1;
  A.xk0 = A.xk1 = A.xk2 = A.xk3 = nullptr; A.wk0 = A.wk1 = A.wk2 = A.wk3 = nullptr; A.ck0 = A.ck1 = A.ck2 = A.ck3 = 0.f;
  A.m_n0 = 1; A.m_s1 = A.m_s0 = A.m_off = 0;
  A.debug = tante_ablate_env("TANTE_HEAD_DEBUG");  // -DTANTE_ABLATE builds only
#ifdef TANTE_ABLATE
  A.stamps = g_head_stamps;
#else
  A.stamps = nullptr;
#endif
  if (C == 128) launch_head<4>(A, (hipStream_t)stream);
  else launch_head<8>(A, (hipStream_t)stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}

/* Every Taylor order's derivative head in ONE launch, one prediction frame:  out = last + sum_k coefs[k] * head_k(rows_k)
 * (tante.py:145-154, 165-171 with output_length = 1).  rows_k, k < n_ord - 1: dense (n_img Hp Wp, C) fp32 copies of the last-slot token
 * rows as backbone k left them (the stream is updated in place by the later backbones); the last order reads the stream itself through
 * (a_n0, a_s1, a_s0, a_off) like tante_head_fused.  The frame is read (from `last`) and written once instead of once per order. */
static int head_fused_multi_impl(int n_ord, const float* const* rows, const void* const* head_streams, const float* coefs, int32_t a_n0,
                                 int64_t a_s1, int64_t a_s0, int64_t a_off, int n_img, int Hp, int Wp, int C, int D, float* out,
                                 int64_t out_bstride, const float* last, int64_t last_bstride, void* stream, bool dense_copies);

extern "C" int tante_head_fused_multi(int n_ord, const float* const* rows, const void* const* head_streams, const float* coefs, int32_t a_n0,
                                      int64_t a_s1, int64_t a_s0, int64_t a_off, int n_img, int Hp, int Wp, int C, int D, float* out,
                                      int64_t out_bstride, const float* last, int64_t last_bstride, void* stream) {
  return head_fused_multi_impl(n_ord, rows, head_streams, coefs, a_n0, a_s1, a_s0, a_off, n_img, Hp, Wp, C, D, out, out_bstride, last, last_bstride, stream, true);
}

/* The same with EVERY order's rows addressed like the last one's, rows[k] = the whole residual stream backbone k left (the backbones
 * write their streams into buffers of their own, tante_axis_hw_oop: nothing is copied aside). */
extern "C" int tante_head_fused_multi_streams(int n_ord, const float* const* rows, const void* const* head_streams, const float* coefs, int32_t a_n0,
                                              int64_t a_s1, int64_t a_s0, int64_t a_off, int n_img, int Hp, int Wp, int C, int D, float* out,
                                              int64_t out_bstride, const float* last, int64_t last_bstride, void* stream) {
  return head_fused_multi_impl(n_ord, rows, head_streams, coefs, a_n0, a_s1, a_s0, a_off, n_img, Hp, Wp, C, D, out, out_bstride, last, last_bstride, stream, false);
}

static int head_fused_multi_impl(int n_ord, const float* const* rows, const void* const* head_streams, const float* coefs, int32_t a_n0,
                                 int64_t a_s1, int64_t a_s0, int64_t a_off, int n_img, int Hp, int Wp, int C, int D, float* out,
                                 int64_t out_bstride, const float* last, int64_t last_bstride, void* stream, bool dense_copies) {
  if (!rows || !head_streams || !coefs || !out || !last) TANTE_FAIL(-1, "tante_head_fused_multi: null pointer");
  if (n_ord < 1 || n_ord > 4) TANTE_FAIL(-2, "tante_head_fused_multi: 1 .. 4 orders");
  if (!tante_head_fused_supported(C, D)) TANTE_FAIL(-2, "tante_head_fused_multi: unsupported C=%d D=%d", C, D);
  if (a_n0 <= 0 || n_img <= 0 || Hp <= 0 || Wp <= 0) TANTE_FAIL(-1, "tante_head_fused_multi: bad shape");
  if (a_s1 % 4 || a_s0 % 4 || a_off % 4 || out_bstride % 2 || last_bstride % 2 || ((uintptr_t)out % 8) || ((uintptr_t)last % 8))
    TANTE_FAIL(-1, "tante_head_fused_multi: alignment");
  for (int k = 0; k < n_ord; ++k)
    if (!rows[k] || !head_streams[k] || ((uintptr_t)rows[k] % 16)) TANTE_FAIL(-1, "tante_head_fused_multi: order %d: null or misaligned rows / stream", k);
  HeadArgs A;
  A.x = rows[n_ord - 1]; A.a_n0 = a_n0; A.a_s1 = a_s1; A.a_s0 = a_s0; A.a_off = a_off;
  A.n_img = n_img; A.Hp = Hp; A.Wp = Wp; A.D = D;
  A.w = (const char*)head_streams[n_ord - 1]; A.out = out; A.out_bstride = out_bstride; A.n_out = 1;
  A.last = last; A.last_bstride = last_bstride;
  for (int i = 0; i < 8; ++i) A.coef[i] = 0.f;
  A.n_ord = n_ord;
  const float* xs[4] = {nullptr, nullptr, nullptr, nullptr};
  const char* ws[4] = {nullptr, nullptr, nullptr, nullptr};
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < n_ord; ++k) { xs[k] = rows[k]; ws[k] = (const char*)head_streams[k]; cs[k] = coefs[k]; }
  A.xk0 = xs[0]; A.xk1 = xs[1]; A.xk2 = xs[2]; A.xk3 = xs[3];
  A.wk0 = ws[0]; A.wk1 = ws[1]; A.wk2 = ws[2]; A.wk3 = ws[3];
  A.ck0 = cs[0]; A.ck1 = cs[1]; A.ck2 = cs[2]; A.ck3 = cs[3];
  if (dense_copies) { A.m_n0 = n_img * Hp * Wp; A.m_s1 = 0; A.m_s0 = C; A.m_off = 0; }      // the copies: dense rows
  else { A.m_n0 = a_n0; A.m_s1 = a_s1; A.m_s0 = a_s0; A.m_off = a_off; }                       // whole streams, addressed like the last order's
  A.debug = 0;
  A.stamps = nullptr;
  if (C == 128) launch_head<4, true>(A, (hipStream_t)stream);
  else launch_head<8, true>(A, (hipStream_t)stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}
