// Fused TransformerBlock (attn_backbone.py:59-83), bf16 path, C = 256 / 8 heads x 32 / hidden 256 -- "feature-sliced" form.
//
//   x <- x + Wo . Attention(LayerNorm1(x)) + bo ;   x <- x + W2 . gelu_tanh(W1 . LayerNorm2(x) + b1) + b2
//
// One workgroup = 8 waves = up to 128 tokens (whole sequences).  The round-1 kernel (block_fused.hip) gave every wave 16 tokens and
// all features: each weight fragment was read from LDS by all 8 waves and fed ONE 16-token MFMA, so LDS reads (1 KiB per MFMA and
// wave) ran as long as the matrix pipe itself and the vector issue port, busy with both, set the time (0.25 of the bf16 peak).
// Here the roles are turned around: a wave owns a SLICE OF THE OUTPUT FEATURES for all tokens of the workgroup,
//
//     QKV + attention : wave h = head h (q, k, v rows of that head, 96 features)
//     out-proj, fc1, fc2 : wave w = output features 32 w .. 32 w + 31
//
// so a wave's weights are its own: they go global/L2 -> registers directly (pre-packed in consumption order, 1 KiB per coalesced
// load, no LDS, no barrier) and every weight fragment feeds 8 token tiles.  What the waves share are the ACTIVATIONS, as bf16
// images [token][256] in LDS (XOR-swizzled 16-byte chunks, conflict-free ds_read_b128 fragments): LayerNorm-ed x, the attention
// output, LayerNorm2-ed x1, the GELU-ed hidden.  One image fragment feeds 2-4 MFMAs instead of 1, a launch has 5 barriers instead
// of 21, and there is no weight stream through LDS at all.  The fp32 residual slice of a wave (32 features x 128 tokens) rides the
// out-proj / fc2 accumulators.  Attention operands chain through registers exactly as in round 1 (accumulator pairs packed to bf16
// ARE the next MFMA's operands): S^T = K Q^T, softmax over accumulator rows, O^T = V^T P^T.
//
// Two algebraic savings on the way (both exact): the key bias never reaches the output (it shifts every score of a query by the
// same amount, softmax removes it), and the value bias commutes with the attention average (rows of P sum to 1), so it is folded
// into the out-proj bias at pack time:  bo' = bo + Wo . bv.
#include "common.hip.h"
#include "fused_common.hip.h"
#include "fs_common.hip.h"
#include "block_sliced.h"

// The FS_EXP_* switches are TIMING experiments: they skip work and produce wrong results by design.  They compile only into the diagnostic
// build (-DTANTE_ABLATE: tools/_ab/ libraries, never tante_amd/lib/libtante_hip.so) -- ADVICE round 5.
#if (defined(FS_EXP_NO_SAVE) || defined(FS_EXP_LN_ID) || defined(FS_EXP_GELU_ID) || defined(FS_EXP_NO_RESID) || defined(FS_EXP_NO_STORE) || \
     defined(FS_EXP_TPROP_NO_STORE) || defined(FS_EXP_ROT) || defined(FS_EXP_NO_WLOAD) || defined(FS_EXP_MFMA32)) && !defined(TANTE_ABLATE)
#error "FS_EXP_* timing experiments produce wrong results on purpose: build them with -DTANTE_ABLATE (tools/build_variant.sh), never into the product library"
#endif

namespace {


struct FsArgs {
  float* x;
  const char* w;
  TanteSeq sq;
  int causal;
  float eps;
  int spw;          // sequences per workgroup
  unsigned magic;   // ceil(65536 / L): slot / L == (slot * magic) >> 16 for slot < 128
  // training forward (block_fs_kernel<..., TRAIN = true>): the block's output goes to `out` (x stays intact: LayerNorm's backward needs
  // it) and everything the backward pass reads is stored on the way, dense (tokens, features) rows in the layouts of the unfused ops
  // (autograd.py): LayerNorm outputs + (mean, rstd), the packed q | k | v projection (q NOT pre-scaled, biases included), the
  // attention output, the residual after the attention half, fc1's pre-activation and its GELU.  Dropout masks are
  // dropout_keep(seed, index, p) with the indices of tante_attention_dropout / the GEMM epilogue / tante_dropout_bwd.
  float* out;
  unsigned short *xh1, *qkv, *o, *xh2, *hpre, *act;
  float *st1, *x1, *st2;
  float p_drop;
  unsigned long long seed_attn, seed_out, seed_mlp;
  const unsigned long long* seed_mix;   // device word XOR-ed into the three seeds (HIP-graph replays of a train step), or null
  unsigned long long* stamps;   // diagnostic builds (-DTANTE_ABLATE) only: per-wave s_memtime at the phase boundaries, else null
  // inference, L = 4 (the T letter): the temporal propagator y_t = x_t + b2[t] + sum_j w2[t][j] gelu(b1[j] + sum_a w1[j][a] x_a)
  // (attn_backbone.py:144-145) applied to the rows as they are loaded for LayerNorm1: w1 (4 x 4), b1, w2 (4 x 4), b2 = 40 floats, or null
  const float* tprop;
  // entry skew (inference, one resident round of two workgroups per CU): the workgroups of the second half of the grid -- the second
  // resident of every CU -- sleep `skew` x 512 cycles before their first load, so that the two residents of a CU stop sharing every phase
  int skew;
};

// In-kernel stamps (cdna_hip_programming.md 7): a -DTANTE_ABLATE build records the shader clock at every phase boundary of every wave
// into a buffer of its own (tools/fs_stamps.py reads the phase SHARES from it); in the product build the macro is empty.
#ifdef TANTE_ABLATE
#define FS_STAMP(k)                                                                                        \
  do {                                                                                                     \
    if (A.stamps) {                                                                                        \
      unsigned long long t_;                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
      __builtin_amdgcn_sched_barrier(0);                                                                   \
      if (lane == 0) A.stamps[((long)vblock * 8 + wave) * 20 + (k)] = t_;                               \
    }                                                                                                      \
  } while (0)
unsigned long long* g_fs_stamps = nullptr;
#else
#define FS_STAMP(k)
#endif

// a / n for 0 <= a < 2^23 by the float reciprocal plus one correction each way (an integer division is ~30 VALU instructions and a
// lane needs two per token; the host checks nseq < 2^23)
__device__ __forceinline__ void fs_divmod(int a, int n, float rn, int& q, int& r) {
  q = (int)((float)a * rn);
  r = a - q * n;
  if (r >= n) { ++q; r -= n; }
  if (r < 0) { --q; r += n; }
}
struct FsSeqMap {
  float rs0, rl0;
  __device__ __forceinline__ explicit FsSeqMap(const TanteSeq& q) : rs0(__builtin_amdgcn_rcpf((float)q.n_s0)), rl0(__builtin_amdgcn_rcpf((float)q.n_l0)) {}
  __device__ __forceinline__ long token(const TanteSeq& q, int s, int l) const {
    int a, b, c, d;
    fs_divmod(s, q.n_s0, rs0, a, b);
    fs_divmod(l, q.n_l0, rl0, c, d);
    return (long)a * q.S1 + (long)b * q.S0 + (long)c * q.P1 + (long)d * q.P0;
  }
};


// ---- the temporal propagator on the matrix pipe (kernels with TPROP) -----------------------------------------------------------------
//   y_t = x_t + b2[t] + sum_j w2[t][j] gelu(b1[j] + sum_a w1[j][a] x_a)        (attn_backbone.py:144-145; erf GELU, per position and channel)
// xs[t] = 4 channels of time step t of ONE sequence, all four in the same lane.  v_mfma_f32_4x4x1_16B_f32: 16 blocks of (4 x 1)(1 x 4)
// outer products, lane 4 b + j = column j of block b -- every lane is its own column, its four accumulator registers are the four
// hidden units (first product) / time steps (second), and lane 4 b + i supplies row i of the 4 x 4 weight, the same for every block.
// fp32 in, fp32 accumulate (a k-ordered fma chain from the bias), no cross-lane traffic.  tp = w1 (4 x 4) | b1 | w2 (4 x 4) | b2.
struct TpropW {
  float w1c[4], w2c[4];
  f32x4 b1v, b2v;
};
__device__ __forceinline__ void tprop_weights(const float* __restrict__ tp, int lane, TpropW& w) {
#pragma unroll
  for (int a = 0; a < 4; ++a) { w.w1c[a] = tp[4 * (lane & 3) + a]; w.w2c[a] = tp[20 + 4 * (lane & 3) + a]; }
  w.b1v = f32x4{tp[16], tp[17], tp[18], tp[19]};
  w.b2v = f32x4{tp[36], tp[37], tp[38], tp[39]};
}
__device__ __forceinline__ void tprop_apply(const TpropW& w, f32x4 (&xs)[4]) {
  f32x4 hid[4], yy[4];      // [channel component e] -> registers = hidden units / time steps
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f32x4 h = w.b1v;
#pragma unroll
    for (int a = 0; a < 4; ++a) h = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w1c[a], xs[a][e], h, 0, 0, 0);
    hid[e] = h;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) hid[e] = gelu_poly4<false>(hid[e]);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f32x4 acc = f32x4{xs[0][e], xs[1][e], xs[2][e], xs[3][e]} + w.b2v;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w.w2c[j], hid[e][j], acc, 0, 0, 0);
    yy[e] = acc;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) xs[t] = f32x4{yy[0][t], yy[1][t], yy[2][t], yy[3][t]};
}

// largest divisor of ntt that keeps  group x nk  score tiles within 8 (32 accumulator registers)
constexpr int fs_group(int ntt, int nk) {
  int g = 1;
  for (int d = 1; d <= ntt; ++d)
    if (ntt % d == 0 && d * nk <= 8) g = d;
  return g;
}

// TPS = token tiles per sequence when sequences are tile-aligned (L = 16 TPS), 1 also for L | 16 (several sequences per tile,
// block-diagonal mask), 0 = any L <= 16 NTT (every key tile, element masks).  NTT = token tiles per workgroup.  NW = waves per
// workgroup: 8 (one workgroup per CU, a wave = one head / 32 output features) or 4 (two independent workgroups per CU, a wave = two
// heads / 64 output features: while one workgroup is in a VALU or memory phase the other one's MFMAs have the matrix pipes).
//
// G = 2 ("paired" form, round 3): ONE 8-wave workgroup per CU made of two independent 4-wave GROUPS, each the 4-wave kernel on its own
// 64 (48) tokens with its own LDS images -- the same instruction stream, but group 1 runs it ONE BARRIER SEGMENT behind group 0 (it
// takes one extra workgroup barrier at entry, group 0 one at exit) and two alignment barriers split the mixed segments, so that the
// segments alternate  LayerNorm1 | q,k,v GEMMs | attention | out-proj | LayerNorm2 | fc1 | GELU | fc2 + store  = vector, MATRIX, vector,
// MATRIX, ...  Waves w and w + 4 share a SIMD: whenever one group is in a matrix segment its partner on the SIMD is in a vector / memory
// segment, by construction and for the whole launch (two independent workgroups start together and stay in step: MFMA-only phases
// beside MFMA-only phases, with the matrix pipe idle through every LayerNorm / softmax / GELU phase -- SQ_VALU_MFMA_COEXEC 11 %).
template <int TPS, int NTT, int NW, bool TRAIN, int G = 1, bool TPROP = false>
__global__ __launch_bounds__(64 * NW * G, 2) void block_fs_kernel(FsArgs A) {
  static_assert(!TPROP || (TPS == 1 && !TRAIN), "the fused temporal propagator: T letter (L = 4), inference");
  static_assert(G == 1 || (G == 2 && NW == 4), "the paired form is two 4-wave groups");
  constexpr int RT = 16 / NW;            // 16-row output tiles per wave
  constexpr int HPW = RT / 2;            // heads per wave
#ifndef FS_PF8
#define FS_PF8 3
#endif
#ifndef FS_PF4
#define FS_PF4 2
#endif
  // weight k-steps in flight.  The register-heaviest instantiations (training forward at 4 token tiles, the element-masked any-L forms,
  // the 8-tile 8-wave forms) spilled 20 - 290 bytes per lane at the full depth: they run one k-step shallower (16 RT fewer registers)
#ifdef FS_NO_TIGHT      // (A/B: every instantiation at the full depth, as in round 2)
  constexpr bool TIGHT = false;
#else
  constexpr bool TIGHT = (NW == 4 && NTT == 4 && (TRAIN || TPS == 0)) || (NW == 8 && NTT == 8);
#endif
  constexpr int PF = (NW == 8 ? FS_PF8 : FS_PF4) - (TIGHT ? 1 : 0);
  constexpr int IMG = 16 * NTT * FS_ROW;
  constexpr int LDS_G = 2 * IMG + 16 * NTT * NW * 8 + FS_BIAS_FLOATS * 4;      // bytes of LDS per group
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  const int grp = G == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);      // wave-uniform, provably
  char* const smem = smem_all + grp * LDS_G;
  char* const bufA = smem;                 // LayerNorm1(x), later LayerNorm2(x1)
  char* const bufB = smem + IMG;           // attention output, later the GELU-ed hidden
  char* const stat = smem + 2 * IMG;       // float2 [16 NTT tokens][NW waves]
  float* const lbias = (float*)(stat + 16 * NTT * NW * 8);   // the 4 x 256 biases (a global load per GEMM start would expose its latency)
  const int tid = threadIdx.x & (64 * NW - 1), lane = tid & 63, kk = lane >> 4, l15 = lane & 15;      // thread within the group
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef FS_XCD_SAMPLE   // experiment: consecutive workgroup ids go to consecutive XCDs; give XCD k a CONTIGUOUS eighth of the sequences (at B = 8:
  // sample k, in every letter's launch and in the propagator pass), so that a launch finds the rows the previous one wrote in its own L2
  const int vblock = (G == 1 && gridDim.x % 8 == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x * G + grp;
#else
  const int vblock = (int)blockIdx.x * G + grp;      // the group's index = the workgroup index of the unpaired form
#endif
#ifndef FS_SKEW
#define FS_SKEW 1      // 0: experiment -- the two groups of the paired form in step (same barriers, no offset, no alignment barriers)
#endif
  if constexpr (G == 2 && FS_SKEW) {
    if (grp == 1) __builtin_amdgcn_s_barrier();      // one segment behind group 0 from here on (group 0 takes its extra barrier at exit)
  }
#ifdef FS_PRIO
  // experiment: the two workgroups of a CU run in lockstep through MFMA-only and VALU-only phases.  Waves in odd hardware slots (HW_ID
  // bits 3:0) get issue priority for the whole kernel, so that the pair drifts apart instead of sharing every phase
  if constexpr (G == 1) {
    if ((__builtin_amdgcn_s_getreg(0x1804) & 1u) == (FS_PRIO & 1)) __builtin_amdgcn_s_setprio(3);
  }
#endif
  if constexpr (!TRAIN && G == 1) {
    if (A.skew > 0 && blockIdx.x >= (gridDim.x >> 1)) {
      for (int i = 0; i < A.skew; ++i) __builtin_amdgcn_s_sleep(8);
    }
  }
  float* const x = A.x;
  const unsigned long long smix = (TRAIN && A.seed_mix) ? *A.seed_mix : 0ull;      // per-step word of a replayed train step
  const unsigned long long sd_attn = A.seed_attn ^ smix;
  const int L = A.sq.L;
  const int seq0 = vblock * A.spw;
  const int nlive = min(A.spw, A.sq.nseq - seq0) * L;     // live token slots of this workgroup (the rest of spw * L is dead)
  const int nslot = A.spw * L;                            // slots in use by whole sequences (<= 16 NTT)

  const FsW wq = fs_wstream(A.w, (unsigned)((RT * wave) * FS_FRAG + lane * 16));   // this wave's row tiles of (matrix 0, k-step 0), this lane's 16 bytes
  u32x4 wb[PF + 1][RT];                                             // the weight stream's register ring
  // ---- slot -> token index (-1 = dead): every wave works the table out for itself, one or two slots per lane (the two divisions of
  // the axis regrouping), and hands the entries around with ds_bpermute -- no LDS table, no barrier in front of the first loads ----
  FS_STAMP(0);
#ifdef TANTE_ABLATE
  if (A.stamps && lane == 0) {   // where this wave runs: HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20), and the 100 MHz wall clock
    A.stamps[((long)vblock * 8 + wave) * 20 + 16] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    A.stamps[((long)vblock * 8 + wave) * 20 + 17] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    A.stamps[((long)vblock * 8 + wave) * 20 + 18] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  int tslot[2];
  {
    const FsSeqMap smap(A.sq);
#pragma unroll
    for (int h = 0; h < (16 * NTT > 64 ? 2 : 1); ++h) {
      const int slot = lane + 64 * h;
      const int s = (int)(((unsigned)slot * A.magic) >> 16), p = slot - s * L;
      tslot[h] = slot < nlive ? (int)smap.token(A.sq, seq0 + s, p) : -1;
    }
  }
  auto tok_of = [&](int slot) { return __shfl(slot < 64 ? tslot[0] : tslot[16 * NTT > 64 ? 1 : 0], slot & 63); };
  // the 4 x 256 biases go to LDS here; their first reader sits behind barrier 1
  for (int i = tid; i < FS_BIAS_FLOATS / 4; i += 64 * NW) *(f32x4*)(lbias + 4 * i) = *(const f32x4*)((const float*)(A.w + FS_W_BYTES) + 4 * i);
  FS_STAMP(1);
  int tokidx[NTT];   // this lane's token of every tile in accumulator layout (column l15 of tile tt)
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) tokidx[tt] = tok_of(16 * tt + l15);

  // ---- this wave's fp32 slice of x ([16 NTT tokens][16 RT features]) between memory and the accumulator layout ----------------------
  // In accumulator layout lane (l15, kk) holds 16 bytes of token l15: neighbouring lanes are different ROWS, and the memory pipe then
  // works a 64-lane instruction off lane by lane (tools/ubench/ta_cost.hip: 63 clocks against 17 for row-contiguous lanes; without the
  // residual loads / the final stores the kernel ran 2.9 / 5.4 us shorter).  So memory is touched in ROW form -- CPR = 4 RT lanes per
  // token row, 64 / CPR rows per instruction -- and a private piece of bufA (free whenever these run) turns one form into the other:
  // 16-byte chunk c of row r sits at chunk c ^ (r & (CPR - 1)), conflict-free for the ds_write_b128 and ds_read_b128 of both directions.
  constexpr int CPR = 4 * RT, RPI = 64 / CPR, ROWB = CPR * 16, TILEB = 16 * ROWB;
  constexpr int NSUB = (IMG / NW) / TILEB >= 2 ? 2 : 1;
  static_assert((IMG / NW) >= TILEB, "staging piece too small");
  char* const stg = bufA + wave * (IMG / NW);
  const int rrow = lane / CPR, rchunk = lane % CPR;            // row form: this lane's row within an instruction and its chunk
  // row of a tile that instruction j of a lane's row group touches: 4 consecutive rows per instruction, or -- in the kernels that carry the
  // temporal propagator (L = 4: rows 4 s .. 4 s + 3 are the four time steps of sequence s) -- row 4 rrow + j, so that a LANE holds the four
  // time steps of one sequence in its four registers (the propagator then needs no cross-lane traffic)
#ifdef FS_TPROP_RECOMPUTE      // measured, round 3: the T launch 46 us this way against 41 us with the rows rewritten in phase 0 -- off
  auto srow = [&](int j) { return TPROP ? 4 * rrow + j : RPI * j + rrow; };
#else
  auto srow = [&](int j) { return RPI * j + rrow; };
#endif
  auto slice_load = [&](const float* __restrict__ src, f32x4 (&raw)[NTT][RT]) {      // row form, straight from memory
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int t = tok_of(16 * tt + srow(j));
        raw[tt][j] = *(const f32x4*)(src + (long)(t < 0 ? 0 : t) * FS_C + 16 * RT * wave + 4 * rchunk);   // dead slots: token 0's row
      }
  };
  auto slice_to_acc = [&](const f32x4 (&raw)[NTT][RT], f32x4 (&acc)[RT][NTT]) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      char* sb = stg + (tt % NSUB) * TILEB;
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int r = srow(j);
        *(f32x4*)(sb + r * ROWB + ((rchunk ^ (r & (CPR - 1))) << 4)) = raw[tt][j];
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt][tt] = *(const f32x4*)(sb + l15 * ROWB + (((4 * rt + kk) ^ (l15 & (CPR - 1))) << 4));
    }
  };
  // bf16 tensors of the training forward (q | k | v, attention output, LayerNorm2's image, fc1's pre-activation and its GELU): the wave's
  // slice of tile tt sits in its own columns of an activation image (logical chunks CPB w .. CPB w + CPB - 1 of every row, swizzled by the
  // row) and goes to memory in row form: CPB lanes x 16 bytes per token row, 64 / CPB rows per instruction.  In accumulator layout these
  // were 8-byte (v: 2-byte) stores to 16 different rows per instruction -- 160 of them per wave, worked off lane by lane by the memory pipe.
  constexpr int CPB = 2 * RT, RPB = 64 / CPB;
  const int brow = lane / CPB, bchunk = lane % CPB;
  auto img_rows_store = [&](const char* img, int tt, unsigned short* __restrict__ dst, long dstride, int dcol) {
#pragma unroll
    for (int j = 0; j < 16 / RPB; ++j) {
      const int r = RPB * j + brow, t = tok_of(16 * tt + r);
      const u32x4 v = *(const u32x4*)(img + tt * 8192 + r * FS_ROW + (((CPB * wave + bchunk) ^ r) << 4));
      // (ordinary stores: written through they buy nothing -- 9.65 ms per train step either way.  The first attempt gave NaN losses: the
      // write-through store is an asm statement, and a GELU that reused its data registers right behind it hit the store-data hazard the
      // compiler's recognizer does not see inside asm -- st_wt16 carries the wait states now, common.hip.h)
#ifdef FS_EXP_NO_SAVE      // timing experiment only (wrong results): what the saved tensors' row stores cost the training forward
      if (t >= 0 && v[0] == 0x12345u) *(u32x4*)(dst + (long)t * dstride + dcol + 16 * RT * wave + 8 * bchunk) = v;
#else
      if (t >= 0) *(u32x4*)(dst + (long)t * dstride + dcol + 16 * RT * wave + 8 * bchunk) = v;
#endif
    }
  };
  auto slice_store = [&](float* __restrict__ dst, const f32x4 (&acc)[RT][NTT]) {
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      char* sb = stg + (tt % NSUB) * TILEB;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) *(f32x4*)(sb + l15 * ROWB + (((4 * rt + kk) ^ (l15 & (CPR - 1))) << 4)) = acc[rt][tt];
#pragma unroll
      for (int j = 0; j < RT; ++j) {
        const int r = RPI * j + rrow, t = tok_of(16 * tt + r);
        const f32x4 v = *(const f32x4*)(sb + r * ROWB + ((rchunk ^ (r & (CPR - 1))) << 4));
#ifdef FS_NT_STORE      // experiment: streaming (non-temporal) stores of the block's output rows
        if (t >= 0) __builtin_nontemporal_store(v, (f32x4*)(dst + (long)t * FS_C + 16 * RT * wave + 4 * rchunk));
#else                   // write-through (common.hip.h): 38.4 -> 37.2 us per launch and +2.4 % on the rollout, three interleaved rounds
        // the inference form only (the training form's launches are bound by their saved-tensor traffic, not by the release)
        if (t >= 0) {
          if constexpr (TRAIN) *(f32x4*)(dst + (long)t * FS_C + 16 * RT * wave + 4 * rchunk) = v;
          else st_wt16(dst + (long)t * FS_C + 16 * RT * wave + 4 * rchunk, v);
        }
#endif
      }
    }
  };

  // ================================ phase 0: LayerNorm1 -> bufA ===================================================================
  constexpr bool ln1_done = TPROP;
  if constexpr (TPROP) {
    {
      // ---- T letter (L = 4) with the temporal propagator fused (round 3: on the matrix pipe) ---------------------------------------
      //   y_t = x_t + b2[t] + sum_j w2[t][j] gelu(b1[j] + sum_a w1[j][a] x_a)        (attn_backbone.py:144-145, per position and channel)
      // The round-2 form kept the row layout of the plain path (a lane = 4 channels of ONE time step) and fetched the other three time
      // steps from the other lane groups: 32 ds_bpermute per 16 bytes -- +12 us per launch, what the stand-alone propagator kernel took.
      // Here a lane owns 4 CHANNELS of ALL FOUR time steps of a sequence (a wave-instruction = one whole 1 KiB token row, four rows per
      // sequence), and the two 4 x 4 contractions run as v_mfma_f32_4x4x1_16B_f32: 16 blocks of (4 x 1)(1 x 4) outer products per
      // instruction, lane 4 b + j = column j of block b, i.e. EVERY LANE IS ITS OWN COLUMN (its channel), the four accumulator registers
      // are the four hidden units / time steps, and lane 4 b + i supplies row i of the 4 x 4 weight (the same for every block).  fp32 in,
      // fp32 accumulate (an exact k-ordered fma chain), no cross-lane traffic at all; the matrix pipe is idle in this phase anyway.
      constexpr int GPW = 4 * NTT / NW;      // sequences (groups of 4 token rows) per wave
      f32x4 xv[GPW][4];
#pragma unroll
      for (int i = 0; i < GPW; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int tk = tok_of(4 * (wave * GPW + i) + t);
          xv[i][t] = *(const f32x4*)(x + (long)(tk < 0 ? 0 : tk) * FS_C + 4 * lane);      // dead slots read token 0's row
        }
      fs_wring_prime<0, RT, PF>(wq, wb);
      TpropW tw;
      tprop_weights(A.tprop, lane, tw);
#pragma unroll
      for (int i = 0; i < GPW; ++i) {
        tprop_apply(tw, xv[i]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 y = xv[i][t];
          const int tk = tok_of(4 * (wave * GPW + i) + t);      // (wave-uniform; cheaper to fetch again than to keep 16 of them)
          const bool live = tk >= 0;
#ifndef FS_TPROP_RECOMPUTE
          // The propagated row goes back to x: the residual slices are re-read from there behind the barriers below (workgroup-scope
          // release / acquire of __syncthreads; the rows of a workgroup are its own).  The alternative -- leave x alone and run the
          // residual slices through the propagator a second time (-DFS_TPROP_RECOMPUTE) -- costs the contraction + 64 GELUs per lane
          // once more on the critical path: 46 us per launch against 41 us (the extra 33 MB of stores drain under the q, k, v GEMMs).
#ifndef FS_EXP_TPROP_NO_STORE      // timing experiment only (wrong results): what the propagated rows' store and re-read cost
          if (live) *(f32x4*)(x + (long)tk * FS_C + 4 * lane) = y;
#endif
#endif
          float sm = (y[0] + y[1]) + (y[2] + y[3]);
          float sq = fmaf(y[0], y[0], fmaf(y[1], y[1], fmaf(y[2], y[2], y[3] * y[3])));
          sm = rows_sum(row16_sum(sm));
          sq = rows_sum(row16_sum(sq));
          const float mean = sm * (1.0f / FS_C);
          const float var = fmaxf(sq * (1.0f / FS_C) - mean * mean, 0.0f);
          const float rstd = live ? rsqrtf(var + A.eps) : 0.0f;
          const float sh = live ? -mean * rstd : 0.0f;
          const int slot = 4 * (wave * GPW + i) + t;
          u32x2 o2;
          o2[0] = pack_bf16x2(fmaf(y[0], rstd, sh), fmaf(y[1], rstd, sh));
          o2[1] = pack_bf16x2(fmaf(y[2], rstd, sh), fmaf(y[3], rstd, sh));
          *(u32x2*)(bufA + slot * FS_ROW + (((lane >> 1) ^ (slot & 15)) << 4) + (lane & 1) * 8) = o2;
        }
      }
    }
  }
  // A wave-instruction reads 4 token rows x 256 contiguous bytes; a row's statistics are reduced over the 16 lanes that share it.
  if constexpr (!ln1_done) {
    constexpr int GPW = 4 * NTT / NW;      // groups of 4 token rows per wave
    static_assert(GPW * NW == 4 * NTT, "token rows must split evenly over the waves");
    f32x4 v[GPW][4];
    bool lv[GPW];
    int tis[GPW];
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      const int ti = tis[i] = tok_of(4 * (wave * GPW + i) + kk);
      lv[i] = ti >= 0;
      const float* row = x + (long)(lv[i] ? ti : 0) * FS_C;      // dead slots read token 0's row (valid memory) and are zeroed below
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = *(const f32x4*)(row + 4 * (l15 + 16 * j));
    }
    fs_wring_prime<0, RT, PF>(wq, wb);     // behind the x rows in the memory queue: LayerNorm1 does not wait for them
#pragma unroll
    for (int i = 0; i < GPW; ++i) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s += v[i][j][e];
          q = fmaf(v[i][j][e], v[i][j][e], q);
        }
      s = row16_sum(s);
      q = row16_sum(q);
      const float mean = s * (1.0f / FS_C);
      const float var = fmaxf(q * (1.0f / FS_C) - mean * mean, 0.0f);
      const float rstd = lv[i] ? rsqrtf(var + A.eps) : 0.0f;
      const float sh = lv[i] ? -mean * rstd : 0.0f;
      const int slot = 4 * (wave * GPW + i) + kk, t15 = slot & 15;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x2 o2;
        o2[0] = pack_bf16x2(fmaf(v[i][j][0], rstd, sh), fmaf(v[i][j][1], rstd, sh));
        o2[1] = pack_bf16x2(fmaf(v[i][j][2], rstd, sh), fmaf(v[i][j][3], rstd, sh));
        const int chunk = (l15 >> 1) + 8 * j;
        *(u32x2*)(bufA + slot * FS_ROW + ((chunk ^ t15) << 4) + (l15 & 1) * 8) = o2;
        if constexpr (TRAIN) {
#ifdef FS_EXP_NO_SAVE
          if (lv[i] && o2[0] == 0x12345u) *(u32x2*)(A.xh1 + (long)tis[i] * FS_C + 4 * (l15 + 16 * j)) = o2;
#else
          if (lv[i]) *(u32x2*)(A.xh1 + (long)tis[i] * FS_C + 4 * (l15 + 16 * j)) = o2;
#endif
        }
      }
      if constexpr (TRAIN) {
        if (lv[i] && l15 == 0) *(float2*)(A.st1 + 2 * (long)tis[i]) = make_float2(mean, rstd);
      }
    }
  }

  // ---- per-lane LDS addressing ----------------------------------------------------------------------------------------------
  // fragment (tile tt, k-step ks) of an image: lane -> row 16 tt + l15, chunk 4 ks + kk, swizzled by the row:
  //   byte = tt * 8192 + l15 * 512 + (ks >> 2) * 256 + rdo[ks & 3],   rdo[j] = ((((j ^ (l15 >> 2)) << 2) | (kk ^ (l15 & 3))) << 4)
  int rdo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rdo[j] = l15 * FS_ROW + ((((j ^ (l15 >> 2)) << 2) | (kk ^ (l15 & 3))) << 4);
  // accumulator tile rt of this wave (features 16 RT w + 16 rt + 4 kk .. +3 of token column l15) -> 8 bytes of row 16 tt + l15
  int wro[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) wro[rt] = l15 * FS_ROW + (((2 * RT * wave + 2 * rt + (kk >> 1)) ^ l15) << 4) + (kk & 1) * 8;

  const float* const bias = lbias + 16 * RT * wave + 4 * kk;   // + 256 m + 16 rt: matrix m, row tile rt (in LDS)

  FS_STAMP(2);
  __syncthreads();
  FS_STAMP(3);

  // ================================ phase 1: this wave's heads: q, k, v projections + attention -> bufB ==========================
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NP = (NTT + 1) / 2;
  u32x4 qf[HPW][NTT], kf[HPW][NTT];
  u32x4 vtf[HPW][2][NP];   // V^T operand fragments: [head][d tile][pair of key tiles]
  {
    f32x4 aq[RT][NTT];   // D[feature 16 j + 4 kk + r][token]
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const f32x4 bq = *(const f32x4*)(bias + 16 * j);
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) aq[j][tt] = bq;
    }
    fs_slice_gemm<0, 24, NTT, RT, false, PF>(wq, wb, bufA, rdo, aq);
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) qf[hh][tt] = pack8(aq[2 * hh][tt], aq[2 * hh + 1][tt]);
    if (TRAIN && A.qkv) {   // the packed projection the three-launch attention backward reads holds q WITHOUT the softmax scale folded into Wq
      const float unscale = 1.0f / (0.17677669529663687f * 1.44269504088896340736f);
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {      // through this wave's columns of bufB (its attention output goes there later)
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const f32x4 qv = aq[j][tt] * unscale;
          u32x2 u;
          u[0] = pack_bf16x2(qv[0], qv[1]); u[1] = pack_bf16x2(qv[2], qv[3]);
          *(u32x2*)(bufB + tt * 8192 + wro[j]) = u;
        }
        img_rows_store(bufB, tt, A.qkv, 3 * FS_C, 0);
      }
    }
  }
  FS_STAMP(4);
  {
    f32x4 ak[RT][NTT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      // inference: no key bias, it cancels in the softmax; training keeps it so that the saved k is the unfused path's
      const f32x4 bk = TRAIN ? *(const f32x4*)(bias + 1024 + 16 * j) : zero4;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) ak[j][tt] = bk;
    }
    fs_slice_gemm<1, 24, NTT, RT, false, PF>(wq, wb, bufA, rdo, ak);
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) kf[hh][tt] = pack8(ak[2 * hh][tt], ak[2 * hh + 1][tt]);
    if (TRAIN && A.qkv) {
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          u32x2 u;
          u[0] = pack_bf16x2(ak[j][tt][0], ak[j][tt][1]); u[1] = pack_bf16x2(ak[j][tt][2], ak[j][tt][3]);
          *(u32x2*)(bufB + tt * 8192 + wro[j]) = u;
        }
        img_rows_store(bufB, tt, A.qkv, 3 * FS_C, FS_C);
      }
    }
  }
  FS_STAMP(5);
  {
    f32x4 av[RT][NTT];   // roles swapped:  D[token][d 16 j + l15]
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      // inference: the value bias is folded into the out-proj bias (rows of P sum to 1); with attention dropout they do not, so the
      // training form adds it here (per feature = per lane in this layout) and uses the plain out-proj bias
      const float bv = TRAIN ? lbias[1280 + 16 * RT * wave + 16 * j + l15] : 0.0f;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) av[j][tt] = f32x4{bv, bv, bv, bv};
    }
    fs_slice_gemm<2, 24, NTT, RT, true, PF>(wq, wb, bufA, rdo, av);
    if (TRAIN && A.qkv) {   // rows of this layout are tokens 4 kk + r, the lane is one feature: 2-byte LDS stores into the image rows
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * kk + r;
#pragma unroll
          for (int j = 0; j < RT; ++j) {
            const __bf16 b = (__bf16)av[j][tt][r];
            *(unsigned short*)(bufB + tt * 8192 + row * FS_ROW + (((CPB * wave + 2 * j + (l15 >> 3)) ^ row) << 4) + (l15 & 7) * 2) =
                __builtin_bit_cast(unsigned short, b);
          }
        }
        img_rows_store(bufB, tt, A.qkv, 3 * FS_C, 2 * FS_C);
      }
    }
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          vtf[hh][dt][p] = (2 * p + 1 < NTT) ? pack8(av[2 * hh + dt][2 * p], av[2 * hh + dt][2 * p + 1]) : pack8(av[2 * hh + dt][2 * p], zero4);
  }
  FS_STAMP(6);
  if constexpr (G == 2 && FS_SKEW) __builtin_amdgcn_s_barrier();      // alignment only: MATRIX segment (q, k, v) | vector segment (attention)

  // ---- attention: per head and query tile, S^T = K Q^T over its key tiles, softmax down the accumulator rows, O^T = V^T P^T ----
  {
    // TPS == 1: the block-diagonal (and causal) pattern inside a tile is the same for every tile: bit r <-> key row 4 kk + r
    unsigned allow1 = 0xfu;
    bool nomask = true;
    if constexpr (TPS == 1) {
      allow1 = 0u;
      const int si = (int)(((unsigned)l15 * A.magic) >> 16), pi = l15 - si * L;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * kk + r, sj = (int)(((unsigned)j * A.magic) >> 16), pj = j - sj * L;
        allow1 |= ((sj == si && (!A.causal || pj <= pi)) ? 1u : 0u) << r;
      }
      nomask = (L == 16) && !A.causal;
    }
    // Staged over ALL query tiles of a head (scores, maxima, exponentials, sums, P V, scale + store): the steps of one tile form a
    // long dependent chain (MFMA -> max -> cross-lane -> exp -> sum -> cross-lane -> rcp -> pack -> MFMA -> pack -> store); written
    // tile by tile the wave sits out every latency of it, written stage by stage the NTT chains interleave.
    constexpr int NK = TPS > 0 ? TPS : NTT;
    constexpr int QG = fs_group(NTT, NK);        // query tiles staged together: QG x NK score tiles (<= 8, 32 registers) in flight
    static_for<HPW * (NTT / QG)>([&](auto hg_c) {
      constexpr int hh = decltype(hg_c)::value / (NTT / QG), q0 = (decltype(hg_c)::value % (NTT / QG)) * QG;
      f32x4 sc[QG][NK];
      unsigned allow[QG];      // bit 4 j + r: key row 4 kk + r of visible tile j may be seen by this lane's query of tile q0 + q
      float mx[QG], sum[QG];
      static_for<QG>([&](auto qt_c) {
        constexpr int q = decltype(qt_c)::value, qt = q0 + q;
        constexpr int k0 = TPS > 0 ? (qt / TPS) * TPS : 0;            // first key tile this query tile can see
#pragma unroll
        for (int j = 0; j < NK; ++j) sc[q][j] = mfma_bf16(kf[hh][k0 + j], qf[hh][qt], zero4);   // rows = keys, column = query
      });
#pragma unroll
      for (int qt = 0; qt < QG; ++qt) {
        if constexpr (TPS == 0) {
          const int qs = 16 * (q0 + qt) + l15, si = (int)(((unsigned)qs * A.magic) >> 16), pi = qs - si * L;
          allow[qt] = 0u;
#pragma unroll
          for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int ksl = 16 * j + 4 * kk + r, sj = (int)(((unsigned)ksl * A.magic) >> 16), pj = ksl - sj * L;
              allow[qt] |= ((sj == si && ksl < nslot && (!A.causal || pj <= pi)) ? 1u : 0u) << (4 * j + r);
            }
        } else {
          allow[qt] = allow1 * 0x11111111u;
        }
      }
      if (TPS != 0 && nomask) {
#pragma unroll
        for (int qt = 0; qt < QG; ++qt) {
          float m = -INFINITY;
#pragma unroll
          for (int j = 0; j < NK; ++j) m = fmaxf(m, fmaxf(fmaxf(sc[qt][j][0], sc[qt][j][1]), fmaxf(sc[qt][j][2], sc[qt][j][3])));
          mx[qt] = m;
        }
      } else {
#pragma unroll
        for (int qt = 0; qt < QG; ++qt) {
          float m = -INFINITY;
#pragma unroll
          for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if ((allow[qt] >> (4 * j + r)) & 1u) m = fmaxf(m, sc[qt][j][r]);
          mx[qt] = m;
        }
      }
#pragma unroll
      for (int qt = 0; qt < QG; ++qt) mx[qt] = rows_max(mx[qt]);
      if (TPS != 0 && nomask) {
#pragma unroll
        for (int qt = 0; qt < QG; ++qt) {
          float sm = 0.0f;
#pragma unroll
          for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              sc[qt][j][r] = __builtin_amdgcn_exp2f(sc[qt][j][r] - mx[qt]);
              sm += sc[qt][j][r];
            }
          sum[qt] = sm;
        }
      } else {
#pragma unroll
        for (int qt = 0; qt < QG; ++qt) {
          float sm = 0.0f;
#pragma unroll
          for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              sc[qt][j][r] = ((allow[qt] >> (4 * j + r)) & 1u) ? __builtin_amdgcn_exp2f(sc[qt][j][r] - mx[qt]) : 0.0f;
              sm += sc[qt][j][r];
            }
          sum[qt] = sm;
        }
      }
#pragma unroll
      for (int qt = 0; qt < QG; ++qt) sum[qt] = rows_sum(sum[qt]);
      if constexpr (TRAIN) {
        if (A.p_drop > 0.0f) {   // dropout on the probabilities, the normaliser stays the undropped sum (tante_attention_dropout's order)
          const float ksc = 1.0f / (1.0f - A.p_drop);
          static_for<QG>([&](auto qt_c) {
            constexpr int qt = decltype(qt_c)::value;
            constexpr int k0 = TPS > 0 ? ((q0 + qt) / TPS) * TPS : 0;
            const int qs = 16 * (q0 + qt) + l15, si = (int)(((unsigned)qs * A.magic) >> 16), pi = qs - si * L;
            const unsigned long long mrow = ((((unsigned long long)(seq0 + si)) * 8 + (HPW * wave + hh)) * L + pi) * L;
#pragma unroll
            for (int j = 0; j < NK; ++j) {
              const int jpos0 = 16 * (k0 + j) + 4 * kk - si * L;          // key position inside the query's sequence (where visible)
              if ((L & 3) == 0) {      // the lane's four keys are one aligned run of mask indices: two hashes instead of four
                const unsigned m4 = dropout_keep4(sd_attn, mrow + (unsigned long long)(jpos0 < 0 ? 0 : jpos0), A.p_drop);
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[qt][j][r] = ((m4 >> r) & 1u) ? sc[qt][j][r] * ksc : 0.0f;
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int jpos = jpos0 + r;
                  sc[qt][j][r] = dropout_keep(sd_attn, mrow + (unsigned long long)(jpos < 0 ? 0 : jpos), A.p_drop) ? sc[qt][j][r] * ksc : 0.0f;
                }
              }
            }
          });
        }
      }
      // O^T[d][query] = sum over key-tile pairs; the pairs are the GLOBAL pairs (2 g, 2 g + 1) the V^T fragments were packed in
      f32x4 o[QG][2];
      static_for<QG>([&](auto qt_c) {
        constexpr int qt = decltype(qt_c)::value;
        constexpr int k0 = TPS > 0 ? ((q0 + qt) / TPS) * TPS : 0;
        constexpr int g0 = k0 / 2, g1 = (k0 + NK - 1) / 2;
        o[qt][0] = o[qt][1] = zero4;
        static_for<g1 - g0 + 1>([&](auto g_c) {
          constexpr int g = g0 + decltype(g_c)::value;
          constexpr int ja = 2 * g - k0, jb = 2 * g + 1 - k0;     // indices into sc[qt][], possibly outside the visible range
          f32x4 pa = zero4, pb = zero4;
          if constexpr (ja >= 0 && ja < NK) pa = sc[qt][ja];
          if constexpr (jb >= 0 && jb < NK) pb = sc[qt][jb];
          const u32x4 pf = pack8(pa, pb);
          o[qt][0] = mfma_bf16(vtf[hh][0][g], pf, o[qt][0]);
          o[qt][1] = mfma_bf16(vtf[hh][1][g], pf, o[qt][1]);
        });
      });
      // this head's 32 output features of the 16 tokens of each tile -> attention-output image
#pragma unroll
      for (int qt = 0; qt < QG; ++qt) {
        const float inv = sum[qt] > 0.0f ? __builtin_amdgcn_rcpf(sum[qt]) : 0.0f;
        const f32x4 o0 = o[qt][0] * inv, o1 = o[qt][1] * inv;
        u32x2 w0, w1;
        w0[0] = pack_bf16x2(o0[0], o0[1]); w0[1] = pack_bf16x2(o0[2], o0[3]);
        w1[0] = pack_bf16x2(o1[0], o1[1]); w1[1] = pack_bf16x2(o1[2], o1[3]);
        *(u32x2*)(bufB + (q0 + qt) * 8192 + wro[2 * hh]) = w0;
        *(u32x2*)(bufB + (q0 + qt) * 8192 + wro[2 * hh + 1]) = w1;
      }
    });
  }
  if constexpr (TRAIN) {     // the attention output (bf16) as the backward reads it: this wave's columns of the image, row form
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) img_rows_store(bufB, tt, A.o, FS_C, 0);
  }

  FS_STAMP(7);
  asm volatile("" ::: "memory");   // keep the residual loads below the attention (they would double its register pressure)
  // residual slice of this wave: x[token][16 RT w .. + 16 RT), row form; the loads fly while the out-proj GEMM runs
  f32x4 xraw[NTT][RT];
#ifndef FS_RESID_FIRST
  fs_wring_prime<3, RT, PF>(wq, wb);      // ahead of the residual rows in the wave's (in-order) memory queue: the out-proj GEMM starts on them
#endif
#ifdef FS_EXP_NO_RESID     // timing experiment only (wrong results)
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
    for (int j = 0; j < RT; ++j) xraw[tt][j] = f32x4{(float)tt, 1.f, 2.f, (float)j};
#else
  slice_load(x, xraw);
#endif
#ifdef FS_RESID_FIRST
  fs_wring_prime<3, RT, PF>(wq, wb);      // (round-2 order, kept for the A/B)
#endif
  FS_STAMP(8);
  __syncthreads();
  FS_STAMP(9);

  // ================================ phase 2: out-proj slice + residual; LayerNorm2 statistics ===================================
  f32x4 x1[RT][NTT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const f32x4 bo = *(const f32x4*)(bias + (TRAIN ? 1536 : 256) + 16 * rt);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) x1[rt][tt] = bo;
  }
  fs_slice_gemm<3, 48, NTT, RT, false, PF>(wq, wb, bufB, rdo, x1);
  // x + dropout(branch): index = row * 256 + column, the GEMM epilogue's / tante_dropout_bwd's (four consecutive columns per lane)
  auto drop_add = [&](f32x4 (&acc)[RT][NTT], f32x4 (&res)[RT][NTT], unsigned long long seed) {
    if constexpr (TRAIN) {
      if (A.p_drop > 0.0f) {
        const float ksc = 1.0f / (1.0f - A.p_drop);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) {
          const unsigned long long i0 = (unsigned long long)(tokidx[tt] < 0 ? 0 : tokidx[tt]) * FS_C + 16 * RT * wave + 4 * kk;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const unsigned k01 = dropout_keep2(seed, i0 + 16 * rt, A.p_drop), k23 = dropout_keep2(seed, i0 + 16 * rt + 2, A.p_drop);
            acc[rt][tt][0] = (k01 & 1u) ? acc[rt][tt][0] * ksc : 0.0f;
            acc[rt][tt][1] = (k01 & 2u) ? acc[rt][tt][1] * ksc : 0.0f;
            acc[rt][tt][2] = (k23 & 1u) ? acc[rt][tt][2] * ksc : 0.0f;
            acc[rt][tt][3] = (k23 & 2u) ? acc[rt][tt][3] * ksc : 0.0f;
          }
        }
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) acc[rt][tt] += res[rt][tt];
  };
#ifdef FS_TPROP_RECOMPUTE
  if constexpr (TPROP) {      // the residual is the PROPAGATED row: the same contraction on this wave's feature slice (see srow) -- here,
    static_assert(!TPROP || RT == 4, "a lane's row group = the four time steps of a sequence");      // behind the out-proj GEMM the loads flew under
    TpropW tw;
    tprop_weights(A.tprop, lane, tw);
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) tprop_apply(tw, xraw[tt]);
  }
#endif
  {
    f32x4 xr[RT][NTT];
    slice_to_acc(xraw, xr);      // bufA: LayerNorm1's image died with the q/k/v GEMMs (barrier 2), LayerNorm2's comes after barrier 3
    drop_add(x1, xr, A.seed_out ^ smix);
  }
  if constexpr (TRAIN) {   // the residual after the attention half: the per-operator LayerNorm2 backward reads it (null when the fused tail backward runs)
    if (A.x1) slice_store(A.x1, x1);
  }
  FS_STAMP(10);
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s += x1[rt][tt][r];
        q = fmaf(x1[rt][tt][r], x1[rt][tt][r], q);
      }
    s = rows_sum(s);
    q = rows_sum(q);
    if (kk == 0) *(float2*)(stat + ((16 * tt + l15) * NW + wave) * 8) = make_float2(s, q);
  }
  __syncthreads();
  FS_STAMP(11);
  // every wave normalises its own feature slice of every token with the full-row statistics, -> bufA (free since phase 1 ended)
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const f32x4* sp = (const f32x4*)(stat + (16 * tt + l15) * NW * 8);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < NW / 2; ++j) {
      const f32x4 a = sp[j];
      s += a[0] + a[2];
      q += a[1] + a[3];
    }
    const float mean = s * (1.0f / FS_C);
    const float var = fmaxf(q * (1.0f / FS_C) - mean * mean, 0.0f);
    const float rstd = rsqrtf(var + A.eps);
    const float sh = -mean * rstd;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      u32x2 o2;
#ifdef FS_EXP_LN_ID        // timing experiment only (wrong results): the price of LayerNorm2's normalise pass
      o2[0] = pack_bf16x2(x1[rt][tt][0], x1[rt][tt][1]);
      o2[1] = pack_bf16x2(x1[rt][tt][2], x1[rt][tt][3]);
#else
      o2[0] = pack_bf16x2(fmaf(x1[rt][tt][0], rstd, sh), fmaf(x1[rt][tt][1], rstd, sh));
      o2[1] = pack_bf16x2(fmaf(x1[rt][tt][2], rstd, sh), fmaf(x1[rt][tt][3], rstd, sh));
#endif
      *(u32x2*)(bufA + tt * 8192 + wro[rt]) = o2;
    }
    if constexpr (TRAIN) {
      img_rows_store(bufA, tt, A.xh2, FS_C, 0);
      if (wave == 0 && kk == 0 && tokidx[tt] >= 0) *(float2*)(A.st2 + 2 * (long)tokidx[tt]) = make_float2(mean, rstd);
    }
  }
  __syncthreads();
  FS_STAMP(12);

  // ================================ phase 3: fc1 slice + GELU -> bufB ============================================================
  {
    f32x4 h[RT][NTT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const f32x4 b1 = *(const f32x4*)(bias + 512 + 16 * rt);
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) h[rt][tt] = b1;
    }
    fs_slice_gemm<4, 48, NTT, RT, false, PF>(wq, wb, bufA, rdo, h);
    FS_STAMP(13);
    if constexpr (G == 2 && FS_SKEW) __builtin_amdgcn_s_barrier();    // alignment only: MATRIX segment (fc1) | vector segment (GELU)
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) {
      if constexpr (TRAIN) {     // the pre-activation passes through the image columns the GELU-ed values take next
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          u32x2 hp;
          hp[0] = pack_bf16x2(h[rt][tt][0], h[rt][tt][1]);
          hp[1] = pack_bf16x2(h[rt][tt][2], h[rt][tt][3]);
          *(u32x2*)(bufB + tt * 8192 + wro[rt]) = hp;
        }
        img_rows_store(bufB, tt, A.hpre, FS_C, 0);
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#ifdef FS_EXP_GELU_ID      // timing experiment only (wrong results): the whole price of the GELU arithmetic (round-5 verdict item 4)
        const f32x4 g = h[rt][tt];
#else
        const f32x4 g = gelu_poly4<true>(h[rt][tt]);
#endif
        u32x2 o2;
        o2[0] = pack_bf16x2(g[0], g[1]);
        o2[1] = pack_bf16x2(g[2], g[3]);
        *(u32x2*)(bufB + tt * 8192 + wro[rt]) = o2;
      }
      if constexpr (TRAIN) img_rows_store(bufB, tt, A.act, FS_C, 0);
    }
  }
  __syncthreads();
  FS_STAMP(14);

  // ================================ phase 4: fc2 slice + residual -> x ============================================================
  {
    f32x4 y2[TRAIN ? RT : 1][TRAIN ? NTT : 1];
    if constexpr (TRAIN) {     // the branch is dropped before the residual is added: it needs accumulators of its own
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 b2 = *(const f32x4*)(bias + 768 + 16 * rt);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) y2[rt][tt] = b2;
      }
      fs_slice_gemm<5, 48, NTT, RT, false, PF>(wq, wb, bufB, rdo, y2);
      drop_add(y2, x1, A.seed_mlp ^ smix);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) x1[rt][tt] = y2[rt][tt];
    } else {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 b2 = *(const f32x4*)(bias + 768 + 16 * rt);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt) x1[rt][tt] += b2;
      }
      fs_slice_gemm<5, 48, NTT, RT, false, PF>(wq, wb, bufB, rdo, x1);
    }
    FS_STAMP(15);
#ifdef TANTE_ABLATE
    if (A.stamps && lane == 0) A.stamps[((long)vblock * 8 + wave) * 20 + 19] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef FS_EXP_NO_STORE     // timing experiment only (wrong results): one store per tile keeps the accumulators alive
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt)
      if (tokidx[tt] >= 0) {
        float* row = (TRAIN ? A.out : x) + (long)tokidx[tt] * FS_C + 16 * RT * wave + 4 * kk;
        f32x4 sacc = x1[0][tt];
#pragma unroll
        for (int rt = 1; rt < RT; ++rt) sacc += x1[rt][tt];
        if (sacc[0] == 1.2345f) *(f32x4*)row = sacc;
      }
#else
    slice_store(TRAIN ? A.out : x, x1);      // bufA: LayerNorm2's image died with the fc1 GEMM (barrier 5)
#endif
  }
  if constexpr (G == 2 && FS_SKEW) {
    if (grp == 0) __builtin_amdgcn_s_barrier();      // pairs with group 1's entry barrier: both groups execute the same number
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Packing.  Fragment = 64 lanes x 8 bf16; lane (l15, kk) holds  W[row0 + l15][32 ks + 8 kk + 0..7]  (natural k order: every
// activation operand comes from an LDS image in natural feature order, or from an accumulator pair whose k index is the head
// dimension on BOTH sides of the product).
//   f = (m * 8 + ks) * 16 + g :  matrix m = q | k | v rows of W_in (q scaled by log2(e) / sqrt(32)) | Wo | W1 | W2, k-step ks,
//   rows 16 g .. 16 g + 15 of it.  A wave that owns RT consecutive row tiles finds the RT fragments of a k-step contiguous, whatever
//   RT is: the same stream serves the 8-wave and the 4-wave kernels.
// Biases (fp32, 7 x 256 by output feature): q (scaled) | out' = b_out + W_out (b_v + W_v beta1) | fc1 | fc2 | k | v | b_out.
// LayerNorm gammas are folded into the columns of W_in / W1, the betas into the biases.
// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fs_pack_body(int bid, const float* __restrict__ w_in, const float* __restrict__ b_in, const float* __restrict__ g1,
                                             const float* __restrict__ be1, const float* __restrict__ w_out, const float* __restrict__ b_out,
                                             const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ g2,
                                             const float* __restrict__ be2, const float* __restrict__ w2, const float* __restrict__ b2,
                                             char* __restrict__ dst) {
  constexpr int C = FS_C;
  const float qscale = 0.17677669529663687f * 1.44269504088896340736f;  // log2(e) / sqrt(32): the kernel's softmax is exp2
  const int nfrag = FS_W_BYTES / FS_FRAG;
  if (bid < nfrag / 4) {
    const int f = bid * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, l15 = lane & 15, kk = lane >> 4;
    const int g = f % 16, ks = (f / 16) % 8, m = f / 128;
    const int row = (m < 3 ? m * C : 0) + 16 * g + l15;
    const float* src = m < 3 ? w_in : (m == 3 ? w_out : (m == 4 ? w1 : w2));
    const float* gamma = m < 3 ? g1 : (m == 4 ? g2 : nullptr);
    const float scale = m == 0 ? qscale : 1.0f;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 32 * ks + 8 * kk + e;
      float xv = src[(long)row * C + k];
      if (gamma) xv *= gamma[k];
      v[e] = xv * scale;
    }
    u32x4 o;
    o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
    *(u32x4*)(dst + (long)f * FS_FRAG + lane * 16) = o;
    return;
  }
  // last block: the biases.  bv' = b_v + W_v beta1 (256 values), then bo' = b_out + W_out bv'
  __shared__ float bv[FS_C];
  const int n = threadIdx.x;   // 256 threads
  if (!be1) {   // training pack: W_in / W1 and their biases arrive with the LayerNorm affines already folded in (FoldFn); plain copies
    float* bias = (float*)(dst + FS_W_BYTES);
    bias[n] = b_in[n] * qscale;
    bias[256 + n] = b_out[n];          // (the inference-only folded out-proj bias is not used by the training kernel)
    bias[512 + n] = b1[n];
    bias[768 + n] = b2[n];
    bias[1024 + n] = b_in[C + n];
    bias[1280 + n] = b_in[2 * C + n];
    bias[1536 + n] = b_out[n];
    return;
  }
  {
    float s = b_in[2 * C + n];
    for (int k = 0; k < C; ++k) s += w_in[(long)(2 * C + n) * C + k] * be1[k];
    bv[n] = s;
  }
  __syncthreads();
  float* bias = (float*)(dst + FS_W_BYTES);
  {
    float s = b_in[n];
    for (int k = 0; k < C; ++k) s += w_in[(long)n * C + k] * be1[k];
    bias[n] = s * qscale;
  }
  {
    float s = b_out[n];
    for (int c = 0; c < C; ++c) s += w_out[(long)n * C + c] * bv[c];
    bias[256 + n] = s;
  }
  {
    float s = b1[n];
    for (int k = 0; k < C; ++k) s += w1[(long)n * C + k] * be2[k];
    bias[512 + n] = s;
  }
  bias[768 + n] = b2[n];
  {   // training form: key and value biases (LayerNorm1's beta folded in), and the out-proj bias WITHOUT the value bias folded in
    float s = b_in[C + n];
    for (int k = 0; k < C; ++k) s += w_in[(long)(C + n) * C + k] * be1[k];
    bias[1024 + n] = s;
  }
  bias[1280 + n] = bv[n];
  bias[1536 + n] = b_out[n];
}
__global__ void fs_pack_kernel(const float* __restrict__ w_in, const float* __restrict__ b_in, const float* __restrict__ g1,
                               const float* __restrict__ be1, const float* __restrict__ w_out, const float* __restrict__ b_out,
                               const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ g2,
                               const float* __restrict__ be2, const float* __restrict__ w2, const float* __restrict__ b2,
                               char* __restrict__ dst) {
  fs_pack_body((int)blockIdx.x, w_in, b_in, g1, be1, w_out, b_out, w1, b1, g2, be2, w2, b2, dst);
}
// the training streams of several blocks in one launch (folded weights: gamma = beta = null)
constexpr int FSP_BLOCKS = FS_W_BYTES / FS_FRAG / 4 + 1;
struct FsPackBatch {
  const float* p[TANTE_FSP_MAX][8];
  char* dst[TANTE_FSP_MAX];
};
__global__ void fs_pack_multi_kernel(FsPackBatch B) {
  const int e = blockIdx.x / FSP_BLOCKS;
  const float* const* q = B.p[e];
  fs_pack_body((int)blockIdx.x - e * FSP_BLOCKS, q[0], q[1], nullptr, nullptr, q[2], q[3], q[4], q[5], nullptr, nullptr, q[6], q[7], B.dst[e]);
}

template <int TPS, int NTT, int NW, bool TRAIN, int G = 1, bool TPROP = false>
void fs_launch_tt(const FsArgs& A, int nwg, hipStream_t s) {
  constexpr int LDS = G * (2 * 16 * NTT * FS_ROW + 16 * NTT * NW * 8 + FS_BIAS_FLOATS * 4);
  static_assert(LDS <= 160 * 1024, "LDS per workgroup");
  static TantePerDevice attr;
  attr.once([&] { (void)hipFuncSetAttribute((const void*)block_fs_kernel<TPS, NTT, NW, TRAIN, G, TPROP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
  hipLaunchKernelGGL((block_fs_kernel<TPS, NTT, NW, TRAIN, G, TPROP>), dim3((nwg + G - 1) / G), dim3(64 * NW * G), LDS, s, A);
}
// TANTE_FS_GROUPS = 2 (tante_set_option) selects the paired form; the default is the unpaired one.  Both compute the same function, bit for
// bit (tools/fs_ab.py asserts it).  MEASURED (round 3, one box, interleaved rounds, cfg2 B = 8): the paired form is 1.2 - 2.3 us SLOWER per
// launch (T 36.3 -> 38.1, H 35.3 -> 37.5, W 34.3 -> 36.5 us), also with the weight stream served from L1 and without the x loads / stores
// (timing builds: 26.0 -> 26.8 us), and with the two groups in step instead of offset (-DFS_SKEW=0: 34.0 -> 35.3 us).  Anti-phase by
// construction buys nothing: what the launch waits for is not the matrix pipe being idle in vector phases.
// the training form exists for the 4-wave kernels (sequences up to 64 tokens: every shipped axis letter); longer ones train unfused
template <int TPS, int NTT, int NW>
void fs_launch_t(const FsArgs& A, int nwg, hipStream_t s) {
  if constexpr (NW == 4) {
    if (A.out) return fs_launch_tt<TPS, NTT, NW, true>(A, nwg, s);
    if constexpr (TPS == 1) {
      if (A.tprop) return fs_launch_tt<TPS, NTT, NW, false, 1, true>(A, nwg, s);
    }
    const int groups = tante_opt("TANTE_FS_GROUPS", 0);
    if (groups == 2) return fs_launch_tt<TPS, NTT, NW, false, 2>(A, nwg, s);
  }
  fs_launch_tt<TPS, NTT, NW, false>(A, nwg, s);
}

// waves per workgroup: 4 = two independent 64-token workgroups per CU (sequences up to 64 tokens), 8 = one 128-token workgroup per
// CU (any L <= 128).  tante_set_option("TANTE_FS_WAVES", 8) forces the 8-wave form everywhere (A/B timing; both compute the same function).
int fs_waves(int L) {
  const int force8 = tante_opt("TANTE_FS_WAVES", 0) == 8;
  return (force8 || L > 64) ? 8 : 4;
}

}  // namespace

#ifdef TANTE_ABLATE
extern "C" void tante_fs_set_stamps(void* p) { g_fs_stamps = (unsigned long long*)p; }   // (blocks * 8 waves * 16) u64, diagnostic builds only
#endif

int tante_fs_supported(int C, int n_head, int hidden, int L, int causal) {
  (void)causal;
  return C == FS_C && n_head == 8 && hidden == FS_C && L >= 1 && L <= 128;   // (launch also needs nseq < 2^23)
}

int64_t tante_fs_stream_bytes(int C, int hidden) { return (C == FS_C && hidden == FS_C) ? FS_W_BYTES + FS_BIAS_FLOATS * 4 : 0; }

void tante_fs_pack(const float* ln1_w, const float* ln1_b, const float* in_w, const float* in_b, const float* out_w, const float* out_b,
                   const float* ln2_w, const float* ln2_b, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                   const float* fc2_b, char* dst, hipStream_t s) {
  hipLaunchKernelGGL(fs_pack_kernel, dim3(FS_W_BYTES / FS_FRAG / 4 + 1), dim3(256), 0, s, in_w, in_b, ln1_w, ln1_b, out_w, out_b, fc1_w,
                     fc1_b, ln2_w, ln2_b, fc2_w, fc2_b, dst);
}

// training pack: only this kernel's stream, from weights whose LayerNorm affines are already folded in (gamma = beta = null)
void tante_fs_pack_folded(const float* in_w, const float* in_b, const float* out_w, const float* out_b, const float* fc1_w,
                          const float* fc1_b, const float* fc2_w, const float* fc2_b, char* dst, hipStream_t s) {
  hipLaunchKernelGGL(fs_pack_kernel, dim3(FS_W_BYTES / FS_FRAG / 4 + 1), dim3(256), 0, s, in_w, in_b, (const float*)nullptr,
                     (const float*)nullptr, out_w, out_b, fc1_w, fc1_b, (const float*)nullptr, (const float*)nullptr, fc2_w, fc2_b, dst);
}

// entries: 8 pointers each (in_w, in_b, out_w, out_b, fc1_w, fc1_b, fc2_w, fc2_b, LayerNorm affines folded in), n <= TANTE_FSP_MAX
void tante_fs_pack_folded_multi(const float* const (*params)[8], char* const* dst, int n, hipStream_t s) {
  FsPackBatch B;
  for (int e = 0; e < n; ++e) {
    for (int k = 0; k < 8; ++k) B.p[e][k] = params[e][k];
    B.dst[e] = dst[e];
  }
  hipLaunchKernelGGL(fs_pack_multi_kernel, dim3((unsigned)(n * FSP_BLOCKS)), dim3(256), 0, s, B);
}

int tante_fs_launch(float* x, const char* stream, const TanteSeq& sq, int causal, float eps, hipStream_t s, const TanteBlockTrain* tr,
                    const float* tprop) {
  if (sq.nseq >= (1 << 23)) return -2;
  if (tprop && (tr || sq.L != 4)) return -5;
  FsArgs A;
  A.x = x; A.w = stream; A.sq = sq; A.causal = causal; A.eps = eps;
  A.out = nullptr;
  A.seed_mix = nullptr;
  A.tprop = tprop;
  if (tr) {
    if (sq.L > 64) return -4;
    A.out = tr->out; A.xh1 = (unsigned short*)tr->xh1; A.st1 = tr->st1; A.qkv = (unsigned short*)tr->qkv; A.o = (unsigned short*)tr->o;
    A.x1 = tr->x1; A.xh2 = (unsigned short*)tr->xh2; A.st2 = tr->st2; A.hpre = (unsigned short*)tr->hpre; A.act = (unsigned short*)tr->act;
    A.p_drop = tr->p_drop; A.seed_attn = tr->seed_attn; A.seed_out = tr->seed_out; A.seed_mlp = tr->seed_mlp;
    A.seed_mix = tante_seed_mix_ptr();
  }
  A.stamps = nullptr;
#ifdef TANTE_ABLATE
  A.stamps = g_fs_stamps;
#endif
  const int L = sq.L, nw = (tr || tprop) ? 4 : fs_waves(L);      // (the training form and the fused propagator exist in the 4-wave kernels)
  A.magic = (65536u + (unsigned)L - 1u) / (unsigned)L;
  // tile-aligned shapes: L | 16 (several sequences per tile), L = 32 / 48 / 64 (2 / 3 / 4 tiles per sequence, non-causal)
  int tps = 0;
  if (16 % L == 0) tps = 1;
  else if (!causal && (L == 32 || L == 48 || L == 64)) tps = L / 16;
  // token tiles per workgroup: the full size (8 tiles at 8 waves, 4 at 4 waves) or three quarters of it; take the one that needs
  // fewer rounds of resident workgroups (256 CUs x 1 or 2), then fewer tiles.  L = 48 only fits the three-quarter form.
  const int full = nw == 8 ? 8 : 4, tq = nw == 8 ? 6 : 3;
  const long resident = 256L * (8 / nw);
  auto rounds = [&](int ntt) {
    const int spw = 16 * ntt / L;
    const long nwg = (sq.nseq + spw - 1) / spw;
    return ((nwg + resident - 1) / resident) * ntt;
  };
  int ntt = full;
  if (tps == 3) ntt = tq;
  else if (tps >= 1 && (16 * tq) % L == 0 && (16 * tq) / L >= 1 && (tps == 1 || tq % tps == 0) && rounds(tq) < rounds(full)) ntt = tq;
  // small batches (inference, 4 waves, L | 16 or L = 32): HALF-size workgroups (2 tiles = 32 tokens) while the full-size grid fills at most
  // a quarter of the resident slots -- cfg2 at B = 1 is 64 full-size workgroups for 256 CUs, each the whole serial chain of a block; 128
  // half-size ones run it on half the tokens.  Measured as captured graphs (tools/_r4_half.sh, same box): B = 1 3 920 -> 4 432 frames/s,
  // B = 2 7 232 -> 7 898; B = 4 (256 full-size workgroups) 12 480 -> 12 150 and B = 6 -0.6 %: not there.  Bit-identical either way.
  if (!tr && nw == 4 && (tps == 1 || tps == 2) && 32 % L == 0 && tante_opt("TANTE_FS_HALF", 1)) {
    const int spw_full = 16 * ntt / L;
    if (((long)sq.nseq + spw_full - 1) / spw_full * 4 <= resident) ntt = 2;
  }
  A.spw = tps ? 16 * ntt / L : (16 * full) / L;
  const int nwg = (sq.nseq + A.spw - 1) / A.spw;
  // one resident round of two workgroups per CU (cfg2 / cfg3 at B = 8): the second residents start late (FsArgs.skew)
  A.skew = (!tr && nw == 4 && nwg > 256 && nwg <= 512) ? tante_opt("TANTE_FS_SKEW", 0) : 0;
  const int key = nw * 100 + tps * 10 + ntt;
  switch (key) {
    case 412:
      if (A.tprop) fs_launch_tt<1, 2, 4, false, 1, true>(A, nwg, s);
      else fs_launch_tt<1, 2, 4, false>(A, nwg, s);
      break;
    case 422: fs_launch_tt<2, 2, 4, false>(A, nwg, s); break;
    case 818: fs_launch_t<1, 8, 8>(A, nwg, s); break;
    case 816: fs_launch_t<1, 6, 8>(A, nwg, s); break;
    case 828: fs_launch_t<2, 8, 8>(A, nwg, s); break;
    case 826: fs_launch_t<2, 6, 8>(A, nwg, s); break;
    case 836: fs_launch_t<3, 6, 8>(A, nwg, s); break;
    case 848: fs_launch_t<4, 8, 8>(A, nwg, s); break;
    case 808: fs_launch_t<0, 8, 8>(A, nwg, s); break;
    case 414: fs_launch_t<1, 4, 4>(A, nwg, s); break;
    case 413: fs_launch_t<1, 3, 4>(A, nwg, s); break;
    case 424: fs_launch_t<2, 4, 4>(A, nwg, s); break;
    case 433: fs_launch_t<3, 3, 4>(A, nwg, s); break;
    case 444: fs_launch_t<4, 4, 4>(A, nwg, s); break;
    case 404: fs_launch_t<0, 4, 4>(A, nwg, s); break;
    default: return -3;
  }
  return 0;
}
