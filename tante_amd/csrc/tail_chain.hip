// Training form of the token-local tail of a rollout call (bf16, C = 256, patch_scale 8: three 2 x 2 stages, no overlap):
//
//   forward  (tante_tail_fwd):  derivative head of every Taylor order (enc_dec_cnn.py:263-277) -> Taylor sum (tante.py:165-171) ->
//                               predicted frame -> its re-encoding for the next call's window (enc_dec_cnn.py:217-229), with every
//                               pre-activation / activation the backward pass reads stored on the way;
//   backward (tante_tail_bwd):  encoder stages backwards -> + the frame's other gradients -> Taylor backwards -> every order's decoder
//                               stages backwards -> gradient of the residual stream's last time slot, with the row operands of the four
//                               wide weight gradients written as dense bf16 matrices (the shared weight-gradient launches read them) and
//                               the two pixel-level weight gradients + the decoder biases' as per-workgroup partials (a reduce launch).
//
// The inference path has had the forward as one launch since round 4 (head_enc.hip).  The training path ran it as 13 launches per call
// and its backward as ~33 (GEMM, activation, im2col, column sum, small weight gradient + reduce, Taylor, slice copies: 5 - 25 us each,
// ~4.5 us of which is the launch floor) -- 0.3 ms per call, 1.2 ms of the 8.5 ms train step at cfg3, for 7 GFLOP and 150 MB.
//
// Everything is local to a token: the 8 x 8 x D pixel block a token's heads write is the block the three encoder stages reduce back to
// that token.  A workgroup owns 16 consecutive tokens (one MFMA column tile) and walks the levels
//     level 0: 16 tokens x 256 ch | level 1: 64 pixels x 128 ch | level 2: 256 pixels x 64 ch | level 3: 1024 pixels x D fields
// with the data of a level as a bf16 image in LDS, columns in HIERARCHICAL pixel order (col = 4 * parent + (kh, kw)): a kernel = stride
// = 2 (transposed) convolution is then a plain GEMM between neighbouring levels and its "im2col" is a reinterpretation of the image --
// level l + 1 as [cols][C] IS level l as [cols / 4][4 C] with k = (kh, kw, c).  Two kinds of step:
//     expanding   (l -> l + 1: decoder forward, encoder backward):  D[(sub, c')][col] = W[(sub, c')][c] . img[col][c]  -> img'[4 col + sub][c']
//     contracting (l + 1 -> l: encoder forward, decoder backward):  D[c][col] = W[c][(sub, c')] . img'[col][(sub, c')] -> img[col][c]
// Waves own output-feature slices (weights come L2 -> registers as packed fragments, tante_tail_pack_*) or, at the pixel level where the
// matrix has one to three row tiles, column slices.  All saved / row-operand tensors are TOKEN-MAJOR (a workgroup's tile of every one of
// them is one contiguous chunk: coalesced 16-byte copies between LDS and memory), so the patch matrices of the weight gradients
// (dW = dY^T patches) need no gather either; frames (nchw fp32) go through an fp32 tile in LDS, 512-byte runs per (field, row).
#include "fused_common.hip.h"
#include "fs_common.hip.h"

namespace {

constexpr int TC_X = 0, TC_Y = 32768, TC_LDS = 32768 + 49152;      // LDS regions: 32 + 48 KiB -- two workgroups per CU

// decoder forward stream | decoder backward | encoder forward | encoder backward (bytes; fragments of 1 KiB, (row tile, k-step) order)
constexpr int DF_X01 = 0, DF_X12 = 262144, DF_X23 = 327680, DF_BIAS = 335872, DF_BYTES = 336896;      // biases: b1[128] | b2[64] | b3[16]
constexpr int DB_C32 = 0, DB_C21 = 8192, DB_C10 = 73728, DB_BYTES = 335872;
constexpr int EF_C32 = 0, EF_C21 = 8192, EF_C10 = 73728, EF_BIAS = 335872, EF_BYTES = 337920;         // biases: b1[64] | b2[128] | b3[256]
constexpr int EB_X01 = 0, EB_X12 = 262144, EB_X23 = 327680, EB_BYTES = 335872;

template <int CPR>
__device__ __forceinline__ int tc_off(int r, int c) {      // byte offset of 16-byte chunk c of row r in an image of CPR chunks per row
  if constexpr (CPR >= 16) return r * (CPR * 16) + ((c ^ (r & 15)) << 4);
  else return r * (CPR * 16) + ((c ^ ((r >> 1) & 7)) << 4);      // CPR == 8: two rows per 256-byte bank row
}
template <int CPR>
__device__ __forceinline__ void tc_put4(char* img, int R, int ch0, const u32x2& v) {      // 4 bf16 at channels ch0 .. ch0 + 3 (ch0 % 4 == 0)
  *(u32x2*)(img + tc_off<CPR>(R, ch0 >> 3) + (ch0 & 4) * 2) = v;
}
template <int CPR>
__device__ __forceinline__ f32x4 tc_get4(const char* img, int R, int ch0) {
  const u32x2 v = *(const u32x2*)(img + tc_off<CPR>(R, ch0 >> 3) + (ch0 & 4) * 2);
  return f32x4{bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1])};
}
__device__ __forceinline__ u32x2 tc_pack4(const f32x4& v) { return u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}; }
__device__ __forceinline__ f32x4 tc_unpack4(const u32x2& v) { return f32x4{bf16_lo(v[0]), bf16_hi(v[0]), bf16_lo(v[1]), bf16_hi(v[1])}; }

// Global accesses of the accumulator-layout epilogues and the tile copies go through buffer resources: address = tile base (4 SGPRs) +
// ONE lane offset per layout (a VGPR) + a wave-uniform offset (an SGPR or an immediate).  As plain pointers every access had a 64-bit
// vector address of its own, computed ahead and kept alive across the phase: the kernels filled 256 registers and spilled.
struct TcBuf {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ TcBuf tc_buf(const void* p) { return TcBuf{__builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000)}; }
__device__ __forceinline__ u32x2 tc_ld8(const TcBuf& b, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, 0); }
__device__ __forceinline__ void tc_st8(const TcBuf& b, int voff, int soff, const u32x2& v) { __builtin_amdgcn_raw_buffer_store_b64(v, b.r, voff, soff, 0); }
__device__ __forceinline__ void tc_st16(const TcBuf& b, int voff, int soff, const u32x4& v) { __builtin_amdgcn_raw_buffer_store_b128(v, b.r, voff, soff, 0); }

// image <-> its token-major tile in memory (rows of CPR x 16 bytes, linear), 16 bytes per thread and pass
template <int CPR, int ROWS>
__device__ __forceinline__ void tc_img_out(const char* img, void* dst, int tid) {
  static_assert(ROWS * CPR % 256 == 0, "whole passes");
  u32x4 v[ROWS * CPR / 256];
#pragma unroll
  for (int q = 0; q < ROWS * CPR / 256; ++q) {
    const int i = tid + 256 * q;
    v[q] = *(const u32x4*)(img + tc_off<CPR>(i / CPR, i % CPR));
  }
  const TcBuf b = tc_buf(dst);
#pragma unroll
  for (int q = 0; q < ROWS * CPR / 256; ++q) tc_st16(b, tid * 16, 4096 * q, v[q]);
}

template <int CPR, int ROWS>
__device__ __forceinline__ void tc_img_in(char* img, const void* src, int tid) {
  static_assert(ROWS * CPR % 256 == 0, "whole passes");
  u32x4 v[ROWS * CPR / 256];
#pragma unroll
  for (int q = 0; q < ROWS * CPR / 256; ++q) v[q] = *(const u32x4*)((const char*)src + (long)(tid + 256 * q) * 16);
#pragma unroll
  for (int q = 0; q < ROWS * CPR / 256; ++q) {
    const int i = tid + 256 * q;
    *(u32x4*)(img + tc_off<CPR>(i / CPR, i % CPR)) = v[q];
  }
}

// The two pixel-level weight gradients (64 x 4 D: dW = sum over the tile's 256 level-2 pixels of  U[row][i] V[row][j]) inside the kernel:
// the contraction index is the ROW of both [256][64] images, so the MFMA fragments (8 consecutive rows of one column) are gfx950's
// transposing LDS reads (cdna_hip_programming.md T10: per 16-lane group lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3
// of a 4-row x 16-column block; lane i receives column i).  acc[t] += U[:, 16 mt .. + 15]^T V[:, 16 t .. + 15].
// (Both reads of a fragment and their wait are ONE asm statement: a transposing read delivers its registers when the LDS answers, not
// when the instruction issues, and the compiler does not know -- with the wait in a later statement it was free to spill the "result"
// in between: the D = 11 instantiation, the only one under register pressure here, summed stale registers into the gradient.)
__device__ __forceinline__ u32x4 tc_tr_frag(const char* p0, const char* p1) {
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(lo), "=&v"(hi)
               : "v"(lds_addr(p0)), "v"(lds_addr(p1))
               : "memory");
  return u32x4{lo[0], lo[1], hi[0], hi[1]};
}
template <int NT>
__device__ __forceinline__ void tc_rows_outer(const char* U, int mt, const char* V, int lane, f32x4 (&acc)[NT]) {
  const int l15 = lane & 15, kk = lane >> 4, q = l15 >> 2, p = l15 & 3;
#pragma unroll 2
  for (int ks = 0; ks < 8; ++ks) {
    const int r0 = 32 * ks + 8 * kk + q;
    const u32x4 a = tc_tr_frag(U + tc_off<8>(r0, 2 * mt + (p >> 1)) + 8 * (p & 1), U + tc_off<8>(r0 + 4, 2 * mt + (p >> 1)) + 8 * (p & 1));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const u32x4 b = tc_tr_frag(V + tc_off<8>(r0, 2 * t + (p >> 1)) + 8 * (p & 1), V + tc_off<8>(r0 + 4, 2 * t + (p >> 1)) + 8 * (p & 1));
      acc[t] = mfma_bf16(a, b, acc[t]);
    }
  }
}
// ... and out to the workgroup's slot of the scratch matrix: [64 rows][16 NT] floats behind the bias partials
template <int NT>
__device__ __forceinline__ void tc_dw_part(const f32x4 (&acc)[NT], float* dst, int mt, int lane) {
  const int l15 = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(16 * mt + 4 * kk + r) * (16 * NT) + 16 * t + l15] = acc[t][r];
}

// acc[j][c] += W[row tile rt0 + j][all k] . img[col tile ct0 + c].  w = the weight stream as a buffer resource (fs_common.hip.h: one
// 32-bit lane offset for the whole kernel, fragment offsets in SGPRs), woff = byte offset of fragment (rt0, k-step 0) of the matrix.
// VIEW4: the image was written one level finer ([4 rows][CPR / 4 chunks] per row of this view); same bytes, the finer image's swizzle.
// The k loop is a real loop (one k-step's fragments in flight ahead of the MFMAs): unrolled, the scheduler hoists every k-step's
// loads and address arithmetic to the top of the phase -- 256 registers and scratch.
template <int KS, int CPR, int RTW, int CTW, bool VIEW4 = false, int RING = 2>
__device__ __forceinline__ void tc_gemm(const FsW& w, int woff, const char* img, int ct0, int lane, f32x4 (&acc)[RTW][CTW]) {
  static_assert(KS % RING == 0, "the ring divides the k-steps");
  const int l15 = lane & 15, kk = lane >> 4;
  u32x4 a[RING][RTW];      // RING k-steps of weight fragments in flight: every phase starts cold and its k loop is short
#pragma unroll
  for (int u = 0; u < RING; ++u)
#pragma unroll
    for (int j = 0; j < RTW; ++j) a[u][j] = ldg_frag(w, woff + (j * KS + u) * 1024);
#pragma unroll 1
  for (int ks0 = 0; ks0 < KS; ks0 += RING) {
    static_for<RING>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int ks = ks0 + u;
      u32x4 b[CTW];
#pragma unroll
      for (int c = 0; c < CTW; ++c) {
        const int r = 16 * (ct0 + c) + l15, ch = 4 * ks + kk;
        if constexpr (VIEW4) b[c] = *(const u32x4*)(img + tc_off<CPR / 4>(4 * r + ch / (CPR / 4), ch % (CPR / 4)));
        else b[c] = *(const u32x4*)(img + tc_off<CPR>(r, ch));
      }
#pragma unroll
      for (int j = 0; j < RTW; ++j)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[j][c] = mfma_bf16(a[u][j], b[c], acc[j][c]);
      if (ks + RING < KS) {
#pragma unroll
        for (int j = 0; j < RTW; ++j) a[u][j] = ldg_frag(w, woff + (j * KS + ks + RING) * 1024);
      }
    });
  }
}
template <int RTW, int CTW>
__device__ __forceinline__ void tc_zero(f32x4 (&acc)[RTW][CTW]) {
#pragma unroll
  for (int j = 0; j < RTW; ++j)
#pragma unroll
    for (int c = 0; c < CTW; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ f32x4 tc_gelu_grad4(const f32x4& p) {
  return f32x4{gelu_erf_grad_fast(p[0]), gelu_erf_grad_fast(p[1]), gelu_erf_grad_fast(p[2]), gelu_erf_grad_fast(p[3])};
}

// a frame tile (D fields x 8 rows x 128 pixels of 16 tokens in one patch row) between memory and the fp32 tile [(d, y)][128]
__device__ __forceinline__ void tc_ftile_in(float* ft, const float* g, int D, int H, int W, int y0, int x0, int tid) {
#pragma unroll 4
  for (int i = tid; i < D * 8 * 32; i += 256) {
    const int row = i >> 5, c = i & 31, d = row >> 3, y = row & 7;
    *(f32x4*)(ft + row * 128 + 4 * c) = *(const f32x4*)(g + ((long)d * H + y0 + y) * W + x0 + 4 * c);
  }
}
__device__ __forceinline__ void tc_ftile_out(const float* ft, float* g, int D, int H, int W, int y0, int x0, int tid) {
#pragma unroll 4
  for (int i = tid; i < D * 8 * 32; i += 256) {
    const int row = i >> 5, c = i & 31, d = row >> 3, y = row & 7;
    *(f32x4*)(g + ((long)d * H + y0 + y) * W + x0 + 4 * c) = *(const f32x4*)(ft + row * 128 + 4 * c);
  }
}
// position of level-3 element (sub3 = (kh3, kw3), field d) of pixel px16 = (kh1, kw1, kh2, kw2) of tile token tok in the fp32 tile
__device__ __forceinline__ int tc_fpos(int tok, int px16, int sub3, int d) {
  const int y = ((px16 >> 3) & 1) * 4 + ((px16 >> 1) & 1) * 2 + (sub3 >> 1);
  const int x = ((px16 >> 2) & 1) * 4 + (px16 & 1) * 2 + (sub3 & 1);
  return (d * 8 + y) * 128 + tok * 8 + x;
}

// ================================================================ forward ==============================================================
// LDS: region X (32 KiB) + region Y (48 KiB) = 80 KiB, two workgroups per CU.  Pre-activations never pass through LDS: the forward stores
// them from the accumulator layout (8 bytes per lane; small tensors), the backward loads them the same way ahead of the GEMM they scale.
template <int RT3>
__global__ __launch_bounds__(256, 2) void tail_fwd_kernel(const TanteTailFwd A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const rX = smem + TC_X;
  char* const rY = smem + TC_Y;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kk = lane >> 4;
  const int tile = blockIdx.x;
  const int D = A.D, H = 8 * A.Hp, W = 8 * A.Wp, HW = A.Hp * A.Wp;
  const long tok0 = 16L * tile;
  const int img = (int)(tok0 / HW), hw0 = (int)(tok0 - (long)img * HW), hp = hw0 / A.Wp, wp0 = hw0 - hp * A.Wp;

  // level-3 element of accumulator row 16 j + 4 kk + r: (sub3, d) and whether it is a real one
  int e_sub[RT3][4], e_d[RT3][4];
#pragma unroll
  for (int j = 0; j < RT3; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = 16 * j + 4 * kk + r;
      e_sub[j][r] = idx < 4 * D ? idx / D : -1;
      e_d[j][r] = idx - (idx / D) * D;
    }
  f32x4 facc[RT3][4];      // the Taylor sum's derivative part: rows (sub3, d), columns (token 4 w + c, pixel l15)
  tc_zero(facc);

  for (int k = 0; k < A.n_ord; ++k) {
    const TanteTailOrdF& O = A.o[k];
    const char* const wf = (const char*)O.w;
    const FsW w_wf = fs_wstream(wf, (unsigned)(lane * 16));
    char* const L0 = rY;                 // [16][256]   8 KiB
    char* const L1 = rY + 8192;          // [64][128]  16 KiB
    // ---- the tokens' rows (fp32, the last time slot of the residual stream) -> bf16 image ----
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + 256 * q, r = i >> 6, c4 = i & 63;
      const long t = tok0 + r;
      const float* row = (const float*)O.x + (t / A.a_n0) * A.a_s1 + (t % A.a_n0) * A.a_s0 + A.a_off;
      const f32x4 v = *(const f32x4*)(row + 4 * c4);
      *(u32x2*)(L0 + tc_off<32>(r, c4 >> 1) + (c4 & 1) * 8) = tc_pack4(v);
    }
    __syncthreads();
    if (O.xl16) tc_img_out<32, 16>(L0, (char*)O.xl16 + tok0 * 512, tid);
    // ---- stage 1: 256 -> (sub, 128) ----
    {
      f32x4 acc[8][1];
      tc_zero(acc);
      tc_gemm<8, 32, 8, 1>(w_wf, DF_X01 + (long)(8 * wave) * 8 * 1024, L0, 0, lane, acc);
      const float* b1 = (const float*)(wf + DF_BIAS);
      const TcBuf gpre = tc_buf((char*)O.pre1 + tok0 * 1024);      // [(token, sub)][128]: lane part l15 * 1024 + kk * 8
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rt = 8 * wave + j, sub = rt >> 3, ch0 = (rt & 7) * 16 + 4 * kk, R = 4 * l15 + sub;
        const f32x4 v = acc[j][0] + *(const f32x4*)(b1 + ch0);
        const u32x2 pb = tc_pack4(v);
        tc_st8(gpre, l15 * 1024 + kk * 8, sub * 256 + (rt & 7) * 32, pb);
        tc_put4<16>(L1, R, ch0, tc_pack4(gelu_poly4<false>(tc_unpack4(pb))));
      }
    }
    __syncthreads();
    tc_img_out<16, 64>(L1, (char*)O.act1 + tok0 * 1024, tid);
    // ---- stage 2: 128 -> (sub, 64): activation -> region X ----
    {
      f32x4 acc[4][4];
      tc_zero(acc);
      tc_gemm<4, 16, 4, 4, false, 4>(w_wf, DF_X12 + (long)(4 * wave) * 4 * 1024, L1, 0, lane, acc);
      const float* b2 = (const float*)(wf + DF_BIAS) + 128;
      const TcBuf gpre = tc_buf((char*)O.pre2 + tok0 * 2048);      // [(token, sub1, sub)][64]: lane part l15 * 512 + kk * 8
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rt = 4 * wave + j, sub = rt >> 2, ch0 = (rt & 3) * 16 + 4 * kk;
        const f32x4 bb = *(const f32x4*)(b2 + ch0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int R = 4 * (16 * c + l15) + sub;
          const u32x2 pb = tc_pack4(acc[j][c] + bb);
          tc_st8(gpre, l15 * 512 + kk * 8, c * 8192 + sub * 128 + (rt & 3) * 32, pb);
          tc_put4<8>(rX, R, ch0, tc_pack4(gelu_poly4<false>(tc_unpack4(pb))));
        }
      }
    }
    __syncthreads();
    tc_img_out<8, 256>(rX, (char*)O.act2 + tok0 * 2048, tid);
    // ---- stage 3: 64 -> (sub3, d): this order's derivative, times its Taylor coefficient ----
    {
      f32x4 acc[RT3][4];
      tc_zero(acc);
      tc_gemm<2, 8, RT3, 4>(w_wf, DF_X23, rX, 4 * wave, lane, acc);
      const float* b3 = (const float*)(wf + DF_BIAS) + 192;
#pragma unroll
      for (int j = 0; j < RT3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float bb = e_sub[j][r] >= 0 ? b3[e_d[j][r]] : 0.0f;
#pragma unroll
          for (int c = 0; c < 4; ++c) facc[j][c][r] += O.coef * (acc[j][c][r] + bb);
        }
    }
    __syncthreads();      // the next order's rows overwrite region Y, its stage 2 region X
  }

  // ---- Taylor sum: + the window's last frame; the predicted frame goes out (fp32) and into the level-3 patch image (bf16) ----
  float* const ft = (float*)rY;
  char* const L3 = rX;      // [256 (token, px16)][64]: (sub3, d) at 2-byte position sub3 * D + d, zero beyond 4 D
  const int y0 = 8 * hp, x0 = 8 * wp0;
  tc_ftile_in(ft, A.base + (long)img * A.base_bstride, D, H, W, y0, x0, tid);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int tok = 4 * wave + c;
#pragma unroll
    for (int j = 0; j < RT3; ++j) {
      f32x4 f = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (e_sub[j][r] >= 0) {
          const int p = tc_fpos(tok, l15, e_sub[j][r], e_d[j][r]);
          f[r] = facc[j][c][r] + ft[p];
          ft[p] = f[r];
        }
      tc_put4<8>(L3, 16 * tok + l15, 16 * j + 4 * kk, tc_pack4(f));
    }
#pragma unroll
    for (int j = RT3; j < 4; ++j) tc_put4<8>(L3, 16 * tok + l15, 16 * j + 4 * kk, u32x2{0u, 0u});
  }
  __syncthreads();
  if (A.out) tc_ftile_out(ft, A.out + (long)img * A.out_bstride, D, H, W, y0, x0, tid);
  if (!A.we) return;
  tc_img_out<8, 256>(L3, (char*)A.f16 + tok0 * 2048, tid);
  __syncthreads();      // the frame tile has left region Y

  // =========================================================== re-encoding ===========================================================
  const char* const we = (const char*)A.we;
  const FsW w_we = fs_wstream(we, (unsigned)(lane * 16));
  {   // ---- stage 1: (sub3, d) -> 64 at the 256 level-2 pixels; activation -> region Y ----
    f32x4 acc[4][4];
    tc_zero(acc);
    tc_gemm<2, 8, 4, 4>(w_we, EF_C32, L3, 4 * wave, lane, acc);
    const float* b1 = (const float*)(we + EF_BIAS);
    const TcBuf gpre = tc_buf((char*)A.pre1e + tok0 * 2048);      // [(token, px16)][64]: lane part l15 * 128 + kk * 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch0 = 16 * j + 4 * kk;
      const f32x4 bb = *(const f32x4*)(b1 + ch0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int R = 16 * (4 * wave + c) + l15;
        const u32x2 pb = tc_pack4(acc[j][c] + bb);
        tc_st8(gpre, l15 * 128 + kk * 8, (4 * wave + c) * 2048 + j * 32, pb);
        tc_put4<8>(rY, R, ch0, tc_pack4(gelu_poly4<false>(tc_unpack4(pb))));
      }
    }
  }
  __syncthreads();
  tc_img_out<8, 256>(rY, (char*)A.act1e + tok0 * 2048, tid);
  {   // ---- stage 2: region Y as [64][(sub, 64)] -> 128; activation -> X[0 : 16K] ----
    f32x4 acc[2][4];
    tc_zero(acc);
    tc_gemm<8, 32, 2, 4, true, 4>(w_we, EF_C21 + (long)(2 * wave) * 8 * 1024, rY, 0, lane, acc);
    const float* b2 = (const float*)(we + EF_BIAS) + 64;
    const TcBuf gpre = tc_buf((char*)A.pre2e + tok0 * 1024);      // [(token, px4)][128]: lane part l15 * 256 + kk * 8
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ch0 = 16 * (2 * wave + j) + 4 * kk;
      const f32x4 bb = *(const f32x4*)(b2 + ch0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int R = 16 * c + l15;
        const u32x2 pb = tc_pack4(acc[j][c] + bb);
        tc_st8(gpre, l15 * 256 + kk * 8, c * 4096 + (2 * wave + j) * 32, pb);
        tc_put4<16>(rX, R, ch0, tc_pack4(gelu_poly4<false>(tc_unpack4(pb))));
      }
    }
  }
  __syncthreads();
  tc_img_out<16, 64>(rX, (char*)A.act2e + tok0 * 1024, tid);
  {   // ---- stage 3: X[0 : 16K] as [16][(sub, 128)] -> 256: the frame's encoding before FiLM (fp32 rows, through region Y) ----
    f32x4 acc[4][1];
    tc_zero(acc);
    tc_gemm<16, 64, 4, 1, true, 4>(w_we, EF_C10 + (long)(4 * wave) * 16 * 1024, rX, 0, lane, acc);
    const float* b3 = (const float*)(we + EF_BIAS) + 192;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch0 = 16 * (4 * wave + j) + 4 * kk;
      *(f32x4*)(rY + (l15 * 256 + ch0) * 4) = acc[j][0] + *(const f32x4*)(b3 + ch0);      // (region Y: its copy-out ended before the stage-2 barrier)
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = tid + 256 * q;
    *(f32x4*)(A.z + tok0 * 256 + 4 * i) = *(const f32x4*)(rY + 16 * i);
  }
}

// ================================================================ backward =============================================================
// Bias gradients of the decoder stages (column sums of the V operands folded over the taps).  Adding them from every workgroup with
// atomics took 150 us of the first form's 210 us launch: 1 536 same-address adds per bias element, and device-scope atomics to one address
// are served one behind the other at the memory side.  Each wave STORES the column sums of its accumulators into the workgroup's row of a
// scratch matrix (per workgroup one slot per order and one for the encoder: [wave][64] stage-3 sums | [128] stage-2 sums | [16] field sums |
// the pixel-level weight gradient's partial) and a small second
// kernel (tail_bias_reduce_kernel, launched by tante_tail_bwd) adds the column sums of that matrix to the gradients.
constexpr int TC_WS = 512 + 64 * 48;      // per workgroup and slot: 512 bias partials | the pixel-level weight gradient's partial, [64][16 RT3]
template <int RTW, int CTW>
__device__ __forceinline__ void tc_bias_part(const f32x4 (&g)[RTW][CTW], float* dst, int lane) {
  const int l15 = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int j = 0; j < RTW; ++j) {
    f32x4 s = g[j][0];
#pragma unroll
    for (int c = 1; c < CTW; ++c) s += g[j][c];
    f32x4 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = row16_sum(s[r]);
    if (l15 == 0) *(f32x4*)(dst + 16 * j + 4 * kk) = t;
  }
}
struct TcRed {
  const float* ws;
  float* db1[TANTE_TAIL_MAX_ORD + 1];
  float* db2[TANTE_TAIL_MAX_ORD + 1];
  float* db3[TANTE_TAIL_MAX_ORD + 1];
  float* dw[TANTE_TAIL_MAX_ORD + 1];      // (64, D, 2, 2) weight: element (row, (sub, d)) at ((row D + d) 4 + sub)
  int n_slot, tiles, D, ncol;
};
// grid (ncol / 64 column blocks, n_slot, 16 row slices), 256 threads = 64 columns x 4 sub-slices
__global__ __launch_bounds__(256) void tail_bias_reduce_kernel(const TcRed R) {
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6, k = blockIdx.y;
  float s0 = 0.f, s1 = 0.f;
  const int stride = 4 * gridDim.z;
  int row = blockIdx.z * 4 + sub;
  for (; row + stride < R.tiles; row += 2 * stride) {
    s0 += R.ws[((long)row * R.n_slot + k) * TC_WS + col];
    s1 += R.ws[((long)(row + stride) * R.n_slot + k) * TC_WS + col];
  }
  if (row < R.tiles) s0 += R.ws[((long)row * R.n_slot + k) * TC_WS + col];
  part[sub][threadIdx.x & 63] = s0 + s1;
  __syncthreads();
  if (sub != 0) return;
  const float v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
  float* dst = nullptr;
  if (col < 256) dst = R.db2[k] ? R.db2[k] + (col & 63) : nullptr;
  else if (col < 384) dst = R.db1[k] ? R.db1[k] + (col - 256) : nullptr;
  else if (col < 512) dst = (R.db3[k] && col < 384 + R.D) ? R.db3[k] + (col - 384) : nullptr;
  else if (R.dw[k]) {
    const int w = (R.ncol - 512) / 64, q = col - 512, r = q / w, n = q - r * w;      // w = 16 RT3 columns per row
    if (n < 4 * R.D) {
      const int sb = n / R.D, d = n - sb * R.D;
      dst = R.dw[k] + ((long)r * R.D + d) * 4 + sb;
    }
  }
  if (dst) atomicAdd(dst, v);
}

template <int RT3>
__global__ __launch_bounds__(256, 2) void tail_bwd_kernel(const TanteTailBwd A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const rX = smem + TC_X;
  char* const rY = smem + TC_Y;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kk = lane >> 4;
  const int tile = blockIdx.x;
  const int D = A.D, H = 8 * A.Hp, W = 8 * A.Wp, HW = A.Hp * A.Wp;
  const long tok0 = 16L * tile;
  const int img = (int)(tok0 / HW), hw0 = (int)(tok0 - (long)img * HW), hp = hw0 / A.Wp, wp0 = hw0 - hp * A.Wp;
  const int y0 = 8 * hp, x0 = 8 * wp0;
  int e_sub[RT3][4], e_d[RT3][4];
#pragma unroll
  for (int j = 0; j < RT3; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = 16 * j + 4 * kk + r;
      e_sub[j][r] = idx < 4 * D ? idx / D : -1;
      e_d[j][r] = idx - (idx / D) * D;
    }
  f32x4 dfr[RT3][4];      // gradient of the predicted frame: rows (sub3, d), columns (token 4 w + c, pixel l15)
  tc_zero(dfr);
  float* const ft = (float*)rY;

  if (A.dz) {
    // =================================================== encoder stages, backwards ===================================================
    const FsW w_we = fs_wstream((const char*)A.we, (unsigned)(lane * 16));
    char* const L0 = rY;      // dz as bf16 [16][256]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + 256 * q, r = i >> 6, c4 = i & 63;
      const f32x4 v = *(const f32x4*)(A.dz + (tok0 + r) * 256 + 4 * c4);
      *(u32x2*)(L0 + tc_off<32>(r, c4 >> 1) + (c4 & 1) * 8) = tc_pack4(v);
    }
    u32x2 p2[8];      // pre2e at this lane's accumulator positions of stage 3
    {
      const TcBuf gpre = tc_buf((const char*)A.pre2e + tok0 * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rt = 8 * wave + j, sub = rt >> 3;
        p2[j] = tc_ld8(gpre, l15 * 1024 + kk * 8, sub * 256 + (rt & 7) * 32);
      }
    }
    __syncthreads();
    tc_img_out<32, 16>(L0, (char*)A.dz16 + tok0 * 512, tid);
    {   // stage 3 backwards: 256 -> (sub, 128), times GELU'(pre2e) -> X[0 : 16K]
      f32x4 acc[8][1];
      tc_zero(acc);
      tc_gemm<8, 32, 8, 1>(w_we, EB_X01 + (long)(8 * wave) * 8 * 1024, L0, 0, lane, acc);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rt = 8 * wave + j, sub = rt >> 3, ch0 = (rt & 7) * 16 + 4 * kk, R = 4 * l15 + sub;
        tc_put4<16>(rX, R, ch0, tc_pack4(acc[j][0] * tc_gelu_grad4(tc_unpack4(p2[j]))));
      }
    }
    u32x2 p1[4][4];   // pre1e at the accumulator positions of stage 2
    {
      const TcBuf gpre = tc_buf((const char*)A.pre1e + tok0 * 2048);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rt = 4 * wave + j, sub = rt >> 2;
#pragma unroll
        for (int c = 0; c < 4; ++c) p1[j][c] = tc_ld8(gpre, l15 * 512 + kk * 8, c * 8192 + sub * 128 + (rt & 3) * 32);
      }
    }
    __syncthreads();
    tc_img_out<16, 64>(rX, (char*)A.dpre2e + tok0 * 1024, tid);
    {   // stage 2 backwards: 128 -> (sub, 64), times GELU'(pre1e) -> Y[0 : 32K]
      f32x4 acc[4][4];
      tc_zero(acc);
      tc_gemm<4, 16, 4, 4, false, 4>(w_we, EB_X12 + (long)(4 * wave) * 4 * 1024, rX, 0, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rt = 4 * wave + j, sub = rt >> 2, ch0 = (rt & 3) * 16 + 4 * kk;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[j][c] = acc[j][c] * tc_gelu_grad4(tc_unpack4(p1[j][c]));
          tc_put4<8>(rY, 4 * (16 * c + l15) + sub, ch0, tc_pack4(acc[j][c]));
        }
      }
      // the first stage's bias gradient: column sums of these values (wave w holds tap w of all 64 channels)
      if (A.bias_ws && A.dwe1) tc_bias_part<4, 4>(acc, A.bias_ws + ((long)tile * (A.n_ord + 1) + A.n_ord) * TC_WS + 64 * wave, lane);
    }
    __syncthreads();      // region X (stage 2's input) is consumed, region Y holds dpre1e
    if (A.dpre1e) tc_img_out<8, 256>(rY, (char*)A.dpre1e + tok0 * 2048, tid);
    if (A.dwe1 && A.bias_ws) {      // the first stage's weight gradient of this tile: dpre1e^T . frame patches
      tc_img_in<8, 256>(rX, (const char*)A.f16 + tok0 * 2048, tid);
      __syncthreads();
      f32x4 dwp[RT3];
#pragma unroll
      for (int t = 0; t < RT3; ++t) dwp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      tc_rows_outer<RT3>(rY, wave, rX, lane, dwp);
      tc_dw_part<RT3>(dwp, A.bias_ws + ((long)tile * (A.n_ord + 1) + A.n_ord) * TC_WS + 512, wave, lane);
    }
    if (A.n_ord == 0 && !A.dbase) return;      // encoder of INPUT frames: the weight-gradient operands are all that is asked for
    // stage 1 backwards: 64 -> (sub3, d): the encoder's share of the frame's gradient
    tc_gemm<2, 8, RT3, 4>(w_we, EB_X23, rY, 4 * wave, lane, dfr);
    __syncthreads();      // region Y is free for the frame tile
  }

  // ---- + the frame's other gradients (loss, the next call's Taylor base); the total is the gradient of this call's base frame too ----
  if (A.dext) tc_ftile_in(ft, A.dext + (long)img * A.dext_bstride, D, H, W, y0, x0, tid);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int tok = 4 * wave + c;
#pragma unroll
    for (int j = 0; j < RT3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (e_sub[j][r] >= 0) {
          const int p = tc_fpos(tok, l15, e_sub[j][r], e_d[j][r]);
          if (A.dext) dfr[j][c][r] += ft[p];
          ft[p] = dfr[j][c][r];
        } else {
          dfr[j][c][r] = 0.0f;
        }
  }
  __syncthreads();
  if (A.dbase) tc_ftile_out(ft, A.dbase + (long)img * A.dbase_bstride, D, H, W, y0, x0, tid);
  // per-field sums of the tile (the last decoder stage's bias gradient, up to the Taylor coefficient)
  float fsum = 0.0f;
  {
    const int d = tid >> 4, part = tid & 15;      // 16 threads per field (D <= 16)
    if (d < D) {
      for (int i = part; i < 256; i += 16) {
        const f32x4 v = *(const f32x4*)(ft + d * 1024 + 4 * i);
        fsum += (v[0] + v[1]) + (v[2] + v[3]);
      }
    }
    fsum = row16_sum(fsum);
  }
  __syncthreads();      // the frame tile has left region Y

  // ===================================================== decoder stages, backwards ===================================================
  for (int k = 0; k < A.n_ord; ++k) {
    const TanteTailOrdB& O = A.o[k];
    const FsW w_wb = fs_wstream((const char*)O.w, (unsigned)(lane * 16));
    float* const wsrow = A.bias_ws ? A.bias_ws + ((long)tile * (A.n_ord + 1) + k) * TC_WS : nullptr;
    const bool dw_here = wsrow && O.dw3 && O.act2;
    if (dw_here) tc_img_in<8, 256>(rY, (const char*)O.act2 + tok0 * 2048, tid);      // (region Y is free: the frame tile / the last order's rows have left)
    if (wsrow && (tid & 15) == 0 && (tid >> 4) < 16) wsrow[384 + (tid >> 4)] = (tid >> 4) < D ? O.coef * fsum : 0.0f;
    char* const L3 = rX;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int tok = 4 * wave + c;
#pragma unroll
      for (int j = 0; j < RT3; ++j) tc_put4<8>(L3, 16 * tok + l15, 16 * j + 4 * kk, tc_pack4(dfr[j][c] * O.coef));
#pragma unroll
      for (int j = RT3; j < 4; ++j) tc_put4<8>(L3, 16 * tok + l15, 16 * j + 4 * kk, u32x2{0u, 0u});
    }
    u32x2 q2[4][4];   // pre2 at the accumulator positions of stage 3 backwards
    {
      const TcBuf gpre = tc_buf((const char*)O.pre2 + tok0 * 2048);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) q2[j][c] = tc_ld8(gpre, l15 * 128 + kk * 8, (4 * wave + c) * 2048 + j * 32);
    }
    __syncthreads();
    if (O.dder) tc_img_out<8, 256>(L3, (char*)O.dder + tok0 * 2048, tid);
    if (dw_here) {      // the last stage's weight gradient of this tile: act2^T . derivative-gradient patches
      f32x4 dwp[RT3];
#pragma unroll
      for (int t = 0; t < RT3; ++t) dwp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      tc_rows_outer<RT3>(rY, wave, L3, lane, dwp);
      tc_dw_part<RT3>(dwp, wsrow + 512, wave, lane);
      __syncthreads();      // region Y is written by the epilogue below
    }
    {   // stage 3 backwards: (sub3, d) -> 64, times GELU'(pre2) -> Y[0 : 32K]
      f32x4 acc[4][4];
      tc_zero(acc);
      tc_gemm<2, 8, 4, 4>(w_wb, DB_C32, L3, 4 * wave, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[j][c] = acc[j][c] * tc_gelu_grad4(tc_unpack4(q2[j][c]));
          tc_put4<8>(rY, 16 * (4 * wave + c) + l15, 16 * j + 4 * kk, tc_pack4(acc[j][c]));
        }
      if (wsrow) tc_bias_part<4, 4>(acc, wsrow + 64 * wave, lane);
    }
    u32x2 q1[2][4];   // pre1 at the accumulator positions of stage 2 backwards
    {
      const TcBuf gpre = tc_buf((const char*)O.pre1 + tok0 * 1024);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) q1[j][c] = tc_ld8(gpre, l15 * 256 + kk * 8, c * 4096 + (2 * wave + j) * 32);
    }
    __syncthreads();
    tc_img_out<8, 256>(rY, (char*)O.dpre2 + tok0 * 2048, tid);
    {   // stage 2 backwards: region Y as [64][(sub, 64)] -> 128, times GELU'(pre1) -> X[0 : 16K]
      f32x4 acc[2][4];
      tc_zero(acc);
      tc_gemm<8, 32, 2, 4, true, 4>(w_wb, DB_C21 + (long)(2 * wave) * 8 * 1024, rY, 0, lane, acc);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[j][c] = acc[j][c] * tc_gelu_grad4(tc_unpack4(q1[j][c]));
          tc_put4<16>(rX, 16 * c + l15, 16 * (2 * wave + j) + 4 * kk, tc_pack4(acc[j][c]));
        }
      if (wsrow) tc_bias_part<2, 4>(acc, wsrow + 256 + 32 * wave, lane);
    }
    __syncthreads();
    tc_img_out<16, 64>(rX, (char*)O.dpre1 + tok0 * 1024, tid);
    {   // stage 1 backwards: X[0 : 16K] as [16][(sub, 128)] -> 256: the gradient of the residual stream's rows (fp32, through region Y)
      f32x4 acc[4][1];
      tc_zero(acc);
      tc_gemm<16, 64, 4, 1, true, 4>(w_wb, DB_C10 + (long)(4 * wave) * 16 * 1024, rX, 0, lane, acc);
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)(rY + (l15 * 256 + 16 * (4 * wave + j) + 4 * kk) * 4) = acc[j][0];      // (Y's copy-out ended before the last barrier)
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + 256 * q, r = i >> 6, c4 = i & 63;
      const long t = tok0 + r;
      float* row = O.dx + (t / A.a_n0) * A.a_s1 + (t % A.a_n0) * A.a_s0 + A.a_off;
      *(f32x4*)(row + 4 * c4) = *(const f32x4*)(rY + (r * 256 + 4 * c4) * 4);
    }
    __syncthreads();
  }
}

// ================================================================= packing =============================================================
// One conv / transposed-conv weight src[a][b][kh][kw] (na x nb x 2 x 2) as MFMA A-operand fragments of the matrix
//   orient 0:  Wm[a][(sub, b)]      (rows a, k = sub * nb + b)        orient 1:  Wm[(sub, b)][a]      (rows sub * nb + b, k = a)
// fragment (rt, ks) at dst + (rt * n_ks + ks) KiB: lane (l15, kk) holds Wm[16 rt + l15][32 ks + 8 kk .. + 7]; zero outside the matrix.
struct TcPackJob {
  const float* src;
  char* dst;
  int na, nb, orient, n_rt, n_ks, frag0;
};
struct TcPack {
  TcPackJob j[6];
  const float* bias[3];
  float* bias_dst[3];
  int bias_n[3];
  int n_frag;
};
__global__ void tail_pack_kernel(const TcPack P) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, l15 = lane & 15, kk = lane >> 4;
  if (f >= P.n_frag) {
    if (blockIdx.x == gridDim.x - 1 && (threadIdx.x >> 6) == 3) {      // (the last wave of the grid never holds a fragment) the biases
      for (int q = 0; q < 3; ++q)
        for (int i = lane; i < P.bias_n[q]; i += 64) P.bias_dst[q][i] = P.bias[q] ? P.bias[q][i] : 0.0f;
    }
    return;
  }
  int q = 0;
#pragma unroll
  for (int t = 1; t < 6; ++t)
    if (f >= P.j[t].frag0) q = t;
  const TcPackJob& J = P.j[q];
  const int fl = f - J.frag0, rt = fl / J.n_ks, ks = fl % J.n_ks;
  const int row = 16 * rt + l15;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * ks + 8 * kk + e;
    int a, sb;
    if (J.orient == 0) { a = row; sb = k; } else { a = k; sb = row; }
    const int sub = sb / J.nb, b = sb - sub * J.nb;
    v[e] = (a < J.na && sb < 4 * J.nb) ? J.src[((long)a * J.nb + b) * 4 + sub] : 0.0f;
  }
  u32x4 o;
  o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
  *(u32x4*)(J.dst + (long)fl * 1024 + lane * 16) = o;
}

int tc_pack(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int D, int dec,
            char* dstF, char* dstB, hipStream_t s) {
  if (!w1 || !w2 || !w3 || !dstF || !dstB || D < 1 || D > 12) return -1;
  TcPack P;
  // decoder: W1 (256, 128, 2, 2), W2 (128, 64, 2, 2), W3 (64, D, 2, 2) = src[ci][co][kh][kw]: forward rows (sub, co), k = ci (orient 1);
  //          backward rows ci, k = (sub, co) (orient 0)
  // encoder: W1 (64, D, 2, 2), W2 (128, 64, 2, 2), W3 (256, 128, 2, 2) = src[co][ci][kh][kw]: forward rows co, k = (sub, ci) (orient 0);
  //          backward rows (sub, ci), k = co (orient 1)
  const int rt3 = (4 * D + 15) / 16;
  int f0 = 0;
  auto job = [&](int i, const float* src, char* dst, int na, int nb, int orient, int n_rt, int n_ks) {
    P.j[i] = TcPackJob{src, dst, na, nb, orient, n_rt, n_ks, f0};
    f0 += n_rt * n_ks;
  };
  if (dec) {
    job(0, w1, dstF + DF_X01, 256, 128, 1, 32, 8);
    job(1, w2, dstF + DF_X12, 128, 64, 1, 16, 4);
    job(2, w3, dstF + DF_X23, 64, D, 1, rt3, 2);
    job(3, w3, dstB + DB_C32, 64, D, 0, 4, 2);
    job(4, w2, dstB + DB_C21, 128, 64, 0, 8, 8);
    job(5, w1, dstB + DB_C10, 256, 128, 0, 16, 16);
    float* bd = (float*)(dstF + DF_BIAS);
    P.bias[0] = b1; P.bias_dst[0] = bd; P.bias_n[0] = 128;
    P.bias[1] = b2; P.bias_dst[1] = bd + 128; P.bias_n[1] = 64;
    P.bias[2] = b3; P.bias_dst[2] = bd + 192; P.bias_n[2] = D;
  } else {
    job(0, w1, dstF + EF_C32, 64, D, 0, 4, 2);
    job(1, w2, dstF + EF_C21, 128, 64, 0, 8, 8);
    job(2, w3, dstF + EF_C10, 256, 128, 0, 16, 16);
    job(3, w3, dstB + EB_X01, 256, 128, 1, 32, 8);
    job(4, w2, dstB + EB_X12, 128, 64, 1, 16, 4);
    job(5, w1, dstB + EB_X23, 64, D, 1, rt3, 2);
    float* bd = (float*)(dstF + EF_BIAS);
    P.bias[0] = b1; P.bias_dst[0] = bd; P.bias_n[0] = 64;
    P.bias[1] = b2; P.bias_dst[1] = bd + 64; P.bias_n[1] = 128;
    P.bias[2] = b3; P.bias_dst[2] = bd + 192; P.bias_n[2] = 256;
  }
  P.n_frag = f0;
  hipLaunchKernelGGL(tail_pack_kernel, dim3((unsigned)(f0 / 4 + 1)), dim3(256), 0, s, P);
  return 0;
}

template <class KERN>
void tc_attr(KERN k) {
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, TC_LDS);
}

}  // namespace

extern "C" int tante_tail_supported(int C, int D, int Hp, int Wp) { return C == 256 && D >= 1 && D <= 12 && Hp >= 1 && Wp >= 16 && Wp % 16 == 0; }

extern "C" int64_t tante_tail_stream_bytes(int which) {
  switch (which) {
    case 4: return (int64_t)TC_WS * 4;      // gradient scratch: bytes per workgroup (16 tokens) and slot (one per Taylor order + one for the encoder)
    case 0: return DF_BYTES;
    case 1: return DB_BYTES;
    case 2: return EF_BYTES;
    case 3: return EB_BYTES;
    default: return 0;
  }
}

extern "C" int tante_tail_pack_dec(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int D,
                                   void* fwd_stream, void* bwd_stream, void* stream) {
  if (tc_pack(w1, b1, w2, b2, w3, b3, D, 1, (char*)fwd_stream, (char*)bwd_stream, (hipStream_t)stream)) TANTE_FAIL(-1, "tante_tail_pack_dec: bad argument");
  TANTE_CHECK_LAUNCH();
  return 0;
}
extern "C" int tante_tail_pack_enc(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int D,
                                   void* fwd_stream, void* bwd_stream, void* stream) {
  if (tc_pack(w1, b1, w2, b2, w3, b3, D, 0, (char*)fwd_stream, (char*)bwd_stream, (hipStream_t)stream)) TANTE_FAIL(-1, "tante_tail_pack_enc: bad argument");
  TANTE_CHECK_LAUNCH();
  return 0;
}

static int tc_check_geom(int n_ord, int n_img, int Hp, int Wp, int D, int a_n0, const char* who) {
  if (n_ord < 0 || n_ord > TANTE_TAIL_MAX_ORD) TANTE_FAIL(-1, "%s: 0 .. %d Taylor orders (got %d)", who, TANTE_TAIL_MAX_ORD, n_ord);
  if (!tante_tail_supported(256, D, Hp, Wp)) TANTE_FAIL(-2, "%s: unsupported shape (D = %d, Hp = %d, Wp = %d: D <= 12, Wp %% 16 == 0)", who, D, Hp, Wp);
  if (n_img < 1 || (n_ord > 0 && (a_n0 < 16 || a_n0 % 16))) TANTE_FAIL(-1, "%s: rows are addressed in blocks of a multiple of 16 tokens (a_n0 = %d)", who, a_n0);
  return 0;
}

extern "C" int tante_tail_fwd(const TanteTailFwd* a, void* stream) {
  if (!a) TANTE_FAIL(-1, "tante_tail_fwd: null");
  if (int rc = tc_check_geom(a->n_ord, a->n_img, a->Hp, a->Wp, a->D, a->a_n0, "tante_tail_fwd")) return rc;
  for (int k = 0; k < a->n_ord; ++k) {
    const TanteTailOrdF& o = a->o[k];
    if (!o.x || !o.w || !o.pre1 || !o.act1 || !o.pre2 || !o.act2) TANTE_FAIL(-1, "tante_tail_fwd: order %d: null pointer", k);
  }
  if (!a->base || (!a->out && a->n_ord > 0)) TANTE_FAIL(-1, "tante_tail_fwd: null frame pointer");
  if (a->n_ord == 0 && !a->we) TANTE_FAIL(-1, "tante_tail_fwd: neither a decoder nor the encoder to run");
  if (a->we && (!a->f16 || !a->pre1e || !a->act1e || !a->pre2e || !a->act2e || !a->z)) TANTE_FAIL(-1, "tante_tail_fwd: encoder outputs missing");
  const long tiles = (long)a->n_img * a->Hp * a->Wp / 16;
  const int rt3 = (4 * a->D + 15) / 16;
  static TantePerDevice attr;
  attr.once([&] { tc_attr(tail_fwd_kernel<1>); tc_attr(tail_fwd_kernel<2>); tc_attr(tail_fwd_kernel<3>); });
  hipStream_t s = (hipStream_t)stream;
  switch (rt3) {
    case 1: hipLaunchKernelGGL(tail_fwd_kernel<1>, dim3((unsigned)tiles), dim3(256), TC_LDS, s, *a); break;
    case 2: hipLaunchKernelGGL(tail_fwd_kernel<2>, dim3((unsigned)tiles), dim3(256), TC_LDS, s, *a); break;
    default: hipLaunchKernelGGL(tail_fwd_kernel<3>, dim3((unsigned)tiles), dim3(256), TC_LDS, s, *a); break;
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_tail_bwd(const TanteTailBwd* a, void* stream) {
  if (!a) TANTE_FAIL(-1, "tante_tail_bwd: null");
  if (int rc = tc_check_geom(a->n_ord, a->n_img, a->Hp, a->Wp, a->D, a->a_n0, "tante_tail_bwd")) return rc;
  for (int k = 0; k < a->n_ord; ++k) {
    const TanteTailOrdB& o = a->o[k];
    if (!o.w || !o.pre1 || !o.pre2 || !o.dpre1 || !o.dpre2 || !o.dx) TANTE_FAIL(-1, "tante_tail_bwd: order %d: null pointer", k);
    if (!o.dder && !(o.dw3 && o.act2 && a->bias_ws)) TANTE_FAIL(-1, "tante_tail_bwd: order %d: the last stage's weight gradient needs dder, or dw3 + act2 + bias_ws", k);
  }
  if (!a->dz && !a->dext) TANTE_FAIL(-1, "tante_tail_bwd: no gradient arrives (dz and dext are both null)");
  if (a->n_ord == 0 && !a->dz) TANTE_FAIL(-1, "tante_tail_bwd: neither a decoder nor the encoder to run");
  if (a->dz && (!a->we || !a->pre1e || !a->pre2e || !a->dz16 || !a->dpre2e)) TANTE_FAIL(-1, "tante_tail_bwd: encoder operands missing");
  if (a->dz && !a->dpre1e && !(a->dwe1 && a->f16 && a->bias_ws)) TANTE_FAIL(-1, "tante_tail_bwd: the first encoder stage's weight gradient needs dpre1e, or dwe1 + f16 + bias_ws");
  const long tiles = (long)a->n_img * a->Hp * a->Wp / 16;
  const int rt3 = (4 * a->D + 15) / 16;
  static TantePerDevice attr;
  attr.once([&] { tc_attr(tail_bwd_kernel<1>); tc_attr(tail_bwd_kernel<2>); tc_attr(tail_bwd_kernel<3>); });
  hipStream_t s = (hipStream_t)stream;
  switch (rt3) {
    case 1: hipLaunchKernelGGL(tail_bwd_kernel<1>, dim3((unsigned)tiles), dim3(256), TC_LDS, s, *a); break;
    case 2: hipLaunchKernelGGL(tail_bwd_kernel<2>, dim3((unsigned)tiles), dim3(256), TC_LDS, s, *a); break;
    default: hipLaunchKernelGGL(tail_bwd_kernel<3>, dim3((unsigned)tiles), dim3(256), TC_LDS, s, *a); break;
  }
  if (a->bias_ws) {
    TcRed R;
    R.ws = a->bias_ws; R.n_slot = a->n_ord + 1; R.tiles = (int)tiles; R.D = a->D; R.ncol = 512 + 64 * 16 * rt3;
    for (int k = 0; k <= TANTE_TAIL_MAX_ORD; ++k) {
      const bool ord = k < a->n_ord, enc = k == a->n_ord && a->dz && a->dwe1;
      R.db1[k] = ord ? a->o[k].db1 : nullptr;
      R.db2[k] = ord ? a->o[k].db2 : (enc ? a->dbe1 : nullptr);
      R.db3[k] = ord ? a->o[k].db3 : nullptr;
      R.dw[k] = ord ? (a->o[k].act2 ? a->o[k].dw3 : nullptr) : (enc ? a->dwe1 : nullptr);
    }
    hipLaunchKernelGGL(tail_bias_reduce_kernel, dim3((unsigned)(R.ncol / 64), (unsigned)R.n_slot, 16), dim3(256), 0, s, R);
  }
  TANTE_CHECK_LAUNCH();
  return 0;
}
