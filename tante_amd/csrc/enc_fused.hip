// Fused patch-embed stages 2 + 3 (bf16):  stage-1 image (channels-last bf16, C/4 channels) -> [Conv2d k = s = 2 -> GELU(erf)] ->
// [Conv2d k = s = 2] -> FiLM(t) + spatial / temporal embeddings -> the fp32 token stream   (enc_dec_cnn.py:221-229, tante.py:136-141).
//
// A token's value depends on its own 4 x 4 block of stage-1 pixels only.  A wave owns 16 tokens: for each of the token's 2 x 2
// stage-2 positions it loads the 2 x 2 x (C/4) input patch straight into MFMA B-operand fragments (16-byte runs of the channels-last
// image), multiplies by W2 (resident in LDS), applies GELU on the accumulators and packs them to bf16 -- which IS the matching
// k-block of stage 3's B operand (W3's columns are packed in accumulator order).  Stage 3 then streams W3 through a two-slot LDS
// ring in 32-row tiles and writes tokens with the FiLM + positional epilogue.  Against the two GEMM launches this replaces, the
// (n_img, H/4, W/4, C/2) intermediate (write + read) and one launch disappear.
#include "fused_common.hip.h"
#include <stdlib.h>

namespace {

constexpr int EHDR = 4096;   // stream header: bias2 (C/2 floats) then bias3 (C floats)

template <int CB>
struct EncGeom {
  static constexpr int C = 32 * CB, C1 = C / 4, C2 = C / 2;
  static constexpr int K2 = 4 * C1, K3 = 4 * C2;                 // contraction lengths (taps x channels)
  static constexpr int KB2 = K2 / 32, KB3 = K3 / 32;             // k-blocks
  static constexpr int NS2 = C2 / 16, BPT = C1 / 32;             // stage-2 16-row tiles; k-blocks per tap
  static constexpr int CPR2 = K2 / 8, CPR3 = K3 / 8;             // 16-byte chunks per weight row
  static constexpr int W2B = C2 * K2 * 2;                        // W2 image bytes
  static constexpr int T3ROWS = 32, T3B = T3ROWS * K3 * 2, NT3 = C / T3ROWS;   // stage-3 tiles
  static constexpr int LDS = EHDR + W2B + 2 * T3B;
  static_assert(C1 % 32 == 0, "stage-1 channels must fill whole k-blocks");
};

__device__ __forceinline__ void eglds(const char* __restrict__ g, char* l, int bytes, int tid) {   // 8 waves x 1 KiB per pass
  const int wave = tid >> 6, lane = tid & 63;
  for (int off = wave * 1024; off < bytes; off += 8192)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off + lane * 16),
                                     (__attribute__((address_space(3))) void*)(l + off), 16, 0, 0);
}

struct EncArgs {
  const unsigned short* h1;   // (n_img, H1, W1, C1) bf16, H1 = 4 Hp, W1 = 4 Wp
  const char* w;              // [header][W2 image][W3 tiles]
  const float *film_a, *film_b, *s_emb;   // (T, C), (T, C), (Hp*Wp, C)
  float* out;                 // (n_img * Hp * Wp, C) fp32
  int n_img, Hp, Wp, T;
  int out_frames;             // > 0: no FiLM; images are ordered (b, f), f < out_frames, and image (b, f) is written to the frame-major
                              // row block (f * (n_img / out_frames) + b) * Hp * Wp  (the rollout's pre-FiLM frame cache)
};

// SPLIT > 1 (small inputs, e.g. ONE frame per rollout call = 64 token groups for 256 CUs): the per-wave chain stage 2 -> 8 stage-3
// tiles is what the launch waits for, so SPLIT workgroups share a token group -- each repeats stage 2 (half of the work) and takes
// NT3 / SPLIT of the stage-3 tiles: per-wave work 2 -> 1 + 1 / SPLIT at SPLIT x the (idle) CUs.
template <int CB, int SPLIT>
__global__ __launch_bounds__(512, 2) void enc23_kernel(const EncArgs A) {
  using G = EncGeom<CB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [header][W2][slot 0][slot 1]
  char* w2s = smem + EHDR;
  char* slots = smem + EHDR + G::W2B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
  eglds(A.w, smem, EHDR + G::W2B, tid);
  constexpr int NT3L = G::NT3 / SPLIT;                       // stage-3 tiles of this workgroup: t_first .. t_first + NT3L - 1
  const int t_first = (SPLIT > 1) ? (int)blockIdx.y * NT3L : 0;
  const char* w3 = A.w + EHDR + G::W2B + (long)t_first * G::T3B;

  const int HW = A.Hp * A.Wp;
  const long rows = (long)A.n_img * HW;
  const long row = ((long)blockIdx.x * 8 + wave) * 16 + l15;
  const bool live = row < rows;
  const long r = live ? row : 0;
  const unsigned r32 = (unsigned)r;      // rows < 2^31 (a row is 1 KiB of output): 32-bit divisions
  const int img = (int)(r32 / (unsigned)HW), hw = (int)(r32 - (unsigned)img * (unsigned)HW), hp = hw / A.Wp, wp = hw - hp * A.Wp;
  const int H1 = 4 * A.Hp, W1 = 4 * A.Wp;
  const unsigned short* base = A.h1 + (((long)img * H1 + 4 * hp) * W1 + 4 * wp) * G::C1;   // the token's 4 x 4 pixel block

  const int row2 = l15 * G::CPR2 * 16, row3 = l15 * G::CPR3 * 16;

  // input patch of position P as B-operand k-blocks: block b = tap (kh, kw) = b / BPT, channels 32 (b % BPT) + 8 kk .. + 7.
  // One register set: block b of the NEXT position is fetched into xin[b] right after this position's last MFMA on it.
  u32x4 xin[G::KB2];
  auto load_block = [&](int P, int b) {
    const int py = P >> 1, px = P & 1, tap = b / G::BPT, kh = tap >> 1, kw = tap & 1;
    const unsigned short* src = base + ((long)(2 * py + kh) * W1 + (2 * px + kw)) * G::C1 + 32 * (b % G::BPT) + 8 * kk;
    return live ? *(const u32x4*)src : u32x4{0u, 0u, 0u, 0u};
  };
#pragma unroll
  for (int b = 0; b < G::KB2; ++b) xin[b] = load_block(0, b);   // in flight together with the W2 image

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  eglds(w3, slots, G::T3B, tid);                    // stage-3 tiles 0 and 1 land while stage 2 computes
  eglds(w3 + G::T3B, slots + G::T3B, G::T3B, tid);

  // ---- stage 2: the four positions P = (py, px); h2[P * NS2/2 + j] = stage-3 k-block (tap P, channels 32 j .. 32 j + 31) --------
  const float* bias2 = (const float*)smem;
  const float* bias3 = (const float*)smem + G::C2;
  u32x4 h2[G::KB3];
  unsigned bt2[G::KB2];
#pragma unroll
  for (int b = 0; b < G::KB2; ++b) bt2[b] = lds_addr(w2s + row2 + (swz_chunk(l15, b * 4 + kk, G::CPR2) << 4));
  static_for<4>([&](auto pc) {
    constexpr int P = decltype(pc)::value;
    static_for<2>([&](auto hc) {   // two halves of the C/2 outputs: 16 accumulator registers live instead of 32
      constexpr int hh = decltype(hc)::value, NH2 = G::NS2 / 2;
      f32x4 acc[NH2];
#pragma unroll
      for (int ns = 0; ns < NH2; ++ns) acc[ns] = *(const f32x4*)(bias2 + (hh * NH2 + ns) * 16 + kk * 4);
      mfma_stream<NH2 * G::KB2, 4>(
          [&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<(hh * NH2 + i % NH2) * 16 * G::CPR2 * 16>{bt2[i / NH2]}; },
          [&](auto ic, const u32x4& wf) {
            constexpr int i = decltype(ic)::value, ns = i % NH2, kb = i / NH2;
            acc[ns] = mfma_bf16(wf, xin[kb], acc[ns]);
            if constexpr (hh == 1 && ns == NH2 - 1 && P < 3) xin[kb] = load_block(P + 1, kb);   // last use of block kb at this position
          });
#pragma unroll
      for (int j = 0; j < NH2 / 2; ++j) {
        u32x4 f = pack8(gelu_poly4<false>(acc[2 * j]), gelu_poly4<false>(acc[2 * j + 1]));
        asm volatile("" : "+v"(f));   // materialise the packed fragment HERE: otherwise the compiler keeps the fp32 halves (spilled) until stage 3
        h2[P * (G::NS2 / 2) + hh * (NH2 / 2) + j] = f;
      }
    });
  });

  // ---- stage 3: 32 output features per W3 tile; FiLM + positional epilogue ------------------------------------------------------
  const int t_idx = img % A.T;
  const float* fa = A.film_a + (long)t_idx * G::C;
  const float* fb = A.film_b + (long)t_idx * G::C;
  const float* se = A.s_emb + (long)hw * G::C;
  const int nb = A.out_frames > 0 ? A.n_img / A.out_frames : 1;
  float* orow = A.out + (A.out_frames > 0 ? ((long)(img % A.out_frames) * nb + img / A.out_frames) * HW + hw : r) * G::C;
  unsigned bt3[G::KB3];   // slot-0 bases; the slot and row-tile offsets are instruction immediates
#pragma unroll
  for (int b = 0; b < G::KB3; ++b) bt3[b] = lds_addr(slots + row3 + (swz_chunk(l15, b * 4 + kk, G::CPR3) << 4));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  static_for<NT3L>([&](auto tc) {
    constexpr int t = decltype(tc)::value;             // local tile index (ring slot, prefetch distance); tg = its global index
    const int tg = t_first + t;
    f32x4 acc[2];
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) acc[ns] = *(const f32x4*)(bias3 + tg * 32 + ns * 16 + kk * 4);
    mfma_stream<2 * G::KB3, 4>(
        [&](auto ic) { constexpr int i = decltype(ic)::value; return LdsAddr<(t & 1) * G::T3B + (i % 2) * 16 * G::CPR3 * 16>{bt3[i / 2]}; },
        [&](auto ic, const u32x4& wf) {
          constexpr int i = decltype(ic)::value, ns = i % 2, kb = i / 2;
          acc[ns] = mfma_bf16(wf, h2[kb], acc[ns]);
        });
    __syncthreads();                                   // every wave is done with slot t & 1
    if constexpr (t + 2 < NT3L) eglds(w3 + (long)(t + 2) * G::T3B, slots + (t & 1) * G::T3B, G::T3B, tid);
    if constexpr (t + 1 < NT3L) {
      // tile t + 1 was issued one tile ago: all but the pieces just issued for t + 2 must have landed.  (vmcnt counts loads, stores
      // and LDS-DMA together in issue order; the epilogue below comes AFTER this wait so its stores never sit in front of it.)
      if constexpr (t + 2 < NT3L) wait_vmcnt<G::T3B / 8192>();
      else wait_vmcnt<0>();
      __syncthreads();
    }
    if (live) {
#pragma unroll
      for (int ns = 0; ns < 2; ++ns) {
        const int n0 = tg * 32 + ns * 16 + kk * 4;
        const f32x4 a = *(const f32x4*)(fa + n0), b = *(const f32x4*)(fb + n0), sv = *(const f32x4*)(se + n0);
        *(f32x4*)(orow + n0) = A.out_frames > 0 ? acc[ns] : acc[ns] * a + b + sv;
      }
    }
  });
}

// position p inside a 32-block of a k-permuted row holds source channel c:  p = 8*kk + 4*dt + r  ->  c = 16*dt + 4*kk + r
__device__ __forceinline__ int ekperm(int p) {
  const int q = p & 31, kq = q >> 3, dt = (q >> 2) & 1, r = q & 3;
  return (p & ~31) + dt * 16 + kq * 4 + r;
}

// conv2 (C2, C1, 2, 2) -> W2 image rows n, k = (kh, kw, ci);  conv3 (C, C2, 2, 2) -> tiles of 32 rows, k = (tap, c2 permuted in 32-blocks)
__global__ void pack_enc_stream_kernel(const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ w3,
                                       const float* __restrict__ b3, int C, char* __restrict__ dst) {
  const int C1 = C / 4, C2 = C / 2, K2 = 4 * C1, K3 = 4 * C2, cpr2 = K2 / 8, cpr3 = K3 / 8;
  const long W2B = (long)C2 * K2 * 2;
  const int part = blockIdx.y;
  if (part == 0) {          // header
    float* h = (float*)dst;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < EHDR / 4; i += gridDim.x * blockDim.x)
      h[i] = i < C2 ? b2[i] : (i < C2 + C ? b3[i - C2] : 0.0f);
  } else if (part == 1) {   // W2
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)C2 * cpr2; idx += (long)gridDim.x * blockDim.x) {
      const int n = (int)(idx / cpr2), c = (int)(idx % cpr2);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = c * 8 + e, tap = k / C1, ci = k % C1;
        v[e] = w2[(((long)n * C1 + ci) * 2 + (tap >> 1)) * 2 + (tap & 1)];
      }
      u32x4 o;
      o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
      *((u32x4*)(dst + EHDR) + (long)n * cpr2 + swz_chunk(n, c, cpr2)) = o;
    }
  } else {                  // W3 tiles
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)C * cpr3; idx += (long)gridDim.x * blockDim.x) {
      const int n = (int)(idx / cpr3), c = (int)(idx % cpr3);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int p = c * 8 + e, tap = p / C2, c2 = ekperm(p % C2);
        v[e] = w3[(((long)n * C2 + c2) * 2 + (tap >> 1)) * 2 + (tap & 1)];
      }
      u32x4 o;
      o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
      const int tile = n / 32, rr = n % 32;
      *((u32x4*)(dst + EHDR + W2B + (long)tile * 32 * K3 * 2) + (long)rr * cpr3 + swz_chunk(rr, c, cpr3)) = o;
    }
  }
}

template <int CB, int SPLIT>
void launch_enc23_s(const EncArgs& A, long rows, hipStream_t s) {
  using G = EncGeom<CB>;
  static TantePerDevice attr;
  attr.once([&] {
    hipFuncSetAttribute((const void*)enc23_kernel<CB, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
  });
  hipLaunchKernelGGL((enc23_kernel<CB, SPLIT>), dim3((unsigned)((rows + 127) / 128), SPLIT), dim3(512), G::LDS, s, A);
}
template <int CB>
void launch_enc23(const EncArgs& A, hipStream_t s) {
  const long rows = (long)A.n_img * A.Hp * A.Wp;
  const int force = tante_opt("TANTE_ENC23_SPLIT", 0);
  const int groups = (int)((rows + 127) / 128);
  const int split = force ? force : (groups <= 64 ? 4 : groups <= 128 ? 2 : 1);      // fill the CUs when the input is one frame
  if (split >= 4) launch_enc23_s<CB, 4>(A, rows, s);
  else if (split == 2) launch_enc23_s<CB, 2>(A, rows, s);
  else launch_enc23_s<CB, 1>(A, rows, s);
}

}  // namespace

extern "C" int tante_enc23_supported(int C) { return C == 256; }

extern "C" int64_t tante_enc23_stream_bytes(int C) {
  const long C1 = C / 4, C2 = C / 2;
  return EHDR + C2 * 4 * C1 * 2 + (long)C * 4 * C2 * 2;
}

extern "C" int tante_pack_enc23(const float* w2, const float* b2, const float* w3, const float* b3, int C, void* enc_stream, void* stream) {
  if (!w2 || !b2 || !w3 || !b3 || !enc_stream) TANTE_FAIL(-1, "tante_pack_enc23: null pointer");
  if (!tante_enc23_supported(C)) TANTE_FAIL(-2, "tante_pack_enc23: unsupported C=%d", C);
  hipLaunchKernelGGL(pack_enc_stream_kernel, dim3(64, 3), dim3(256), 0, (hipStream_t)stream, w2, b2, w3, b3, C, (char*)enc_stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_enc23_fused(const void* h1, int n_img, int Hp, int Wp, int C, const void* enc_stream, const float* film_a,
                                 const float* film_b, const float* s_emb, int T, float* out, void* stream) {
  if (!h1 || !enc_stream || !film_a || !film_b || !s_emb || !out) TANTE_FAIL(-1, "tante_enc23_fused: null pointer");
  if (!tante_enc23_supported(C)) TANTE_FAIL(-2, "tante_enc23_fused: unsupported C=%d", C);
  if (n_img <= 0 || Hp <= 0 || Wp <= 0 || T <= 0) TANTE_FAIL(-1, "tante_enc23_fused: bad shape");
  if (((uintptr_t)h1 % 16) || ((uintptr_t)out % 16) || ((uintptr_t)film_a % 16) || ((uintptr_t)film_b % 16) || ((uintptr_t)s_emb % 16))
    TANTE_FAIL(-1, "tante_enc23_fused: alignment");
  EncArgs A;
  A.h1 = (const unsigned short*)h1; A.w = (const char*)enc_stream; A.film_a = film_a; A.film_b = film_b; A.s_emb = s_emb; A.out = out;
  A.n_img = n_img; A.Hp = Hp; A.Wp = Wp; A.T = T; A.out_frames = 0;
  launch_enc23<8>(A, (hipStream_t)stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_enc23_frames(const void* h1, int n_img, int frames, int Hp, int Wp, int C, const void* enc_stream, float* out, void* stream) {
  if (!h1 || !enc_stream || !out) TANTE_FAIL(-1, "tante_enc23_frames: null pointer");
  if (!tante_enc23_supported(C)) TANTE_FAIL(-2, "tante_enc23_frames: unsupported C=%d", C);
  if (n_img <= 0 || frames <= 0 || n_img % frames || Hp <= 0 || Wp <= 0) TANTE_FAIL(-1, "tante_enc23_frames: bad shape");
  if (((uintptr_t)h1 % 16) || ((uintptr_t)out % 16)) TANTE_FAIL(-1, "tante_enc23_frames: alignment");
  EncArgs A;
  A.h1 = (const unsigned short*)h1; A.w = (const char*)enc_stream; A.film_a = A.film_b = A.s_emb = out; A.out = out;   // tables unused
  A.n_img = n_img; A.Hp = Hp; A.Wp = Wp; A.T = 1; A.out_frames = frames;
  launch_enc23<8>(A, (hipStream_t)stream);
  TANTE_CHECK_LAUNCH();
  return 0;
}
