// Device helpers shared by the feature-sliced block kernels (block_sliced.hip: forward, inference and training; block_bwd.hip: the
// backward of the block's MLP half): the weight-fragment stream and its register ring, the slice GEMM over an LDS image, cross-lane
// reductions without LDS traffic.
#pragma once
#include "common.hip.h"
#include "fused_common.hip.h"

namespace {

constexpr int FS_C = 256;
constexpr int FS_FRAG = 1024;                              // one wave-wide MFMA operand fragment: 64 lanes x 16 B
constexpr int FS_W_BYTES = 6 * FS_C * FS_C * 2;               // 768 KiB: q, k, v, Wo, W1, W2 as bf16 fragments
constexpr int FS_BIAS_FLOATS = 7 * FS_C;                   // by output feature: q | out' (folded) | fc1 | fc2 | k | v | out (plain)
constexpr int FS_ROW = 512;                                // bytes per token row of an LDS image

__device__ __forceinline__ u32x4 ldg_frag(const char* p) { return *(const u32x4*)p; }
// The weight stream through a buffer resource: address = base (4 SGPRs) + lane offset (ONE 32-bit VGPR for the whole kernel) + fragment
// offset (an SGPR the scalar unit sets).  As plain global loads every fragment needed its own 64-bit vector address -- v_lshl_add_u64 /
// v_add_co + v_addc per load, ~250 of the block kernel's 2 600 vector instructions per wave, in a kernel bound by vector issue.
// (-DFS_NO_BUFW: the plain loads, for A/B.)  The stream is < 2 GiB; num_records is left at the maximum (nothing reads past the stream).
struct FsW {
#ifdef FS_NO_BUFW
  const char* p;
#else
  __amdgpu_buffer_rsrc_t r;
  unsigned voff;
#endif
};
__device__ __forceinline__ FsW fs_wstream(const char* base, unsigned lane_off) {
#ifdef FS_NO_BUFW
  return FsW{base + lane_off};
#else
  return FsW{__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000), lane_off};
#endif
}
__device__ __forceinline__ u32x4 ldg_frag(const FsW& w, int byte_off) {
#ifdef FS_NO_BUFW
  return *(const u32x4*)(w.p + byte_off);
#else
  return __builtin_amdgcn_raw_buffer_load_b128(w.r, (int)w.voff, byte_off, 0);
#endif
}
__device__ __forceinline__ u32x4 ldg_frag(const char* p, int byte_off) { return *(const u32x4*)(p + byte_off); }
// TIMING EXPERIMENT ONLY (-DFS_EXP_ROT, wrong results): every workgroup reads the k-steps of a matrix in an order rotated by a per-workgroup
// amount (fs_exp_rot, wave-uniform), the MFMAs consume them as if nothing had happened.  Prices the hot-spot hypothesis: all 512
// workgroups ask the L2 for the SAME 16 KiB of weight fragments at the same moment (the q and k GEMMs, right behind LayerNorm1, take
// 6 k cycles in the stamps where the v / fc1 / fc2 GEMMs take 3 k).
#ifdef FS_EXP_ROT
__device__ int fs_exp_rot_of() { return (int)((blockIdx.x >> 3) & 7); }
#define FS_GROT(g) (8 * ((g) >> 3) + ((((g) & 7) + fs_exp_rot_of()) & 7))
#else
#define FS_GROT(g) (g)
#endif
// timing experiment only (-DFS_EXP_NO_WLOAD, wrong results): every fragment load of the weight stream re-reads the stream's FIRST k-step
// (16 KiB per workgroup, L1-resident) -- the same instructions and waits, none of the L2 -> L1 traffic.  Prices the weight stream.
#ifdef FS_EXP_NO_WLOAD
#define FS_WOFF(f) ((f) % 16)
#else
#define FS_WOFF(f) (f)
#endif

// ---- cross-lane reductions without LDS traffic -------------------------------------------------------------------------------------
// over the 16 lanes of a row (lanes that share lane >> 4): DPP quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
template <int CTRL>
__device__ __forceinline__ float fs_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += fs_dpp<0xB1>(v);
  v += fs_dpp<0x4E>(v);
  v += fs_dpp<0x141>(v);
  return v + fs_dpp<0x140>(v);
}
// over the four rows of a wave (lanes l15 + 16 kk, i.e. the accumulator rows of one column): v_permlane16_swap exchanges the odd rows of
// its first operand with the even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the second;
// fed the same value twice they return the xor-16 / xor-32 partners side by side
template <class Op>
__device__ __forceinline__ float rows_reduce(float v, Op op) {
#ifdef FS_SHFL
  v = op(v, __shfl_xor(v, 16));
  return op(v, __shfl_xor(v, 32));
#else
  const unsigned u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  v = op(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const unsigned w = __float_as_uint(v);
  auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return op(__uint_as_float(b[0]), __uint_as_float(b[1]));
#endif
}
__device__ __forceinline__ float rows_max(float v) { return rows_reduce(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ float rows_sum(float v) { return rows_reduce(v, [](float a, float b) { return a + b; }); }

// acc[j][tt] (+)= W[16-row tile j of the wave's 16 RT-row slice][all 256 k] . image[token tile tt], k-step outermost.
//
// Weights: the six matrices of the block (q, k, v, Wo, W1, W2) are ONE stream of 48 k-steps per wave (fragment (g, j) at
// wq + (16 g + j) KiB, g = 8 m + ks), and the register ring `wb` that prefetches it PF k-steps ahead lives for the whole kernel: while
// a GEMM runs its last k-steps the first fragments of the NEXT matrix are already on their way, across the barriers and the
// attention / LayerNorm / GELU phases in between, so no GEMM starts by waiting for L2.  Image fragments come through the explicit
// LDS read ring of mfma_stream (counted lgkmcnt, reads RING fragments ahead of their MFMAs), each feeding all RT row tiles.  All
// RT x NTT accumulators are live, a matrix's weights never are: that keeps the kernel inside 256 registers at two waves per SIMD.
// The inline-asm reads and waits carry memory clobbers, so the compiler cannot sink the prefetch loads towards their use (left alone
// it loads every fragment right in front of its first MFMA and waits for it there).
// SWAP = false: D[feature][token] = mfma(W, img);  SWAP = true: D[token][feature] = mfma(img, W)  (the V projection).
constexpr int FS_KSTEPS = 48;
// first PF k-steps of matrix M into the ring (the stream is primed twice per launch: before LayerNorm1 for q | k | v, after the
// attention for Wo | W1 | W2; in between the GEMMs keep it running across their own boundaries, GEND = where the run ends)
template <int M, int RT, int PF, class WP>
__device__ __forceinline__ void fs_wring_prime(const WP& wq, u32x4 (&wb)[PF + 1][RT]) {
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int j = 0; j < RT; ++j) wb[(8 * M + p) % (PF + 1)][j] = ldg_frag(wq, FS_WOFF(16 * FS_GROT(8 * M + p) + j) * FS_FRAG);
}
// HOOK (optional): called once per k-step with std::integral_constant<int, ks>, in front of the k-step's MFMAs -- work that should ride
// the matrix phase instead of forming a burst of its own behind it (block_bwd_fs.hip spreads its row stores over the GEMMs this way)
struct FsNoHook {
  template <class KC>
  __device__ __forceinline__ void operator()(KC) const {}
};
template <int M, int GEND, int NTT, int RT, bool SWAP, int PF, class WP, class HOOK = FsNoHook>
__device__ __forceinline__ void fs_slice_gemm(const WP& wq, u32x4 (&wb)[PF + 1][RT], const char* img, const int (&rdo)[4],
                                              f32x4 (&acc)[RT][NTT], HOOK&& hook = FsNoHook{}) {
  unsigned ab[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ab[j] = lds_addr(img + rdo[j]);
#ifdef FS_EXP_MFMA32
  u32x4 fs_prev_tf = u32x4{0u, 0u, 0u, 0u};
#endif
  mfma_stream<8 * NTT, 4>(
      [&](auto ic) {
        constexpr int i = decltype(ic)::value, ks = i / NTT, tt = i % NTT;
        return LdsAddr<tt * 8192 + (ks >> 2) * 256>{ab[ks & 3]};
      },
      [&](auto ic, const u32x4& tf) {
        constexpr int i = decltype(ic)::value, ks = i / NTT, tt = i % NTT, g = 8 * M + ks;
        if constexpr (tt == 0 && g + PF < GEND) {
#pragma unroll
          for (int j = 0; j < RT; ++j) wb[(g + PF) % (PF + 1)][j] = ldg_frag(wq, FS_WOFF(16 * FS_GROT(g + PF) + j) * FS_FRAG);
        }
        if constexpr (tt == 0) hook(std::integral_constant<int, ks>{});
#ifdef FS_EXP_MFMA32
        // TIMING EXPERIMENT ONLY (wrong results): the same loads, the same operand and accumulator registers, half as many MFMAs of
        // twice the size -- v_mfma_f32_32x32x16_bf16 on the 2 x 2 groups of this wave's 16 x 16 tiles.  Prices the instruction shape
        // (8 of 32 issue cycles held instead of 8 of 16) before anyone rewrites the accumulator layout of every epilogue for it.
        if constexpr (RT % 2 == 0 && NTT % 2 == 0) {
          if constexpr (tt % 2 == 1) {
#pragma unroll
            for (int j = 0; j < RT; j += 2) {
              f32x16 big = __builtin_shufflevector(__builtin_shufflevector(acc[j][tt - 1], acc[j][tt], 0, 1, 2, 3, 4, 5, 6, 7),
                                                   __builtin_shufflevector(acc[j + 1][tt - 1], acc[j + 1][tt], 0, 1, 2, 3, 4, 5, 6, 7),
                                                   0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
              big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wb[g % (PF + 1)][j]), __builtin_bit_cast(bf16x8, tf), big, 0, 0, 0);
              big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wb[g % (PF + 1)][j + 1]), __builtin_bit_cast(bf16x8, fs_prev_tf), big, 0, 0, 0);
              acc[j][tt - 1] = __builtin_shufflevector(big, big, 0, 1, 2, 3);
              acc[j][tt] = __builtin_shufflevector(big, big, 4, 5, 6, 7);
              acc[j + 1][tt - 1] = __builtin_shufflevector(big, big, 8, 9, 10, 11);
              acc[j + 1][tt] = __builtin_shufflevector(big, big, 12, 13, 14, 15);
            }
          } else {
            fs_prev_tf = tf;
          }
          return;
        }
#endif
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          if constexpr (SWAP) acc[j][tt] = mfma_bf16(tf, wb[g % (PF + 1)][j], acc[j][tt]);
          else acc[j][tt] = mfma_bf16(wb[g % (PF + 1)][j], tf, acc[j][tt]);
        }
      });
}

}  // namespace
