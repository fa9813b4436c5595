// Shared device helpers for the TANTE gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>

#include "../../include/tante_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define TANTE_WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------------
void tante_set_error(const char* fmt, ...);
#define TANTE_FAIL(code, ...)      \
  do {                             \
    tante_set_error(__VA_ARGS__);  \
    return (code);                 \
  } while (0)
#define TANTE_CHECK_LAUNCH()                                                        \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) TANTE_FAIL(-3, "%s: %s", __func__, hipGetErrorString(e__)); \
  } while (0)

// ---- once-per-device launch attributes --------------------------------------------------------------
// hipFuncSetAttribute acts on the CURRENT device, so "already set" is remembered per device (a process that drives several GPUs
// would otherwise launch with the default 64 KiB LDS limit on every device but the first).  Setting twice is harmless, so two
// threads racing here only repeat the call.
struct TantePerDevice {
  std::atomic<unsigned long long> done{0};
  template <class F>
  void once(F&& f) {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
      f();
      done.fetch_or(bit, std::memory_order_release);
    }
  }
};

// ---- zero fill ------------------------------------------------------------------------------------
// Accumulators that kernels add into are cleared by a KERNEL on the caller's stream, not by hipMemsetAsync: inside a captured HIP graph
// (train.GraphedTrainStep) the memset becomes a memset node, and replays of such graphs were seen to run with the buffers NOT cleared
// once other work (allocations, host copies) had happened between replays -- stale sums from the previous replay in the loss and in the
// FiLM / colsum / weight gradients.  A kernel node is ordered like every other launch.  bytes must be a multiple of 4.
static __global__ void tante_zero_kernel(unsigned* __restrict__ p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
static __global__ void tante_zero16_kernel(uint4* __restrict__ p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = uint4{0u, 0u, 0u, 0u};
}
static inline hipError_t tante_zero_async(void* p, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  if (bytes % 4) return hipErrorInvalidValue;
  if (bytes % 16 == 0 && ((uintptr_t)p % 16) == 0) {
    const long n = (long)(bytes / 16);
    hipLaunchKernelGGL(tante_zero16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint4*)p, n);
  } else {
    const long n = (long)(bytes / 4);
    hipLaunchKernelGGL(tante_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (unsigned*)p, n);
  }
  return hipGetLastError();
}

// ---- workspace of the propagator weight-gradient kernels (tante_axis_wgrad_ws, tante_axis_mlp_bwd_fused_ws) -----------------------------
// partials of up to TANTE_AW_MAXWG workgroups, TANTE_AW_SLAB floats each, then one arrival counter per group of TANTE_AW_GS workgroups
constexpr int TANTE_AW_GS = 16;
constexpr int TANTE_AW_MAXWG = 1024;
constexpr int TANTE_AW_SLAB = 64 * 64 + 64;
constexpr long TANTE_AW_WS_FLOATS = (long)TANTE_AW_MAXWG * TANTE_AW_SLAB + TANTE_AW_MAXWG / TANTE_AW_GS;

// ---- tuning options ------------------------------------------------------------------------------
// The product library never reads the environment.  Its launch heuristics (workgroup counts, kernel-variant choices: every one an
// A/B switch between forms that compute the same function) look their overrides up in ONE process-wide table that only
// tante_set_option() writes (include/tante_hip.h; defined in pointwise.hip).  The Python binding forwards TANTE_* environment
// variables into it at load time (tante_amd/_lib.py), which is what the measurement scripts under tools/ use.
int tante_opt(const char* name, int dflt);

// ---- timing-ablation switches ------------------------------------------------------------------
// TANTE_*_DEBUG skip parts of a kernel to time the rest; the results are WRONG by construction, so the shipped library
// never reads them: they exist only in a -DTANTE_ABLATE build (tools/ab_lib.sh builds one beside the product library).
static inline int tante_ablate_env(const char* name) {
#ifdef TANTE_ABLATE
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
#else
  (void)name;
  return 0;
#endif
}

// ---- bf16 pack / unpack (round-to-nearest-even via v_cvt_pk_bf16_f32, NaN-preserving) ---------
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  bf16x2 p;
  p[0] = (__bf16)lo;
  p[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// ---- write-through stores -----------------------------------------------------------------------------------------------------
// A kernel's ordinary stores sit dirty in its XCD's L2 until the kernel-end release writes them back: B bytes cost ~B / 6 TB/s behind ALL
// of the kernel's compute (MI355X_MICROARCH.md, kernel boundary; measured here: 33.5 MB of block-kernel output = 5.6 us, its timing build
// without the store).  A store with sc0 sc1 goes to memory as it is issued, so results that are final when they are written -- a kernel's
// output rows, the saved tensors of a training forward -- stream out under the rest of the kernel and the release finds nothing to do.
// Only for data no other workgroup of the SAME launch reads back.  (-DTANTE_PLAIN_STORES: ordinary stores, for A/B.)
__device__ __forceinline__ void st_wt16(void* p, const u32x4& v) {
#ifdef TANTE_PLAIN_STORES
  *(u32x4*)p = v;
#else
  // (s_nop 1: a VALU write to the data registers of a store wider than 64 bits needs wait states behind it; the compiler's hazard
  // recognizer covers its own stores, not the inside of an asm statement -- without it a GELU that reused these registers right behind
  // the store corrupted the saved pre-activations of the training forward)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ void st_wt16(void* p, const f32x4& v) { st_wt16(p, __builtin_bit_cast(u32x4, v)); }
__device__ __forceinline__ void st_wt8(void* p, const u32x2& v) {
#ifdef TANTE_PLAIN_STORES
  *(u32x2*)p = v;
#else
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#endif
}

// ---- activations (always evaluated in fp32) ---------------------------------------------------
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): one v_exp, one v_rcp, 6 fma -- a third of ocml erff.
// Used by the bf16 path only; the fp32 parity path keeps erff.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}
// GELU of the bf16 compute path:  x * sat(0.5 + x * Q(min(x^2, X0^2)))  with Q a degree-7 minimax fit of (Phi(x) - 1/2) / x on
// [0, X0] (fitted in float64, checked in float32 over [-10, 10]: |error| <= 8.3e-5 for the erf form, 7.0e-5 for the tanh form --
// a twentieth of a bf16 ulp at |x| ~ 1, where the result is rounded to bf16 next anyway).  No transcendental and no branch: on
// f32x4 it compiles to v_pk_fma_f32 / v_pk_mul_f32, 7 VALU instructions per element against ~16 + v_exp + v_rcp (both
// quarter-rate) before -- the GELUs were a third of the fused kernels' VALU time.  The fp32 parity path keeps erff / tanhf.
template <bool TANH, class V>
__device__ __forceinline__ V gelu_poly(V x, V (*splat)(float)) {
  V s = __builtin_elementwise_min(x * x, splat(TANH ? 16.0f : 18.0625f));
  V q = splat(TANH ? -1.464602106e-09f : -8.949876396e-10f);
  q = q * s + splat(TANH ? 1.154394198e-07f : 7.897345000e-08f);
  q = q * s + splat(TANH ? -3.974079846e-06f : -3.026334298e-06f);
  q = q * s + splat(TANH ? 7.951747102e-05f : 6.673120515e-05f);
  q = q * s + splat(TANH ? -1.043852768e-03f : -9.494310943e-04f);
  q = q * s + splat(TANH ? 9.655649774e-03f : 9.293001145e-03f);
  q = q * s + splat(TANH ? -6.609884650e-02f : -6.551689655e-02f);
  q = q * s + splat(TANH ? 3.986152411e-01f : 3.984565735e-01f);
  V phi = x * q + splat(0.5f);
  phi = __builtin_elementwise_min(__builtin_elementwise_max(phi, splat(0.0f)), splat(1.0f));
  return x * phi;
}
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ float splat1(float v) { return v; }
template <bool TANH>
__device__ __forceinline__ f32x4 gelu_poly4(const f32x4& x) { return gelu_poly<TANH, f32x4>(x, splat4); }
template <bool TANH>
__device__ __forceinline__ float gelu_poly1(float x) { return gelu_poly<TANH, float>(x, splat1); }

__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float c = 0.79788456080286535588f;  // sqrt(2/pi)
  return 0.5f * x * (1.0f + tanhf(c * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case TANTE_ACT_GELU_ERF: return gelu_erf_f(x);
    case TANTE_ACT_GELU_TANH: return gelu_tanh_f(x);
    case TANTE_ACT_RELU: return fmaxf(x, 0.0f);
    default: return x;
  }
}

// ---- LDS weight-tile swizzle -------------------------------------------------------------------
// A packed weight tile is [nt rows][cpr 16-byte chunks].  MFMA operand reads are ds_read_b128 with
// lane l -> (row l&15, chunk 4*cb + (l>>4)).  XOR-ing the chunk index with the row makes every
// 16-lane ds_read_b128 service group hit 16 distinct 16-byte slots of the 256-byte bank row
// (cdna_hip_programming.md 5.5 T2; group lists in MI355X_MICROARCH.md "LDS").
__host__ __device__ __forceinline__ int swz_chunk(int r, int c, int cpr) {
  return (cpr >= 16) ? (c ^ (r & 15)) : (c ^ ((r >> 1) & 7));  // cpr == 8 -> two rows per bank row
}

// d/dx GELU_erf(x) = Phi(x) + x phi(x) for the bf16 compute path: Phi from the forward's polynomial (|error| <= 8.3e-5), phi from one exp2
__device__ __forceinline__ float gelu_erf_grad_fast(float x) {
  const float s = fminf(x * x, 18.0625f);
  float q = -8.949876396e-10f;
  q = q * s + 7.897345000e-08f;
  q = q * s + -3.026334298e-06f;
  q = q * s + 6.673120515e-05f;
  q = q * s + -9.494310943e-04f;
  q = q * s + 9.293001145e-03f;
  q = q * s + -6.551689655e-02f;
  q = q * s + 3.984565735e-01f;
  const float Phi = fminf(fmaxf(fmaf(x, q, 0.5f), 0.0f), 1.0f);
  return fmaf(x * 0.3989422804f, __builtin_amdgcn_exp2f(-0.72134752f * x * x), Phi);
}

// derivative of an activation at its pre-activation value (training: act_bwd kernel and the data-gradient GEMM's epilogue)
__device__ __forceinline__ float act_df(float x, int act) {
  switch (act) {
    case TANTE_ACT_GELU_ERF: {  // Phi(x) + x phi(x)
      const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
      return cdf + x * 0.39894228040143267794f * expf(-0.5f * x * x);
    }
    case TANTE_ACT_GELU_TANH: {
      const float c = 0.79788456080286535588f, u = c * (x + 0.044715f * x * x * x), t = tanhf(u);
      return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * c * (1.0f + 3.0f * 0.044715f * x * x);
    }
    case TANTE_ACT_RELU: return x > 0.f ? 1.0f : 0.f;
    default: return 1.0f;
  }
}

// ---- per-step seed word in device memory (tante_set_seed_mix) ---------------------------------------------------------------------
// The dropout seeds are kernel arguments by value: a HIP graph of a train step replays them unchanged.  The kernels that draw masks
// therefore XOR every seed with a 64-bit word they READ from device memory when one is registered for the device: the host rewrites
// the word before each replay (an 8-byte copy) and the captured step draws fresh masks.  Null (the default): seeds are used as given.
const unsigned long long* tante_seed_mix_ptr();     // the current device's registered word (host side), defined in train.hip

// ---- counter-based dropout mask: keep(seed, idx) is a pure function, so backward regenerates the forward's mask -------
// One 32-bit hash serves TWO consecutive elements (idx >> 1 is hashed; the even element takes the low 16 bits, the odd one the high
// 16): the kernels work on runs of 2 or 4 consecutive elements and hash half / a quarter as often as they decide.  The hash is a Weyl
// step of the pair index, keyed by both words of the seed, through murmur3's 32-bit finaliser: ~11 VALU instructions per pair.  (Round 1
// used a splitmix64 finaliser per pair, three 64-bit multiplies = ~40 instructions; with one call per element of the attention
// probabilities that was 10 - 14 us of the L = 48 attention forward / backward launches.)  16 uniform bits per decision: the keep
// probability is 1 - floor(65536 p) / 65536.
__device__ __forceinline__ unsigned dropout_mix32(unsigned h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  return h ^ (h >> 16);
}
__device__ __forceinline__ unsigned dropout_hash(unsigned long long seed, unsigned long long pair) {
  unsigned a = (unsigned)pair * 0x9E3779B1u + (unsigned)seed;
  a ^= ((unsigned)(pair >> 32) + (unsigned)(seed >> 32)) * 0x7FEB352Du;     // the seed's high word always takes part
  return dropout_mix32(a);
}
__device__ __forceinline__ unsigned dropout_threshold(float p) { return (unsigned)(p * 65536.0f); }
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned long long idx, float p) {
  const unsigned h = dropout_hash(seed, idx >> 1);
  return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= dropout_threshold(p);
}
// the two decisions of the pair that starts at the EVEN index idx0 (bit 0: idx0, bit 1: idx0 + 1)
__device__ __forceinline__ unsigned dropout_keep2(unsigned long long seed, unsigned long long idx0, float p) {
  const unsigned h = dropout_hash(seed, idx0 >> 1), thr = dropout_threshold(p);
  return ((h & 0xffffu) >= thr ? 1u : 0u) | ((h >> 16) >= thr ? 2u : 0u);
}
// the four decisions of the run that starts at idx0, a multiple of 4 (bits 0..3)
__device__ __forceinline__ unsigned dropout_keep4(unsigned long long seed, unsigned long long idx0, float p) {
  return dropout_keep2(seed, idx0, p) | (dropout_keep2(seed, idx0 + 2, p) << 2);
}
