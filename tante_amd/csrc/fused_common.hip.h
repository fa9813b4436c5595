// Helpers shared by the register-chained fused kernels (block_fused.hip, enc_fused.hip): static loops, the explicit LDS read
// ring that feeds MFMA A operands, bf16 fragment packing.
#pragma once
#include "common.hip.h"
#include <utility>

namespace {

// compile-time loop: the tile index must be a constant so that every register array is statically indexed
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {  // all but the N youngest vector-memory operations have completed
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// N weight-fragment MFMAs with a D-deep register ring of ds_read_b128s kept in flight.  Left to itself the compiler emits
// "ds_read_b128; s_waitcnt lgkmcnt(0); v_mfma" per fragment -- one full LDS round trip (~120 cycles) per 16-cycle MFMA -- and
// even with the reads hoisted in the source its waits stay lgkmcnt(0) for as long as an LDS-DMA is in flight (a FLAT-encoded
// global_load_lds makes it flush both counters at every dependency).  So the ring is explicit: the reads are inline asm, fragment
// i + D is issued before MFMA i runs, and the wait in front of MFMA i is a counted lgkmcnt(number of younger reads); LDS returns
// in order, and an extra compiler-issued LDS read in between only makes the count conservative.  The stream starts from
// lgkmcnt(0) so that no scalar load (out-of-order in the same counter) is pending while counts are relied on.
// addr(ic) -> LDS pointer of fragment ic, use(ic, frag) issues the MFMA; ic is an integral_constant.
template <int OFF>
struct LdsAddr {   // LDS byte address = base (a VGPR) + OFF (an instruction immediate, < 64 KiB): no VALU add per read
  unsigned base;
  static constexpr int off = OFF;
};
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128(LdsAddr<OFF> a) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is a 16-bit immediate");
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a.base), "n"(OFF) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4& frag) {  // all but the N youngest LDS operations have returned; ties `frag` to the wait
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait2(u32x4& f0, u32x4& f1) {  // one wait for two fragments
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(N));
}
template <int N, int D, class AddrF, class UseF>
__device__ __forceinline__ void mfma_stream(AddrF&& addr, UseF&& use) {
  static_assert(D <= 15 && D % 2 == 0, "lgkmcnt is a 4-bit field; fragments are consumed in pairs");
  u32x4 ring[D];
#ifdef TANTE_MFMA_SETPRIO
  __builtin_amdgcn_s_setprio(1);
#endif
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  static_for<(D < N ? D : N)>([&](auto ic) { ring[decltype(ic)::value] = lds_read128(addr(ic)); });
  // fragments are consumed in pairs behind ONE counted wait: the kernels that use this are bound by instruction issue, and an
  // s_waitcnt per MFMA is an issue slot per MFMA
  static_for<N / 2>([&](auto jc) {
    constexpr int i = 2 * decltype(jc)::value;
    u32x4 c0 = ring[i % D], c1 = ring[(i + 1) % D];
    if constexpr (i + D < N) ring[i % D] = lds_read128(addr(std::integral_constant<int, i + D>{}));
    if constexpr (i + 1 + D < N) ring[(i + 1) % D] = lds_read128(addr(std::integral_constant<int, i + 1 + D>{}));
    lds_wait2<(N - 2 - i < D ? N - 2 - i : D)>(c0, c1);
    use(std::integral_constant<int, i>{}, c0);
    use(std::integral_constant<int, i + 1>{}, c1);
  });
  if constexpr (N % 2 == 1) {
    u32x4 c0 = ring[(N - 1) % D];
    lds_wait<0>(c0);
    use(std::integral_constant<int, N - 1>{}, c0);
  }
#ifdef TANTE_MFMA_SETPRIO
  __builtin_amdgcn_s_setprio(0);
#endif
}

__device__ __forceinline__ u32x4 pack8(const f32x4& a, const f32x4& b) {
  u32x4 f;
  f[0] = pack_bf16x2(a[0], a[1]);
  f[1] = pack_bf16x2(a[2], a[3]);
  f[2] = pack_bf16x2(b[0], b[1]);
  f[3] = pack_bf16x2(b[2], b[3]);
  return f;
}

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

}  // namespace
