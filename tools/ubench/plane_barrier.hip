// Microbenchmark for the round-3 verdict's item 1(c): what does fusing the H-letter and the W-letter block launches of one (b, t) plane
// behind a PLANE-SCOPED barrier buy?  The seam is priced without the blocks' arithmetic: every workgroup (256 threads, 64 tokens x 1 KiB
// fp32 rows, two workgroups per CU -- block_fs_kernel's geometry at cfg2, B = 8: 512 workgroups, 32 planes x 16) reads its 64 token rows
// in the H letter's grouping (two columns of the 32 x 32 plane), busy-waits CORE ticks of the shader clock (the block's compute core,
// 26 us measured with its memory phases ablated), writes the rows back, and does the same again in the W letter's grouping (two rows
// of the plane) -- every row it reads there was written by another workgroup of the same plane.
//   mode 0   two launches, write-through stores (what the product does today)
//   mode 1   ONE launch; plain stores, no release (the 16 workgroups of a plane share an XCD and its L2: blockIdx -> (xcd, plane, part),
//            checked through HW_REG_XCC_ID), arrival counter per plane, sc1-load poll + s_sleep, agent acquire (L1 invalidate), plain loads
//   mode 2   ONE launch; sc1 (write-through) stores, arrival, poll, sc1 loads, no fence (MI355X_MICROARCH.md's measured hand-off form)
//   mode 3   ONE launch; plain stores, agent RELEASE fence (L2 write-back) by one lane, arrival, poll, agent acquire, plain loads
// Prints us per H + W pair for each mode and CORE in {0, 26 us}; checks the data (every row incremented exactly twice per pair).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ f32x4 ld_plain(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 ld_sc1(const float* p) {
  f32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
  return r;
}
__device__ __forceinline__ void st_plain(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void st_sc1(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_wt(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }

__device__ __forceinline__ void busy(unsigned long long ticks) {
  if (!ticks) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}

// token of (plane, letter, part, slot): H letter: columns 2 part, 2 part + 1, slot = 32 * c + h;  W letter: rows 2 part, 2 part + 1, slot = 32 * r + w
__device__ __forceinline__ long token_of(int plane, int letter, int part, int slot) {
  const int a = slot >> 5, l = slot & 31;
  return letter == 0 ? (long)plane * 1024 + l * 32 + (2 * part + a) : (long)plane * 1024 + (2 * part + a) * 32 + l;
}

template <int LD, int ST>      // LD 0 plain 1 sc1;  ST 0 plain 1 sc1 2 write-through (sc0 sc1)
__device__ __forceinline__ void phase(float* x, int plane, int letter, int part, unsigned long long core) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float* p = x + token_of(plane, letter, part, wave * 16 + j) * 256 + lane * 4;
    v[j] = LD ? ld_sc1(p) : ld_plain(p);
  }
  if (LD)      // the asm loads are invisible to the compiler's wait insertion: one counted wait tied to all sixteen registers
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
                   "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                 :
                 : "memory");
  busy(core);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float* p = x + token_of(plane, letter, part, wave * 16 + j) * 256 + lane * 4;
    const f32x4 o = v[j] + 1.0f;
    if (ST == 0) st_plain(p, o); else if (ST == 1) st_sc1(p, o); else st_wt(p, o);
  }
}

__device__ __forceinline__ void map_block(int& plane, int& part) {
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;      // blocks b and b + 8 share an XCD (observed round-robin placement)
  plane = xcd + 8 * (j >> 4);
  part = j & 15;
}

__global__ __launch_bounds__(256, 2) void one_letter(float* x, int letter, unsigned long long core) {
  int plane, part;
  map_block(plane, part);
  phase<0, 2>(x, plane, letter, part, core);
}

__device__ unsigned long long g_stamps[512 * 4];
template <int MODE>
__global__ __launch_bounds__(256, 2) void fused_pair(float* x, int* cnt, unsigned* xccmask, unsigned long long core) {
  int plane, part;
  map_block(plane, part);
  if (threadIdx.x == 0) g_stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) atomicOr(xccmask + plane, 1u << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u));
  phase<0, MODE == 2 ? 1 : 0>(x, plane, 0, part, core);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE == 4) {
  } else if (threadIdx.x == 0) {
    if (MODE == 3) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    g_stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
    __hip_atomic_fetch_add(cnt + plane, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;      // bounded: a barrier that cannot complete (workgroups of a plane not co-resident) must not hang the GPU
#ifdef POLL_RMW
    while (MODE != 5 && __hip_atomic_fetch_add(cnt + plane, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16) {
#else
    while (MODE != 5 && __hip_atomic_load(cnt + plane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16) {
#endif
      __builtin_amdgcn_s_sleep(8);
      if (++spins > 400000) { atomicOr(xccmask + plane, 0x80000000u); break; }
    }
    if (MODE != 2 && MODE != 5) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  __syncthreads();
  if (threadIdx.x == 0) g_stamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memtime();
  phase<MODE == 2 ? 1 : 0, 2>(x, plane, 1, part, core);
  if (threadIdx.x == 0) g_stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime();
}

__global__ void calib(unsigned long long ticks) { busy(ticks); }

int main(int argc, char** argv) {
  const int planes = 32, nwg = planes * 16, iters = 40;
  const long n = (long)planes * 1024 * 256;
  float* x; int* cnt; unsigned* mask;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&cnt, (iters + 8) * planes * 4)); CK(hipMalloc(&mask, planes * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int dev_clock_khz = 0; CK(hipDeviceGetAttribute(&dev_clock_khz, hipDeviceAttributeClockRate, 0));
  printf("plane barrier microbenchmark: %d workgroups x 256 threads, 64 rows of 1 KiB each, %d planes x 16; device clock %d kHz\n", nwg, planes, dev_clock_khz);
  // s_memtime ticks per microsecond, measured: one workgroup busy-waits 2 M ticks
  double ticks_per_us = 100.0;
  {
    hipLaunchKernelGGL(calib, dim3(1), dim3(64), 0, 0, 1000ull);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(calib, dim3(1), dim3(64), 0, 0, 2000000ull);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    ticks_per_us = 2000000.0 / (1e3 * ms);
    printf("s_memtime: %.1f ticks per us\n", ticks_per_us);
  }
  for (double core_us : {0.0, 26.0}) {
    for (int mode = 0; mode < 6; ++mode) {
      CK(hipMemset(x, 0, n * 4)); CK(hipMemset(cnt, 0, (iters + 8) * planes * 4)); CK(hipMemset(mask, 0, planes * 4));
      const unsigned long long core = (unsigned long long)(core_us * ticks_per_us);
      auto pair = [&](int it) {
        if (mode == 0) {
          hipLaunchKernelGGL(one_letter, dim3(nwg), dim3(256), 0, 0, x, 0, core);
          hipLaunchKernelGGL(one_letter, dim3(nwg), dim3(256), 0, 0, x, 1, core);
        } else if (mode == 1) hipLaunchKernelGGL(fused_pair<1>, dim3(nwg), dim3(256), 0, 0, x, cnt + it * planes, mask, core);
        else if (mode == 2) hipLaunchKernelGGL(fused_pair<2>, dim3(nwg), dim3(256), 0, 0, x, cnt + it * planes, mask, core);
        else if (mode == 3) hipLaunchKernelGGL(fused_pair<3>, dim3(nwg), dim3(256), 0, 0, x, cnt + it * planes, mask, core);
        else if (mode == 4) hipLaunchKernelGGL(fused_pair<4>, dim3(nwg), dim3(256), 0, 0, x, cnt + it * planes, mask, core);
        else hipLaunchKernelGGL(fused_pair<5>, dim3(nwg), dim3(256), 0, 0, x, cnt + it * planes, mask, core);
      };
      for (int w = 0; w < 4; ++w) pair(w);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < iters; ++it) pair(4 + it);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<float> h(1 << 16);
      CK(hipMemcpy(h.data(), x + 12345 * 256, h.size() * 4, hipMemcpyDeviceToHost));
      bool ok = true;
      for (float v : h) ok = ok && v == 2.0f * (iters + 4);
      std::vector<unsigned> hm(planes);
      CK(hipMemcpy(hm.data(), mask, planes * 4, hipMemcpyDeviceToHost));
      bool one_xcd = true;
      bool timed_out = false;
      for (unsigned m : hm) { timed_out = timed_out || (m & 0x80000000u); m &= 0x7fffffffu; one_xcd = one_xcd && (mode == 0 || (m && !(m & (m - 1)))); }
      if (timed_out) printf("  !! a plane barrier timed out (workgroups not co-resident)\n");
      if (mode == 2 && argc > 1) {
        std::vector<unsigned long long> st(512 * 4);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
        for (int pl : {0, 17}) {
          unsigned long long t0 = ~0ull;
          for (int b = 0; b < 512; ++b) { int xcd = b & 7, j = b >> 3; if (xcd + 8 * (j >> 4) == pl && st[b * 4] < t0) t0 = st[b * 4]; }
          printf("  plane %d (ticks since its first start): ", pl);
          for (int b = 0; b < 512; ++b) { int xcd = b & 7, j = b >> 3; if (xcd + 8 * (j >> 4) == pl) printf("[b%d %llu %llu %llu %llu] ", b, st[b * 4] - t0, st[b * 4 + 1] - t0, st[b * 4 + 2] - t0, st[b * 4 + 3] - t0); }
          printf("\n");
        }
      }
      printf("core %4.0f us  mode %d  %-58s %7.2f us per H+W pair   data %s   %s\n", core_us, mode,
             mode == 0 ? "two launches, write-through stores" : mode == 1 ? "one launch: plain stores, no release, acquire, plain loads"
             : mode == 2 ? "one launch: sc1 stores, sc1 loads, no fence" : mode == 3 ? "one launch: plain stores, release + acquire fences, plain loads"
             : mode == 4 ? "one launch, NO barrier (timing only, data wrong)" : "one launch, arrive but no wait (timing only, data wrong)",
             1e3 * ms / iters, ok ? "ok" : "WRONG", mode == 0 ? "" : (one_xcd ? "every plane on one XCD" : "planes SPLIT over XCDs"));
    }
  }
  return 0;
}
