// Microbenchmark: what two waves on one SIMD share.  One workgroup of 512 threads per CU (8 waves = 2 per SIMD); waves 0-3 run
// role A, waves 4-7 role B; roles: 0 idle, 1 MFMA 16x16x32 bf16 stream, 2 v_fma_f32 stream, 3 v_pk_fma_f32 stream, 4 v_exp_f32 stream,
// 5 mixed (1 MFMA + 2 v_fma per iteration).  Prints cycles (s_memtime) per role-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, float seed) {
  if constexpr (ROLE == 0) return seed;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) p[i] = f32x2{seed + i, seed - i};
  for (int it = 0; it < iters; ++it) {
    if constexpr (ROLE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    } else if constexpr (ROLE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
    } else if constexpr (ROLE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    } else if constexpr (ROLE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (ROLE >= 6 && ROLE <= 9) {     // 1 MFMA + (ROLE - 5) independent v_fma (6: 1, 7: 2 ... ) -- NV = ROLE - 5 + (ROLE == 9 ? 1 : 0)
      constexpr int NV = ROLE == 6 ? 1 : ROLE == 7 ? 3 : ROLE == 8 ? 4 : 6;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[(i * NV + q) & 15]) : "v"(seed));
      }
    } else if constexpr (ROLE == 10) {                 // one dependent v_fma chain
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[0]) : "v"(seed));
    } else if constexpr (ROLE == 11) {                 // 1 MFMA + 1 ds_read_b128-free pk_fma + 2 fma
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[2 * i]) : "v"(seed));
      }
    } else if constexpr (ROLE == 5) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[2 * i]) : "v"(seed));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[2 * i + 1]) : "v"(seed));
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3] + p[i][0] + p[i][1];
  for (int i = 0; i < 16; ++i) r += v[i];
  return r;
}

template <int RA, int RB>
__global__ __launch_bounds__(512, 2) void k(int iters, float seed, float* out, unsigned long long* cyc) {
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float r = wave < 4 ? run_role<RA>(iters, seed) : run_role<RB>(iters, seed);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (r == 12345.678f) out[0] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int RA, int RB>
void go(const char* name, int per_iter_a, int per_iter_b) {
  const int iters = 2000, nb = 256;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, nb * 8 * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<RA, RB>), dim3(nb), dim3(512), 0, 0, iters, 1.0f, out, cyc);
  hipDeviceSynchronize();
  unsigned long long h[256 * 8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int i = 0; i < nb; ++i) { for (int w = 0; w < 4; ++w) a += h[i * 8 + w]; for (int w = 4; w < 8; ++w) b += h[i * 8 + w]; }
  a /= nb * 4.0; b /= nb * 4.0;
  printf("%-34s  A: %8.0f cyc", name, a);
  if (per_iter_a) printf(" = %6.2f cyc/instr", a / (iters * (double)per_iter_a));
  printf("   B: %8.0f cyc", b);
  if (per_iter_b) printf(" = %6.2f cyc/instr", b / (iters * (double)per_iter_b));
  printf("\n");
  hipFree(out); hipFree(cyc);
}

int main() {
  go<1, 0>("mfma | idle", 8, 0);
  go<1, 1>("mfma | mfma", 8, 8);
  go<2, 0>("v_fma | idle", 16, 0);
  go<2, 2>("v_fma | v_fma", 16, 16);
  go<3, 0>("v_pk_fma | idle", 8, 0);
  go<3, 3>("v_pk_fma | v_pk_fma", 8, 8);
  go<4, 0>("v_exp | idle", 16, 0);
  go<4, 4>("v_exp | v_exp", 16, 16);
  go<1, 2>("mfma | v_fma", 8, 16);
  go<1, 3>("mfma | v_pk_fma", 8, 8);
  go<1, 4>("mfma | v_exp", 8, 16);
  go<5, 0>("mixed(1 mfma + 2 fma) | idle", 24, 0);
  go<5, 5>("mixed | mixed", 24, 24);
  go<10, 0>("dependent v_fma chain | idle", 16, 0);
  go<10, 10>("dep chain | dep chain", 16, 16);
  go<1, 10>("mfma | dep chain", 8, 16);
  go<6, 0>("1 mfma + 1 fma | idle (per group)", 8, 0);
  go<7, 0>("1 mfma + 3 fma | idle (per group)", 8, 0);
  go<8, 0>("1 mfma + 4 fma | idle (per group)", 8, 0);
  go<9, 0>("1 mfma + 6 fma | idle (per group)", 8, 0);
  go<6, 6>("1 mfma + 1 fma | same (per group)", 8, 8);
  go<7, 7>("1 mfma + 3 fma | same (per group)", 8, 8);
  go<8, 8>("1 mfma + 4 fma | same (per group)", 8, 8);
  go<9, 9>("1 mfma + 6 fma | same (per group)", 8, 8);
  go<11, 0>("1 mfma + pk_fma + fma | idle (per group)", 8, 0);
  go<11, 11>("1 mfma + pk_fma + fma | same (per group)", 8, 8);
  go<2, 7>("v_fma | 1 mfma + 3 fma", 16, 8);
  return 0;
}
