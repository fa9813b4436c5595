// Microbenchmark: what a 16-byte-per-lane global access costs the CU's memory pipe by ADDRESS PATTERN.  One workgroup of NW waves per
// CU; every wave issues `n` independent global_load_dwordx4 (or stores) over an L2-resident buffer (loads: one megabyte shared by all waves; stores: 1.1 MB per wave, i.e. write-combining in L2) and the kernel reports the shader
// clocks per wave-instruction, averaged over the CU's waves (they share one texture-address unit).
//   pattern 0: lane-linear (1 KiB contiguous per instruction)
//   pattern 1: MFMA operand rows: lane & 15 -> row (1 KiB apart), lane >> 4 -> 16-byte piece     (16 rows x 64 B per instruction)
//   pattern 2: 4 rows x 256 B: lane >> 4 -> row, lane & 15 -> piece                               (LayerNorm-style row reads)
//   pattern 3: lane & 15 -> row, lane >> 4 -> piece at 32-byte pitch                               (the head's read-modify-write)
//   pattern 4: lane-linear 8 bytes per lane (dwordx2), 512 B contiguous
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/ta_cost tools/ubench/ta_cost.hip ; run: tools/ubench/ta_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int PAT, bool STORE>
__global__ __launch_bounds__(512) void k(float* buf, long wave_stride, int n, unsigned long long* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* base = buf;   // every wave reads the same L2-resident megabyte (stores: private, `wave_stride` floats apart)
  if (STORE) base += ((long)blockIdx.x * 8 + wave) * wave_stride;
  long off;
  if (PAT == 0) off = lane * 4;
  else if (PAT == 1) off = (long)(lane & 15) * 256 + (lane >> 4) * 4;
  else if (PAT == 2) off = (long)(lane >> 4) * 256 + (lane & 15) * 4;
  else if (PAT == 3) off = (long)(lane & 15) * 8 + (lane >> 4) * 65536;
  else off = lane * 2;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; i += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float* p = base + off + (long)(((i + j) * 7 + wave * 3 + blockIdx.x) & 31) * 4096;       // 32 windows of 16 KiB
      if (STORE) *(f32x4*)p = f32x4{(float)i, 1.f, 2.f, 3.f};
      else v[j] = __builtin_nontemporal_load((const f32x4*)p);
    }
    if (!STORE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (acc[0] == 1.2345f) buf[0] = acc[1];
}

template <int PAT, bool STORE>
void run(float* buf, long wave_stride, int n, unsigned long long* out, int nwg, const char* name) {
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<PAT, STORE>), dim3(nwg), dim3(512), 0, 0, buf, wave_stride, n, out);
  hipDeviceSynchronize();
  unsigned long long h[8 * 1024];
  hipMemcpy(h, out, sizeof(unsigned long long) * nwg * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < nwg * 8; ++i) s += (double)h[i];
  printf("%-44s %s: %7.1f clocks per wave-instruction (8 waves per CU -> %6.1f per CU-instruction)\n", name, STORE ? "store" : "load ", s / (nwg * 8) / n,
         s / (nwg * 8) / n / 8);
}

int main() {
  const int nwg = 256, n = 256;
  const long wave_stride = 32L * 4096 + 65536 * 4;    // floats: 32 windows of 16 KiB + the 4 planes of pattern 3
  float* buf; unsigned long long* out;
  hipMalloc(&buf, sizeof(float) * wave_stride * nwg * 8);
  hipMemset(buf, 0, sizeof(float) * wave_stride * nwg * 8);
  hipMalloc(&out, sizeof(unsigned long long) * nwg * 8);
  run<0, false>(buf, wave_stride, n, out, nwg, "lane-linear 1 KiB");
  run<1, false>(buf, wave_stride, n, out, nwg, "16 rows x 64 B (lane&15 = row)");
  run<2, false>(buf, wave_stride, n, out, nwg, "4 rows x 256 B (lane>>4 = row)");
  run<3, false>(buf, wave_stride, n, out, nwg, "16 B at 32 B pitch, 4 planes");
  run<0, true>(buf, wave_stride, n, out, nwg, "lane-linear 1 KiB");
  run<1, true>(buf, wave_stride, n, out, nwg, "16 rows x 64 B (lane&15 = row)");
  run<2, true>(buf, wave_stride, n, out, nwg, "4 rows x 256 B (lane>>4 = row)");
  run<3, true>(buf, wave_stride, n, out, nwg, "16 B at 32 B pitch, 4 planes");
  return 0;
}
