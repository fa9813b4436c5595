// Microbenchmark (tools/ubench): what a block-backward workgroup's bf16 ROW STORES cost by the shape of one store instruction.
// 512 workgroups x 4 waves, a workgroup owns 48 rows of 512 bytes in each of NT tensors (cfg3: 24 576 tokens x 256 bf16 = 12.6 MB each),
// every wave stores 16 bytes per lane, 6 instructions per tensor:
//   piece  : the wave's own 128-byte column slice of 8 rows per instruction  (8 lanes per row: block_bwd_fs.hip's img_rows_store)
//   row    : 2 whole 512-byte rows per instruction                           (32 lanes per row; needs the other waves' columns: a barrier)
//   piece8 : 8 bytes per lane, 32 bytes per row and wave, 16 rows per instruction, 12 instructions (accumulator layout: block_bwd.hip)
// rows of a workgroup are contiguous (W letter) or `stride` rows apart (H / T letters).  Prints GB/s of stored bytes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/row_stores tools/ubench/row_stores.hip && tools/ubench/row_stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(unsigned short* __restrict__ dst, long tensor_elems, int nt, long rows_total, int seqstride, int spin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // workgroup's 48 rows: row r of the workgroup = token (blockIdx.x * 48 + r) for seqstride == 1, else a strided pattern
  const long wg = blockIdx.x;
  auto tok = [&](int r) -> long {
    if (seqstride == 1) return wg * 48 + r;
    // 3 sequences of 16 rows, rows of a sequence `seqstride` tokens apart (H letter: stride Wp = 48)
    const long s = wg * 3 + r / 16, l = r % 16;
    return (s / seqstride) * (16L * seqstride) + (s % seqstride) + l * seqstride;
  };
  u32x4 v = u32x4{(unsigned)lane, (unsigned)wave, (unsigned)wg, 7u};
  for (int t = 0; t < nt; ++t) {
    unsigned short* base = dst + (long)t * tensor_elems;
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int r = 8 * j + (lane >> 3);
        *(u32x4*)(base + tok(r) * 256 + 64 * wave + 8 * (lane & 7)) = v;
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int r = 2 * (wave + 4 * j) + (lane >> 5);
        *(u32x4*)(base + tok(r) * 256 + 8 * (lane & 31)) = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int r = 16 * (j / 4) + (lane & 15);
        *(u32x2*)(base + tok(r) * 256 + 64 * wave + 16 * (j % 4) + 4 * (lane >> 4)) = u32x2{v[0], v[1]};
      }
    }
    // some compute between tensors (the GEMM phase): spin iterations of dependent FMAs
    float a = (float)lane;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    if (a == 123.456f) v[3] = 1u;
  }
}

int main() {
  const long rows = 24576, elems = rows * 256;
  const int nt = 8;      // 8 tensors = 100 MB per launch
  unsigned short* d;
  hipMalloc(&d, (size_t)nt * elems * 2 * 4);      // 4 rotating sets
  hipMemset(d, 0, (size_t)nt * elems * 2 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"piece (8 rows x 128 B)", "row (2 rows x 512 B)", "piece8 (16 rows x 32 B, 8 B per lane)"};
  for (int spin : {0, 2000}) {
    for (int stride : {1, 48, 768}) {
      for (int mode = 0; mode < 3; ++mode) {
        auto launch = [&](int set) {
          unsigned short* p = d + (size_t)set * nt * elems;
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 0, 0, p, elems, nt, rows, stride, spin);
          else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, p, elems, nt, rows, stride, spin);
          else hipLaunchKernelGGL(k<2>, dim3(512), dim3(256), 0, 0, p, elems, nt, rows, stride, spin);
        };
        for (int i = 0; i < 4; ++i) launch(i);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch(i & 3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = 1e3 * ms / reps, gb = (double)nt * elems * 2 / 1e9;
        printf("spin %4d  row stride %3d  %-40s %7.1f us per launch  %7.1f GB/s\n", spin, stride, names[mode], us, gb / (us * 1e-6));
      }
    }
  }
  return 0;
}
