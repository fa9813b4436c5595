// Loss / metric reductions and the optimiser step of the TANTE train / eval harness.
//   tante_metric_sums     one pass over (pred, ref) -> per (b, t, c) sums over the spatial axes; every metric of
//                         trainer/metrics.py (MSE, NMSE, NNMSE, L2RE, RMSE, NRMSE, VMSE, VRMSE) is a closed form of them
//   tante_mse_grad        d mean(MSE) / d pred  (trainer/trainer.py:189: loss = MSE(y_pred, y_ref).mean())
//   tante_sumsq           sum of squares of a flat fp32 bucket (the global gradient norm of clip_grad_norm_, trainer.py:193)
//   tante_adamw_step      clip-scale + AdamW update over the flat parameter bucket (torch.optim.AdamW, configs/tante.yaml:38-41)
#include "common.hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// pred / ref: (B, T, HW, C) channels-last; pred may be a strided view: element (b,t,s,c) at b*pb + t*pt + s*ps + c*pc.
// One workgroup per (b, t, spatial chunk): lanes own channels (c = lane % C when C <= 64 ...) -- generic: thread i
// walks elements i, i+256, ... of its chunk so that consecutive lanes read consecutive (s, c) of the contiguous ref.
// sums[(b*T + t)*C + c][0..4] += { sum (x-y)^2, sum y^2, sum y, sum (y-p)^2, sum (y-p) }  (fp32 atomics; tiny output) with the pivot
// p = ref[b, t, first pixel, c]: the variance of the 'std' metrics is formed from the SHIFTED moments -- sum y^2 - (sum y)^2 / n
// cancels catastrophically in fp32 for a field whose mean is large against its spread (near-constant pressure / density frames)
__global__ __launch_bounds__(256) void metric_sums_kernel(const float* __restrict__ pred, long pb, long pt, long ps, long pc,
                                                          const float* __restrict__ ref, int T, long HW, int C, long chunk,
                                                          float* __restrict__ sums) {
  extern __shared__ float acc[];  // [C][5]
  const long bt = blockIdx.y;
  const long b = bt / T, t = bt - b * T;
  for (int i = threadIdx.x; i < 5 * C; i += 256) acc[i] = 0.0f;
  __syncthreads();
  const long s0 = (long)blockIdx.x * chunk, s1 = min(HW, s0 + chunk);
  const float* rp = ref + (bt * HW) * C;
  const float* pp = pred + b * pb + t * pt;
  // element index e in [s0*C, s1*C): s = e / C, c = e % C; 256 threads stride by 256 elements; if 256 % C == 0 a thread
  // always sees the same channel and can keep its partial sums in registers
  const bool fixed_c = (256 % C) == 0;
  if (fixed_c) {
    const int c = threadIdx.x % C;
    float d2 = 0.f, y2 = 0.f, y1 = 0.f, z2 = 0.f, z1 = 0.f;
    const float piv = rp[c];
    for (long e = s0 * C + threadIdx.x; e < s1 * C; e += 256) {
      const long s = e / C;
      const float y = rp[e], xv = pp[s * ps + (long)c * pc];
      const float d = xv - y, z = y - piv;
      d2 += d * d; y2 += y * y; y1 += y; z2 += z * z; z1 += z;
    }
    atomicAdd(&acc[5 * c], d2); atomicAdd(&acc[5 * c + 1], y2); atomicAdd(&acc[5 * c + 2], y1);
    atomicAdd(&acc[5 * c + 3], z2); atomicAdd(&acc[5 * c + 4], z1);
  } else {
    for (long e = s0 * C + threadIdx.x; e < s1 * C; e += 256) {
      const long s = e / C;
      const int c = (int)(e - s * C);
      const float y = rp[e], xv = pp[s * ps + (long)c * pc];
      const float d = xv - y, z = y - rp[c];
      atomicAdd(&acc[5 * c], d * d); atomicAdd(&acc[5 * c + 1], y * y); atomicAdd(&acc[5 * c + 2], y);
      atomicAdd(&acc[5 * c + 3], z * z); atomicAdd(&acc[5 * c + 4], z);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 5 * C; i += 256) atomicAdd(&sums[bt * C * 5 + i], acc[i]);
}

// the same for 1, 2 or 4 channels: a thread owns a PIXEL (its C reference values are one contiguous piece; the prediction's C values are
// contiguous too when it is channels-last, C runs of consecutive pixels per wave when it is the model's channels-first output viewed
// channels-last), four pixels in flight, per-thread sums reduced over the wave -- the kernel above keeps one 4-byte load pair in flight per
// thread and ends in 5 LDS atomics per thread on 5 C addresses: 46 us for 50 MB.
template <int C>
__global__ __launch_bounds__(256) void metric_sums_px_kernel(const float* __restrict__ pred, long pb, long pt, long ps, long pc,
                                                             const float* __restrict__ ref, int T, long HW, long chunk, float* __restrict__ sums) {
  __shared__ float part[4][5 * C];
  const long bt = blockIdx.y;
  const long b = bt / T, t = bt - b * T;
  const float* rp = ref + (bt * HW) * C;
  const float* pp = pred + b * pb + t * pt;
  float piv[C], acc[5 * C];
#pragma unroll
  for (int c = 0; c < C; ++c) piv[c] = rp[c];
#pragma unroll
  for (int i = 0; i < 5 * C; ++i) acc[i] = 0.f;
  const long s0 = (long)blockIdx.x * chunk, s1 = min(HW, s0 + chunk);
  auto add = [&](int c, float y, float xv) {
    const float d = xv - y, z = y - piv[c];
    acc[5 * c] += d * d; acc[5 * c + 1] += y * y; acc[5 * c + 2] += y; acc[5 * c + 3] += z * z; acc[5 * c + 4] += z;
  };
  long s = s0 + threadIdx.x;
  for (; s + 768 < s1; s += 1024) {      // four pixels in flight per thread: one load pair per trip left the pass latency-bound (1.1 TB/s)
    float y[4][C], xv[4][C];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) { y[u][c] = rp[(s + 256 * u) * C + c]; xv[u][c] = pp[(s + 256 * u) * ps + (long)c * pc]; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < C; ++c) add(c, y[u][c], xv[u][c]);
  }
  for (; s < s1; s += 256) {
#pragma unroll
    for (int c = 0; c < C; ++c) add(c, rp[s * C + c], pp[s * ps + (long)c * pc]);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 5 * C; ++i) {
    float v = acc[i];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if (lane == 0) part[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5 * C) atomicAdd(&sums[bt * C * 5 + threadIdx.x], (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// grad[b,t,s,c] (contiguous channels-last) = scale * (pred - ref)
__global__ void mse_grad_kernel(const float* __restrict__ pred, long pb, long pt, long ps, long pc, const float* __restrict__ ref,
                                int T, long HW, int C, float scale, float* __restrict__ grad, long n) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int c = (int)(e % C);
  const long r = e / C, s = r % HW, bt = r / HW, b = bt / T, t = bt - b * T;
  grad[e] = scale * (pred[b * pb + t * pt + s * ps + (long)c * pc] - ref[e]);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ out) {
  double s = 0.0;
  const long n4 = (((uintptr_t)g & 15) == 0) ? n / 4 : 0;      // 16-byte pieces (a float sum of four squares, then double), the tail scalar
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = ((const f32x4*)g)[i];
    s += (double)((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
  }
  for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = g[i];
    s += (double)v * v;
  }
  __shared__ double part[4];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// p, m, v, g: flat fp32 buckets.  coef = min(1, max_norm / (||g|| + 1e-6)) is computed on the device from the
// sum of squares so that the step needs no host synchronisation (max_norm <= 0: no clipping).
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g,
                             long n, const double* __restrict__ sumsq, float max_norm, float lr, float b1, float b2, float eps,
                             float wd, float bc1, float bc2_sqrt, float grad_scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float coef = grad_scale;
  if (max_norm > 0.0f) {
    const float total = (float)sqrt(*sumsq) * grad_scale;
    coef *= fminf(max_norm / (total + 1e-6f), 1.0f);
  }
  const float gi = g[i] * coef;
  float pi = p[i] * (1.0f - lr * wd);            // decoupled weight decay
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
}

__global__ void clip_value_kernel(float* __restrict__ g, long n, float c) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // torch.clamp_'s semantics (clip_grad_value_, r_trainer.py:155): +-inf clamps to +-c, NaN STAYS NaN (fminf / fmaxf would turn it into
  // a finite value, and a GradScaler behind the clip would no longer see the overflow and skip the step)
  if (i < n) {
    const float v = g[i];
    g[i] = v != v ? v : fminf(fmaxf(v, -c), c);
  }
}

// dt[b][l] = drt[b] / L : backward of  rt[b] = mean_l clamp_ST(t[b][l]) + ep  (the clamp is straight-through, tante.py:195-198)
__global__ void rt_bwd_kernel(const float* __restrict__ drt, int L, long n, float* __restrict__ dt) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dt[i] = drt[i / L] / (float)L;
}

}  // namespace

static std::atomic<const unsigned long long*> g_seed_mix[64];
const unsigned long long* tante_seed_mix_ptr() {
  int d = 0;
  (void)hipGetDevice(&d);
  return g_seed_mix[d & 63].load(std::memory_order_acquire);
}
extern "C" int tante_set_seed_mix(const uint64_t* device_word) {
  int d = 0;
  (void)hipGetDevice(&d);
  g_seed_mix[d & 63].store((const unsigned long long*)device_word, std::memory_order_release);
  return 0;
}

extern "C" int tante_clip_value(float* g, int64_t n, float clip, void* stream) {
  if (!g || n <= 0 || clip <= 0.0f) TANTE_FAIL(-1, "tante_clip_value: bad argument");
  hipLaunchKernelGGL(clip_value_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, (long)n, clip);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_rt_reduce_bwd(const float* drt, int B, int L, float* dt, void* stream) {
  if (!drt || !dt || B <= 0 || L <= 0) TANTE_FAIL(-1, "tante_rt_reduce_bwd: bad argument");
  const long n = (long)B * L;
  hipLaunchKernelGGL(rt_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, drt, L, n, dt);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_metric_sums(const float* pred, int64_t pb, int64_t pt, int64_t ps, int64_t pc, const float* ref, int B, int T,
                                 int64_t HW, int C, float* sums, void* stream) {
  if (!pred || !ref || !sums || B <= 0 || T <= 0 || HW <= 0 || C <= 0) TANTE_FAIL(-1, "tante_metric_sums: bad argument");
  if (C > 1024) TANTE_FAIL(-2, "tante_metric_sums: too many channels");
  hipStream_t s = (hipStream_t)stream;
  if (tante_zero_async(sums, (size_t)B * T * C * 5 * sizeof(float), s) != hipSuccess) TANTE_FAIL(-3, "tante_metric_sums: memset failed");
  if ((C == 4 || C == 2 || C == 1) && HW >= 4096) {
    const long pchunk = 4096;      // 16 pixels per thread
    const dim3 grid((unsigned)((HW + pchunk - 1) / pchunk), (unsigned)(B * T));
    if (C == 4) hipLaunchKernelGGL(metric_sums_px_kernel<4>, grid, dim3(256), 0, s, pred, (long)pb, (long)pt, (long)ps, (long)pc, ref, T, (long)HW, pchunk, sums);
    else if (C == 2) hipLaunchKernelGGL(metric_sums_px_kernel<2>, grid, dim3(256), 0, s, pred, (long)pb, (long)pt, (long)ps, (long)pc, ref, T, (long)HW, pchunk, sums);
    else hipLaunchKernelGGL(metric_sums_px_kernel<1>, grid, dim3(256), 0, s, pred, (long)pb, (long)pt, (long)ps, (long)pc, ref, T, (long)HW, pchunk, sums);
    TANTE_CHECK_LAUNCH();
    return 0;
  }
  long chunks = (HW * C + 256 * 64 - 1) / (256 * 64);  // ~64 elements per thread
  if (chunks < 1) chunks = 1;
  const long chunk = (HW + chunks - 1) / chunks;
  hipLaunchKernelGGL(metric_sums_kernel, dim3((unsigned)((HW + chunk - 1) / chunk), (unsigned)(B * T)), dim3(256), 5 * C * sizeof(float), s,
                     pred, (long)pb, (long)pt, (long)ps, (long)pc, ref, T, (long)HW, C, chunk, sums);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_mse_grad(const float* pred, int64_t pb, int64_t pt, int64_t ps, int64_t pc, const float* ref, int B, int T,
                              int64_t HW, int C, float scale, float* grad, void* stream) {
  if (!pred || !ref || !grad || B <= 0 || T <= 0 || HW <= 0 || C <= 0) TANTE_FAIL(-1, "tante_mse_grad: bad argument");
  const long n = (long)B * T * HW * C;
  hipLaunchKernelGGL(mse_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, (long)pb, (long)pt,
                     (long)ps, (long)pc, ref, T, (long)HW, C, scale, grad, n);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_sumsq(const float* g, int64_t n, double* out, void* stream) {
  if (!g || !out || n <= 0) TANTE_FAIL(-1, "tante_sumsq: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (tante_zero_async(out, sizeof(double), s) != hipSuccess) TANTE_FAIL(-3, "tante_sumsq: memset failed");
  long blocks = (n + 256 * 16 - 1) / (256 * 16);
  if (blocks > 256) blocks = 256;      // every workgroup ends in ONE atomic on the same double: a thousand of them were most of this kernel's 20 us
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, (long)n, out);
  TANTE_CHECK_LAUNCH();
  return 0;
}

extern "C" int tante_adamw_step(float* p, float* m, float* v, const float* g, int64_t n, const double* sumsq, float max_norm,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                                void* stream) {
  if (!p || !m || !v || !g || n <= 0 || step < 1) TANTE_FAIL(-1, "tante_adamw_step: bad argument");
  if (max_norm > 0.0f && !sumsq) TANTE_FAIL(-1, "tante_adamw_step: clipping needs the gradient sum of squares");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, m, v, g, (long)n, sumsq,
                     max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale);
  TANTE_CHECK_LAUNCH();
  return 0;
}
