// ta355 optimizer kernels (HBM-bound, fp32 masters): global-norm clip + AdamW
// (configs/training/production.yaml:5-9: adamw_torch_fused, max_grad_norm 1.0; decay groups of
// scripts/train.py:427-432).  The clip coefficient is read from device memory so the step needs no host sync.
#include "common.h"
#include "../../include/ta355.h"

// Round 5: DETERMINISTIC.  Every block writes its partial sum to part[blockIdx.x]; a one-block kernel adds the partials in index
// order.  (Rounds 1-4 ended every block with atomicAdd(accum, ...): the order of up to 1 024 float additions depended on which block
// finished first, so two data-parallel ranks holding the SAME all-reduced gradient computed clip coefficients that differed in the
// last bits and their replicas drifted apart bit by bit -- found by tests/test_gpu_round5.py::
// test_trainer_two_ranks_real_kernels_on_one_gpu, the first time the N > 1 path ran on real kernels.)
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = ((const float4*)g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ accum) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) accum[0] += (red[0] + red[1]) + (red[2] + red[3]);
}

// one element of the update; contraction off so that the scalar and the float4 kernel round identically
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float coef, float lr, float wd, float beta1,
                                          float beta2, float eps, float bc1, float bc2) {
#pragma clang fp contract(off)
  const float gi = g * coef;
  float pi = p * (1.0f - lr * wd);
  const float mi = beta1 * m + (1.0f - beta1) * gi;
  const float vi = beta2 * v + (1.0f - beta2) * gi * gi;
  m = mi; v = vi;
  pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  p = pi;
}

// p,g,m,v fp32 [n].  sqnorm: device scalar with sum(g^2) over ALL trainable grads (after the all-reduce), or null.
// grad_scale (host) / denom[0] (device, optional) multiplies g first: grads are accumulated as SUMS of per-token
// gradients and the label-token count travels in the same all-reduce buffer (HF Trainer's sum-CE / global count).
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                             const float* __restrict__ sqnorm, float max_norm, float grad_scale,
                             const float* __restrict__ denom) {
  if (denom) grad_scale /= fmaxf(denom[0], 1.0f);     // e.g. the all-reduced label-token count
  float coef = grad_scale;
  if (sqnorm && max_norm > 0.f) {
    const float total = sqrtf(sqnorm[0]) * fabsf(grad_scale);
    coef *= fminf(1.0f, max_norm / (total + 1e-6f));
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_elem(pi, g[i], mi, vi, coef, lr, wd, beta1, beta2, eps, bc1, bc2);
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}

// The same update over a whole flat buffer of SEGMENTS (one per parameter tensor, every boundary a multiple of 4 elements) in one
// launch: segment s = [seg_end[s-1], seg_end[s]) has base learning rate seg_lr[s] (times lr_mult, the schedule) and weight
// decay seg_wd[s].  float4 accesses; a thread finds its segment by bisection of the (small, cached) boundary table.  Same
// per-element arithmetic as adamw_kernel: bit-identical results.
__global__ __launch_bounds__(256) void adamw_multi_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, long n4, const long* __restrict__ seg_end,
                                                          const float* __restrict__ seg_lr, const float* __restrict__ seg_wd,
                                                          int nseg, float lr_mult, float beta1, float beta2, float eps, float bc1,
                                                          float bc2, const float* __restrict__ sqnorm, float max_norm,
                                                          float grad_scale, const float* __restrict__ denom) {
  if (denom) grad_scale /= fmaxf(denom[0], 1.0f);
  float coef = grad_scale;
  if (sqnorm && max_norm > 0.f) {
    const float total = sqrtf(sqnorm[0]) * fabsf(grad_scale);
    coef *= fminf(1.0f, max_norm / (total + 1e-6f));
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long e0 = i * 4;
    int lo = 0, hi = nseg - 1;                       // first segment whose end is beyond e0
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_end[mid] > e0) hi = mid; else lo = mid + 1; }
    const float lr = seg_lr[lo] * lr_mult, wd = seg_wd[lo];
    const float4 g4 = ((const float4*)g)[i];
    float4 p4 = ((float4*)p)[i], m4 = ((float4*)m)[i], v4 = ((float4*)v)[i];
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
    float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) adam_elem(pp[k], gg[k], mm[k], vv[k], coef, lr, wd, beta1, beta2, eps, bc1, bc2);
    ((float4*)m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    ((float4*)v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    ((float4*)p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
  }
}

extern "C" int ta_grad_sqnorm(const float* g, long n, float* accum, float* scratch, hipStream_t st) {
  if (n <= 0) return TA_OK;
  if (!accum || !scratch) return TA_ERR_ARG;
  long b = (n / 4 + 255) / 256; if (b > TA_SQNORM_SCRATCH_FLOATS) b = TA_SQNORM_SCRATCH_FLOATS; if (b < 1) b = 1;
  TA_LAUNCH(sqnorm_kernel, dim3((int)b), dim3(256), 0, st, g, n, scratch);
  TA_LAUNCH(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, (const float*)scratch, (int)b, accum);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, const float* sqnorm, float max_norm, float grad_scale,
                             const float* denom, hipStream_t st) {
  if (n <= 0) return TA_OK;
  if (step < 1) return TA_ERR_ARG;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  long b = (n + 255) / 256; if (b > 2048) b = 2048;
  TA_LAUNCH(adamw_kernel, dim3((int)b), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                     sqnorm, max_norm, grad_scale, denom);
  TA_CHECK_LAUNCH(); return TA_OK;
}

extern "C" int ta_adamw_step_multi(float* p, const float* g, float* m, float* v, long n, const long* seg_end, const float* seg_lr,
                                   const float* seg_wd, int nseg, float lr_mult, float beta1, float beta2, float eps, int step,
                                   const float* sqnorm, float max_norm, float grad_scale, const float* denom, hipStream_t st) {
  if (n <= 0) return TA_OK;
  if (step < 1 || nseg < 1 || (n & 3) || (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15)) return TA_ERR_ARG;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  long b = (n / 4 + 255) / 256; if (b > 4096) b = 4096;
  TA_LAUNCH(adamw_multi_kernel, dim3((int)b), dim3(256), 0, st, p, g, m, v, n / 4, seg_end, seg_lr, seg_wd, nseg, lr_mult, beta1,
            beta2, eps, bc1, bc2, sqnorm, max_norm, grad_scale, denom);
  TA_CHECK_LAUNCH(); return TA_OK;
}
