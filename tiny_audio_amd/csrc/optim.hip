// ta355 optimizer kernels (HBM-bound, fp32 masters): global-norm clip + AdamW
// (configs/training/production.yaml:5-9: adamw_torch_fused, max_grad_norm 1.0; decay groups of
// scripts/train.py:427-432).  The clip coefficient is read from device memory so the step needs no host sync.
#include "common.h"

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, long n, float* __restrict__ accum) {
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
  if (threadIdx.x == 0) atomicAdd(accum, red[0] + red[1] + red[2] + red[3]);
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
    const float gi = g[i] * coef;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = pi;
  }
}

extern "C" int ta_grad_sqnorm(const float* g, long n, float* accum, hipStream_t st) {
  if (n <= 0) return TA_OK;
  long b = (n / 4 + 255) / 256; if (b > 1024) b = 1024; if (b < 1) b = 1;
  TA_LAUNCH(sqnorm_kernel, dim3((int)b), dim3(256), 0, st, g, n, accum);
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
