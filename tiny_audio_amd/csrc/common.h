// ta355 -- common device helpers for gfx950 (MI355X / CDNA4).  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                   // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;   // MFMA 16x16 C/D fragment

#define TA_OK 0
#define TA_ERR_ARG 1
#define TA_ERR_LAUNCH 2

// Launch through this macro: it first clears HIP's sticky "last error" (the host framework may have left a benign
// one behind, e.g. from a pointer-attribute probe), so that TA_CHECK_LAUNCH reports only OUR launch's status.
#define TA_LAUNCH(...)                \
  do {                                \
    (void)hipGetLastError();          \
    hipLaunchKernelGGL(__VA_ARGS__);  \
  } while (0)

#define TA_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return TA_ERR_LAUNCH;           \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// float -> bf16, round-to-nearest-even, through the gfx950 hardware converter (v_cvt_pk_bf16_f32: one VALU op per
// PAIR instead of ~5 integer ops per element -- the bf16 epilogues were VALU-bound on the manual rounding).
typedef __attribute__((ext_vector_type(2))) __bf16 ta_bf2_t;
typedef __attribute__((ext_vector_type(2))) float ta_f2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  ta_f2_t v = {lo, hi};
  ta_bf2_t b = __builtin_convertvector(v, ta_bf2_t);
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU (nn.GELU() default / nn.functional.gelu)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// erf-GELU for GEMM epilogues of the FROZEN encoder (conv1/conv2/fc1): Abramowitz-Stegun 7.1.26, |erf error| <= 1.5e-7
// -- far below the bf16 output quantum -- in ~15 VALU ops instead of erff's ~45 (the exact-erf epilogue cost fc1
// 7 VALU per MFMA, profiles/r01_a_pmc_summary.md).  Trainable-path GELU (projector) keeps the exact erff above.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = x * 0.70710678118654752f, az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-az * az * 1.4426950408889634f);
  const float erfz = copysignf(1.0f - p * t * e, z);
  return 0.5f * x * (1.0f + erfz);
}

static inline int ta_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
