// Sustained MFMA issue rate by shape on gfx950: 8 waves per CU (2 per SIMD) issuing dependent-free MFMAs from registers.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k16(float* out, int iters, unsigned seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + ((threadIdx.x * 7 + i * 13 + seed) & 0x7f)); b[i] = (short)(0xbf80 + ((threadIdx.x * 5 + i * 11 + seed) & 0x7f)); }
  f32x4 acc[40];
  for (int i = 0; i < 40; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 40; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345f) out[0] = s;
}
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k32(float* out, int iters, unsigned seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + ((threadIdx.x * 7 + i * 13 + seed) & 0x7f)); b[i] = (short)(0xbf80 + ((threadIdx.x * 5 + i * 11 + seed) & 0x7f)); }
  f32x16 acc[10];
  for (int i = 0; i < 10; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 10; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 1.2345f) out[0] = s;
}
template <typename F> double run(F launch, double flops_per_launch) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return flops_per_launch * 10 / (ms * 1e-3) / 1e12;
}
int main() {
  float* out; hipMalloc(&out, 4);
  const int iters = 2000, grid = 256 * 4;
  // per launch: grid blocks x waves x iters x MFMAs x flops
  printf("16x16x32  8 waves/CU-block: %.0f TF/s\n", run([&] { hipLaunchKernelGGL(k16<8>, dim3(grid), dim3(512), 0, 0, out, iters, 1u); }, (double)grid * 8 * iters * 40 * 16384.0));
  printf("32x32x16  8 waves/CU-block: %.0f TF/s\n", run([&] { hipLaunchKernelGGL(k32<8>, dim3(grid), dim3(512), 0, 0, out, iters, 1u); }, (double)grid * 8 * iters * 10 * 32768.0));
  printf("16x16x32  4 waves/CU-block: %.0f TF/s\n", run([&] { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1u); }, (double)grid * 4 * iters * 40 * 16384.0));
  printf("32x32x16  4 waves/CU-block: %.0f TF/s\n", run([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1u); }, (double)grid * 4 * iters * 10 * 32768.0));
  return 0;
}
