// Issue cost (shader cycles per wave-instruction) of the VALU ops a softmax is made of, on gfx950: 16 independent register
// chains per op so that nothing waits on a result; 1 and 3 waves per SIMD.  s_memtime counts at a fixed 100 MHz, so the cycle
// figure is derived from wall time x the measured shader clock of a plain v_fma chain (4 cycles per wave64 instruction).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void k(float* out, int iters) {
  float r[16], q[16];
  for (int i = 0; i < 16; ++i) { r[i] = 0.001f * (threadIdx.x + i); q[i] = 1.0f + 0.01f * i; }
  float c = 0.999f, d = 0.5f;
  for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
#define CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define PERM(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i]), "+v"(q[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
    if (OP == 0) { REP16(FMA) }
    if (OP == 1) { REP16(EXP) }
    if (OP == 2) { REP16(MAX3) }
    if (OP == 3) { REP16(CVT) }
    if (OP == 4) { REP16(SUB) }
    if (OP == 5) { REP16(PERM) }
    if (OP == 6) { REP16(RCP) }
    if (OP == 7) {   // packed f32: two results per lane per instruction
      typedef __attribute__((ext_vector_type(2))) float f2;
      f2* p = (f2*)r; f2 cc = {c, c}, dd = {d, d};
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cc), "v"(dd));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(cc), "v"(dd));
    }
    if (OP == 8) {   // the softmax mix per 2 scores: exp, exp, max3, cvt_pk
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(r[4 * i]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(r[4 * i + 1]));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[4 * i + 2]) : "v"(c), "v"(d));
        asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[4 * i + 3]) : "v"(c));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += r[i] + q[i];
  if (s == 1.2345f) out[0] = s;
}

template <int OP> double run(int threads, int blocks_per_cu, float* out, int n_inst) {
  const int iters = 4000, grid = 256 * blocks_per_cu;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(threads), 0, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // ns per wave-instruction per SIMD: every SIMD runs (threads / 256) * blocks_per_cu waves, each iters * n_inst instructions
  const double waves_per_simd = threads / 256.0 * blocks_per_cu;
  return ms * 1e6 / 5 / (waves_per_simd * iters * (double)n_inst);
}

int main() {
  float* out; hipMalloc(&out, 4);
  const char* names[] = {"v_fma_f32", "v_exp_f32", "v_max3_f32", "v_cvt_pk_bf16_f32", "v_sub_f32", "v_permlane16_swap", "v_rcp_f32", "v_pk_fma_f32", "mix 2exp+max3+cvt"};
  for (int w = 1; w <= 3; w += 2) {
    double t[9];
    t[0] = run<0>(256, w, out, 16); t[1] = run<1>(256, w, out, 16); t[2] = run<2>(256, w, out, 16); t[3] = run<3>(256, w, out, 16);
    t[4] = run<4>(256, w, out, 16); t[5] = run<5>(256, w, out, 16); t[6] = run<6>(256, w, out, 16); t[7] = run<7>(256, w, out, 16);
    t[8] = run<8>(256, w, out, 16);
    printf("%d wave(s) per SIMD: ns per wave-instruction per SIMD (v_fma = 4 shader cycles -> clock %.2f GHz)\n", w, 4.0 / t[0]);
    for (int i = 0; i < 9; ++i) printf("  %-20s %.3f ns = %.2f cycles\n", names[i], t[i], t[i] / t[0] * 4.0);
  }
  return 0;
}
