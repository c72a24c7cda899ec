// What a k-step of the one-wave-per-SIMD GEMM (csrc/gemm_v7.hip) pays for its fillers: 4 waves per CU, every wave 64 AGPR-accumulator
// MFMAs (16x16x32) or 32 (32x32x16) per k-step of 32 columns, with / without the 16 fragment reads, the 8 LDS-DMA pieces (spread,
// staggered per wave, or as a burst), plain global loads instead (register staging, with / without the ds_write), the per-k-step
// wait + barrier.  Prints shader cycles per k-step (s_memtime, mean over the waves of the chip) and the wall time.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dma_cost dma_cost.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { RD = 1, DMA = 2, STAG = 4, BURST = 8, GLD = 16, DSW = 32, M32 = 64, SYNC = 128 };

__device__ __forceinline__ void glds16_s(const char* base, unsigned off, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ const char* uniform_ptr(const char* q) {
  const unsigned long v = (unsigned long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long)hi << 32) | lo);
}

template <int MODE, int WV>
__device__ __forceinline__ void kloop(char* smem, const char* src, int iters, int lane, int wave, unsigned long long* cyc, float* sink) {
  constexpr int STAGE = 32 * 1024;
  f32x4 acc[64];
  f32x16 acc32[16];
  if constexpr (MODE & M32) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc32[i][e] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 af[8], bf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { af[i][e] = (short)(0x3c00 + ((lane * 7 + i * 13 + e) & 0x3f)); bf[i][e] = (short)(0xbc00 + ((lane * 5 + i * 11 + e) & 0x3f)); }
  const int rd = (lane & 15) * 64 + ((lane >> 4) << 4);
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem) + wave * 1024;
  unsigned off[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) off[d] = (unsigned)(16 * (wave + 4 * d) + (lane >> 2)) * 2560u + (lane & 3) * 16;
  const char* base = uniform_ptr(src + (size_t)(blockIdx.x & 3) * (512 * 2560));   // four operand windows for the whole chip: L2 hits (a GEMM shares its panels between CUs)
  uint4 stg[8];
  int rs = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const char* Sn = smem + ((rs + 1) & 3) * STAGE;
    const unsigned ist = lds_w + ((rs + 3) & 3) * STAGE;
    const char* b = uniform_ptr(base + (it % 40) * 64);
    constexpr int OFS = (MODE & STAG) ? 2 * WV : 0;
    if constexpr ((MODE & BURST) != 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) glds16_s(b, off[d], ist + d * 4096);
    }
#pragma unroll
    for (int s = 0; s < 64; ++s) {
      const int j = s >> 3, i = s & 7;
      if constexpr (MODE & M32) {
        if ((s & 1) == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc32[s >> 2]) : "v"(bf[j]), "v"(af[i]));
      } else {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[s]) : "v"(bf[j]), "v"(af[i]));
      }
      if constexpr ((MODE & RD) != 0) {
        if (j == 7) af[i] = *(const bf16x8*)(Sn + rd + i * 1024);
        if (i == 7) bf[j] = *(const bf16x8*)(Sn + 8192 + rd + j * 1024);
      }
      constexpr int PH = (MODE & STAG) ? 0 : 4;                  // spread: after MFMA 8 d + 4; staggered: after MFMA 8 d + 2 wave
      if (s >= OFS && (s - OFS) % 8 == PH && (s - OFS) / 8 < 8) {
        const int d = (s - OFS) / 8;
        if constexpr ((MODE & DMA) != 0 && (MODE & BURST) == 0) glds16_s(b, off[d], ist + d * 4096);
        if constexpr ((MODE & GLD) != 0) stg[d] = *(const uint4*)(b + off[d]);
      }
      if constexpr ((MODE & DSW) != 0) {                           // (the piece loaded one k-step ago goes to LDS before its register is re-loaded)
        if (s % 8 == 1 && it > 0) *(uint4*)(smem + ((rs + 2) & 3) * STAGE + (wave + 4 * (s / 8)) * 1024 + lane * 16) = stg[s / 8];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr ((MODE & GLD) != 0 && (MODE & DSW) == 0) {
#pragma unroll
      for (int d = 0; d < 8; ++d) { typedef __attribute__((ext_vector_type(4))) unsigned u4; u4 t = {stg[d].x, stg[d].y, stg[d].z, stg[d].w}; asm volatile("" ::"v"(t)); }
    }
    if constexpr ((MODE & SYNC) != 0) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    rs = (rs + 1) & 3;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s_ = 0.f;
  if constexpr (MODE & M32) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) s_ += acc32[i][e];
  } else {
#pragma unroll
    for (int i = 0; i < 64; ++i) s_ += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  }
  if (s_ == 1.2345f) sink[0] = s_;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const char* src, int iters, unsigned long long* cyc, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[128 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 128 * 1024 / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  __syncthreads();
  if constexpr ((MODE & STAG) != 0) {
    if (wave == 0) kloop<MODE, 0>(smem, src, iters, lane, wave, cyc, sink);
    else if (wave == 1) kloop<MODE, 1>(smem, src, iters, lane, wave, cyc, sink);
    else if (wave == 2) kloop<MODE, 2>(smem, src, iters, lane, wave, cyc, sink);
    else kloop<MODE, 3>(smem, src, iters, lane, wave, cyc, sink);
  } else {
    kloop<MODE, 0>(smem, src, iters, lane, wave, cyc, sink);
  }
}

template <int MODE> void run(const char* name, const char* src, unsigned long long* cyc, float* sink) {
  const int iters = 2000, grid = 256;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, src, 200, cyc, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, src, iters, cyc, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  const double per = s / h.size() / iters;
  printf("%-58s %7.0f cycles / k-step   (MFMA alone: 1024)   wall %8.1f us   %.2f GHz   %6.0f TF/s\n", name, per, ms * 1e3, per * iters / (ms * 1e3) / 1e3,
         (double)grid * 4 * iters * 64 * 16384.0 / (ms * 1e-3) / 1e12);
}

int main() {
  char* src; hipMalloc(&src, (size_t)256 * 512 * 2560 + (1 << 20)); hipMemset(src, 0x3c, (size_t)256 * 512 * 2560 + (1 << 20));
  unsigned long long* cyc; hipMalloc(&cyc, 256 * 4 * 8);
  float* sink; hipMalloc(&sink, 4);
  run<0>("16x16x32 MFMAs only", src, cyc, sink);
  run<RD>("+ 16 fragment reads", src, cyc, sink);
  run<DMA>("+ 8 DMA pieces, spread (no reads)", src, cyc, sink);
  run<RD | DMA>("+ reads + DMA spread (the gemm_v7 body)", src, cyc, sink);
  run<RD | DMA | SYNC>("+ reads + DMA spread + vmcnt(8) + barrier per k-step", src, cyc, sink);
  run<RD | DMA | STAG>("+ reads + DMA, slots staggered per wave", src, cyc, sink);
  run<RD | DMA | STAG | SYNC>("+ reads + DMA staggered + wait + barrier", src, cyc, sink);
  run<RD | DMA | BURST>("+ reads + DMA as one burst", src, cyc, sink);
  run<RD | GLD>("+ reads + 8 global_load_dwordx4 to VGPRs (no ds_write)", src, cyc, sink);
  run<RD | GLD | DSW>("+ reads + 8 global loads + 8 ds_write_b128", src, cyc, sink);
  run<M32>("32x32x16 MFMAs only", src, cyc, sink);
  run<M32 | RD | DMA>("32x32x16 + reads + DMA spread", src, cyc, sink);
  run<M32 | RD | DMA | SYNC>("32x32x16 + reads + DMA spread + wait + barrier", src, cyc, sink);
  run<M32 | RD | DMA | STAG | SYNC>("32x32x16 + reads + DMA staggered + wait + barrier", src, cyc, sink);
  return 0;
}
