// What a grid-wide barrier + a small all-to-all data exchange costs inside ONE persistent kernel on gfx950 (256 CUs in 8 XCDs, one L2 per
// XCD): the price of a link of the decode chain if the chain were one kernel instead of five launches per layer (csrc/decode_fused.hip:
// ~5.5 us fixed per launch).  Every phase: workgroup w stores its 512-byte piece of a 128-KB buffer (= the phase number), barrier, every
// workgroup reads the WHOLE buffer and counts pieces that are not the phase number (stale = the exchange is not coherent).
//   mode 0  memory-model form: plain stores, release fence (agent), one counter (atomic add, agent), spin, acquire fence, plain loads
//   mode 1  one counter, no cache maintenance: data stored / loaded with sc0 sc1 (write-through / miss-always), counter relaxed
//   mode 2  flag array: workgroup w stores its flag (sc0 sc1), everybody polls all 256 flags with one 1-KB load per wave; data as mode 1
//   mode 3  mode 2 without the data exchange (barrier only)
//   mode 4  mode 2 + `buffer_wbl2 sc1` in front of the flag store (what the memory model puts in front of a releasing store)
//   mode 5  mode 1 + `buffer_wbl2 sc1` in front of the counter's atomic
//   mode 6  mode 0 without the acquire fences (data loaded with sc0 sc1 instead): which half of mode 0 is the expensive one
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/probe/grid_barrier scripts/probe/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ u32x4 ld_sys16(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned ld_sys4(const void* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_sys16(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sys4(void* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned* data, unsigned* counter, unsigned* flags, unsigned* errs, int phases) {
  const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
  unsigned bad = 0;
  for (int ph = 1; ph <= phases; ++ph) {
    // ---- produce: 512 bytes per workgroup (32 threads x 16 B)
    if (MODE != 3 && tid < 32) {
      const u32x4 v = {(unsigned)ph, (unsigned)ph, (unsigned)ph, (unsigned)ph};
      if (MODE == 0 || MODE == 6) *(u32x4*)(data + wg * 128 + tid * 4) = v;
      else st_sys16(data + wg * 128 + tid * 4, v);
    }
    // ---- barrier
    if (MODE == 0 || MODE == 6) {
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)ph * nwg;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else if (MODE == 1 || MODE == 5) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        if (MODE == 5) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)ph * nwg;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        if (MODE == 4) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        st_sys4(flags + wg, (unsigned)ph);
      }
      if (tid < 64) {                                             // one wave polls all flags: 4 per lane
        for (;;) {
          u32x4 f = ld_sys16(flags + tid * 4);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const bool ok = (tid * 4 + 0 >= nwg || f.x >= (unsigned)ph) && (tid * 4 + 1 >= nwg || f.y >= (unsigned)ph) &&
                          (tid * 4 + 2 >= nwg || f.z >= (unsigned)ph) && (tid * 4 + 3 >= nwg || f.w >= (unsigned)ph);
          if (__builtin_amdgcn_read_exec() == __builtin_amdgcn_ballot_w64(ok)) break;
        }
      }
      __syncthreads();
    }
    // ---- consume: the whole buffer (nwg x 512 B = 128 KB): 512 threads x 16 B x 16
    if (MODE != 3) {
      u32x4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int e = (j * 512 + tid) * 4;
        if (e < nwg * 128) { if (MODE == 0) v[j] = *(const u32x4*)(data + e); else v[j] = ld_sys16(data + e); }
        else v[j] = (u32x4){(unsigned)ph, (unsigned)ph, (unsigned)ph, (unsigned)ph};
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        bad += (v[j].x != (unsigned)ph) + (v[j].w != (unsigned)ph);
        if (v[j].x != (unsigned)ph && (j * 512 + tid) * 4 < nwg * 128) {          // histogram: [same XCD as the writer?][ph - value + 4, clamped to 0..8]
          const int writer = ((j * 512 + tid) * 4) / 128;
          int d = (int)ph - (int)v[j].x + 4; d = d < 0 ? 0 : (d > 8 ? 8 : d);
          atomicAdd(errs + 1 + ((writer & 7) == (wg & 7) ? 9 : 0) + d, 1u);
        }
      }
      // everybody must be done READING phase ph before anyone overwrites it: the next phase's barrier comes after the next store, so
      // alternate two buffers instead of a second barrier
      data += (ph & 1) ? nwg * 128 : -nwg * 128;
    }
  }
  if (bad) atomicAdd(errs, bad);
}

template <int MODE> void run(int nwg, int phases) {
  unsigned *data, *counter, *flags, *errs;
  hipMalloc(&data, 2 * nwg * 512); hipMalloc(&counter, 256); hipMalloc(&flags, 4096); hipMalloc(&errs, 128);
  hipMemset(data, 0, 2 * nwg * 512); hipMemset(counter, 0, 256); hipMemset(flags, 0, 4096); hipMemset(errs, 0, 128);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(512), 0, 0, data, counter, flags, errs, phases);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned e[19]; hipMemcpy(e, errs, 76, hipMemcpyDeviceToHost);
  printf("mode %d  %d workgroups  %d phases: %.3f us per phase, stale pieces seen: %u\n", MODE, nwg, phases, ms * 1e3f / phases, e[0]);
  if (e[0]) {
    printf("    phase - value (-4..+4), writer on another XCD:"); for (int i = 0; i < 9; ++i) printf(" %u", e[1 + i]);
    printf("\n    phase - value (-4..+4), writer on the same XCD: "); for (int i = 0; i < 9; ++i) printf(" %u", e[10 + i]);
    printf("\n");
  }
  hipFree(data); hipFree(counter); hipFree(flags); hipFree(errs);
}

int main(int argc, char** argv) {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int nwg = argc > 1 ? atoi(argv[1]) : pr.multiProcessorCount;
  const int phases = argc > 2 ? atoi(argv[2]) : 2000;
  printf("%s: %d CUs\n", pr.name, pr.multiProcessorCount);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(nwg, phases / 4); run<1>(nwg, phases); run<2>(nwg, phases); run<3>(nwg, phases); run<4>(nwg, phases); run<5>(nwg, phases); run<6>(nwg, phases / 4);
  }
  return 0;
}
