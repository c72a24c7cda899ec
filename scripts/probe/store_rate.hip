// Per-CU global-store issue rate on gfx950: one 512-thread workgroup per CU writes a 256 x 640-B tile of a row-major matrix
// from registers (no loads, no LDS), in the access shapes a GEMM epilogue can choose from, timed with s_memtime around the
// store loop (issue) and after s_waitcnt vmcnt(0) (acknowledged).  Large LDS allocation forces one workgroup per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// MODE 0: 16 B per lane, fully contiguous (1 KB per instruction)        -- the staged epilogue
// MODE 1: 16 B per lane, 16 rows x 64 B per instruction                  -- the permlane-swapped "wide" epilogue
// MODE 2: 8 B per lane, 16 rows x 32 B per instruction                   -- the raw MFMA accumulator layout in bf16
// MODE 3: 4 B per lane, fully contiguous (256 B per instruction)
// MODE 4: 16 B per lane, fully contiguous, nontemporal
// MODE 5: 16 B per lane, 8 rows x 128 B per instruction, line-aligned       -- what an in-register (DPP) row merge could reach
// MODE 6: as 5 but every segment starts 32 B into a line (the 80-column wave spans of the 320-wide tile)
// MODE 7: 16 B per lane, 4 rows x 256 B per instruction
// MODE 8: 16 B per lane, rows of 160 B (10 lanes per row, 6.4 rows per instruction) -- wave-local staging of an 80-column span
template <int MODE>
__global__ __launch_bounds__(512) void k(char* out, long ld_bytes, unsigned long long* stamps, int reps) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* tile = out + (long)blockIdx.x * 256 * ld_bytes;            // 256 rows of this workgroup
  uint4 v = make_uint4(tid, tid * 3, tid * 5, tid * 7);
  if (tid == 1000) lds[0] = 1;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0 || MODE == 4) {
#pragma unroll
      for (int it = 0; it < 20; ++it) {                             // 256 rows x 40 chunks / 512 threads
        const int item = it * 512 + tid, row = item / 40, ch = item - row * 40;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        if (MODE == 4) __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, (u32x4*)(tile + row * ld_bytes + ch * 16));
        else *(uint4*)(tile + row * ld_bytes + ch * 16) = v;
      }
    } else if (MODE == 1) {                                          // wave: 32 rows x 640 B as 2 row groups x 10 column groups of 64 B
#pragma unroll
      for (int it = 0; it < 20; ++it) {
        const int rg = it / 10, cg = it % 10;
        *(uint4*)(tile + (wave * 32 + rg * 16 + (lane & 15)) * ld_bytes + cg * 64 + (lane >> 4) * 16) = v;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int it = 0; it < 40; ++it) {
        const int rg = it / 20, cg = it % 20;
        *(uint2*)(tile + (wave * 32 + rg * 16 + (lane & 15)) * ld_bytes + cg * 32 + (lane >> 4) * 8) = make_uint2(v.x, v.y);
      }
    } else if (MODE == 5 || MODE == 6) {                             // wave: 32 rows x 640 B as 4 row groups of 8 x 5 column groups of 128 B
#pragma unroll
      for (int it = 0; it < 20; ++it) {
        const int rg = it / 5, cg = it % 5;
        *(uint4*)(tile + (wave * 32 + rg * 8 + (lane >> 3)) * ld_bytes + cg * 128 + (lane & 7) * 16 + (MODE == 6 ? 32 : 0)) = v;
      }
    } else if (MODE == 7) {                                          // 4 rows x 256 B; columns 0..511 + a 128-B remainder handled as 8 x 128
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int rg = it / 2, cg = it % 2;
        *(uint4*)(tile + (wave * 32 + rg * 4 + (lane >> 4)) * ld_bytes + cg * 256 + (lane & 15) * 16) = v;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it)
        *(uint4*)(tile + (wave * 32 + it * 8 + (lane >> 3)) * ld_bytes + 512 + (lane & 7) * 16) = v;
    } else if (MODE == 8) {                                          // wave w: 128 rows x 160 B at column byte 160 * (w & 3), rows (w >> 2) * 128 ..
#pragma unroll
      for (int it = 0; it < 20; ++it) {
        const int item = it * 64 + lane, row = item / 10, ch = item - row * 10;
        *(uint4*)(tile + ((wave >> 2) * 128 + row) * ld_bytes + (wave & 3) * 160 + ch * 16) = v;
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int it = 0; it < 80; ++it) {
        const int item = it * 512 + tid, row = item / 160, ch = item - row * 160;
        *(unsigned*)(tile + row * ld_bytes + ch * 4) = v.x;
      }
    }
    v.x += 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { stamps[((long)blockIdx.x * 8 + wave) * 2] = t1 - t0; stamps[((long)blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}

template <int MODE> void run(const char* name, int grid, char* out, long ld, unsigned long long* st_d, int reps) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 120 * 1024, 0, out, ld, st_d, reps);
  hipDeviceSynchronize();
  std::vector<unsigned long long> st(grid * 16);
  hipMemcpy(st.data(), st_d, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> a, b;
  for (int i = 0; i < grid * 8; ++i) { a.push_back((double)st[2 * i]); b.push_back((double)st[2 * i + 1]); }
  std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
  const double bytes = 256.0 * 640 * reps;
  printf("%-44s grid %3d: issue %7.0f cycles (%5.1f B/clk/CU), acknowledged %7.0f (%5.1f B/clk/CU)\n", name, grid, a[a.size() / 2], bytes / a[a.size() / 2],
         b[b.size() / 2], bytes / b[b.size() / 2]);
}

int main() {
  const long ld = 5120 * 2;                                // row stride of a [M, 5120] bf16 matrix
  char* out; hipMalloc(&out, 256L * 256 * ld);
  unsigned long long* st; hipMalloc(&st, 256 * 16 * 8);
  for (int grid : {1, 256}) {
    run<0>("16 B/lane, contiguous 1 KB per instr", grid, out, ld, st, 1);
    run<1>("16 B/lane, 16 rows x 64 B per instr", grid, out, ld, st, 1);
    run<2>("8 B/lane, 16 rows x 32 B per instr", grid, out, ld, st, 1);
    run<3>("4 B/lane, contiguous 256 B per instr", grid, out, ld, st, 1);
    run<4>("16 B/lane, contiguous, nontemporal", grid, out, ld, st, 1);
    run<5>("16 B/lane, 8 rows x 128 B aligned", grid, out, ld, st, 1);
    run<6>("16 B/lane, 8 rows x 128 B, +32 B misaligned", grid, out, ld, st, 1);
    run<7>("16 B/lane, 4 rows x 256 B (+ 128-B remainder)", grid, out, ld, st, 1);
    run<8>("16 B/lane, rows of 160 B (wave-local spans)", grid, out, ld, st, 1);
  }
  run<0>("16 B/lane contiguous, 4 tiles back to back", 256, out, ld, st, 4);
  return 0;
}
