// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds its own element index; lane L reads at element address addr(L).
// Prints, per lane, the four 16-bit values it received, for two address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  int addr;
  if (mode == 0) addr = lane * 4;                                 // contiguous 8-B chunks
  else addr = (lane & 15) * 64 + (lane >> 4) * 4;                 // lane%16 -> row (stride 64 el), lane/16 -> 4-el column group
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + addr));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = r[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
