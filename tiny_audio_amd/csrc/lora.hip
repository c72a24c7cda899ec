// ta355 LoRA helper kernels (stage-2 training, BASELINE configs[4]; peft LoraLayer semantics wired at
// tiny_audio/asr_modeling.py:289-301: y = W x + (alpha/r) B (A x), r = 8, alpha = 32, targets q,k,v,o,gate,up,down).
//
// Adapters of the linears that share an input are fused per GROUP g in {qkv, o, gate|up, down}:
//   Acat_g [64, in_g]  rows [j*r, (j+1)*r) = A of member j, remaining rows zero (64 = one GEMM K-tile)
//   Bext_g [N_g, 64]   rows of member j carry B_j in columns [j*r, (j+1)*r), zero elsewhere
// so the adapted linear is ONE extra K-tile of the frozen GEMM:  y = [x | xa] [W | Bext]^T  with  xa = x (s Acat)^T.
#include "common.h"
#include "internal.h"

// The fp32 MASTERS keep the exact peft parameter count: la_g [members*r, in_g] (member A's stacked on rows) and
// lb_g [N_g, r] (member B's stacked on rows).  The 64-wide bf16 images are rebuilt from them every forward.

// A master fp32 [R, in] -> s*A image bf16 [64, in] (rows >= R zero) and its transpose [in, 64]
__global__ __launch_bounds__(256) void lora_pack_a_kernel(const float* __restrict__ in, float scale, bf16_t* __restrict__ out,
                                                          bf16_t* __restrict__ outT, int R, int Cn) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)64 * Cn) return;
  const int j = (int)(idx / Cn), c = (int)(idx % Cn);
  const bf16_t v = j < R ? f2bf(in[idx] * scale) : (bf16_t)0;
  out[idx] = v;
  outT[(long)c * 64 + j] = v;
}
// B master fp32 [N, r] -> Bext image bf16 [N, 64] (row n of member j owns columns [j*r, (j+1)*r)) and transpose [64, N]
__global__ __launch_bounds__(256) void lora_pack_b_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                          bf16_t* __restrict__ outT, int N, int r, int b0, int b1) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * 64) return;
  const int n = (int)(idx >> 6), j = (int)(idx & 63);
  const int jj = j - (n < b0 ? 0 : (n < b1 ? 1 : 2)) * r;
  const bf16_t v = (jj >= 0 && jj < r) ? f2bf(in[(long)n * r + jj]) : (bf16_t)0;
  out[idx] = v;
  outT[(long)j * N + n] = v;
}

// out[c*so_c + j*so_j] (+)= post * sum_m X[m, c] * Y[m, j]   for j < R (R <= 24), X bf16 [M, C], Y bf16 [M, ldy].
// Skinny "TN" product for the adapter gradients: dB = dY^T xa and dA = (dY B)^T x.  One thread per column c, the
// M range is cut into gridDim.y chunks combined with atomics (out must be zeroed).  Only columns j in
// [jlo(c), jlo(c) + r) are kept when r > 0 (member row boundaries b0, b1: block structure of Bext) and land in
// out[c*so_c + (j - jlo)*so_j], i.e. directly in the [N, r] master layout.
__global__ __launch_bounds__(256) void lora_skinny_tn_kernel(const bf16_t* __restrict__ X, int Cn, const bf16_t* __restrict__ Y,
                                                             int ldy, int R, float* __restrict__ out, long so_c, long so_j,
                                                             int M, float post, int r, int b0, int b1) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int per = (M + gridDim.y - 1) / gridDim.y;
  const int m0 = blockIdx.y * per, m1 = min(M, m0 + per);
  if (c >= Cn) return;
  float acc[24];
#pragma unroll
  for (int j = 0; j < 24; ++j) acc[j] = 0.f;
  for (int m = m0; m < m1; ++m) {
    const float x = bf2f(X[(long)m * Cn + c]);
    const uint4* yr = (const uint4*)(Y + (long)m * ldy);       // same address for the whole wave: broadcast loads
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (q * 8 < R) {
        const uint4 v = yr[q];
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[q * 8 + 2 * k] = fmaf(x, bf2f(u[k] & 0xffff), acc[q * 8 + 2 * k]);
          acc[q * 8 + 2 * k + 1] = fmaf(x, bf2f(u[k] >> 16), acc[q * 8 + 2 * k + 1]);
        }
      }
    }
  }
  int jlo = 0, jhi = R;
  if (r > 0) { const int mem = c < b0 ? 0 : (c < b1 ? 1 : 2); jlo = mem * r; jhi = jlo + r; }
#pragma unroll
  for (int j = 0; j < 24; ++j)
    if (j >= jlo && j < jhi) atomicAdd(out + (long)c * so_c + (long)(j - jlo) * so_j, acc[j] * post);
}

int ta_i_lora_pack_a(const float* in, float scale, void* out, void* outT, int R, int Cn, hipStream_t st) {
  TA_LAUNCH(lora_pack_a_kernel, dim3(ta_cdiv(64 * Cn, 256)), dim3(256), 0, st, in, scale, (bf16_t*)out, (bf16_t*)outT, R, Cn);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
int ta_i_lora_pack_b(const float* in, void* out, void* outT, int N, int r, int b0, int b1, hipStream_t st) {
  TA_LAUNCH(lora_pack_b_kernel, dim3(ta_cdiv(N * 64, 256)), dim3(256), 0, st, in, (bf16_t*)out, (bf16_t*)outT, N, r, b0, b1);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
int ta_i_lora_skinny_tn(const void* X, int Cn, const void* Y, int ldy, int R, float* out, long so_c, long so_j, int M, float post,
                        int r, int b0, int b1, hipStream_t st) {
  if (R > 24 || R % 8) return TA_ERR_ARG;
  int chunks = M / 256; if (chunks < 1) chunks = 1; if (chunks > 32) chunks = 32;
  TA_LAUNCH(lora_skinny_tn_kernel, dim3(ta_cdiv(Cn, 256), chunks), dim3(256), 0, st, (const bf16_t*)X, Cn, (const bf16_t*)Y, ldy, R,
            out, so_c, so_j, M, post, r, b0, b1);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
