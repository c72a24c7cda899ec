// Host-side helpers shared by the composite translation units (api.hip, generate.hip): workspace carving, the plain
// GEMM wrapper, the split-K heuristic.  Internal linkage on purpose (each TU gets its own copy).
#pragma once
#include "common.h"
#include "internal.h"
#include "../../include/ta355.h"

namespace {
struct Carver {
  char* base; size_t off;
  explicit Carver(void* b) : base((char*)b), off(0) {}
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
  size_t total() const { return (off + 255) & ~(size_t)255; }
};
inline int pad64(int x) { return (x + 63) / 64 * 64; }
inline int gemm(const void* A, const void* W, void* C, int M, int N, int K, const float* bias, const float* res, int act,
                int out_bf16, hipStream_t st) {
  return ta_gemm_bf16_nt(A, W, C, M, N, K, K, 0, 0, N, 0, 0, 0, bias, res, act, out_bf16, 1, nullptr, st);
}
inline int gemm_opt(const void* A, const void* W, void* C, int M, int N, int K, const float* bias, const float* res, int act,
                    int out_bf16, const ta_gemm_opts& o, hipStream_t st) {
  return ta_gemm_bf16_nt_opt(A, W, C, M, N, K, K, 0, 0, N, 0, 0, 0, bias, res, act, out_bf16, 1, nullptr, nullptr, nullptr, nullptr,
                             &o, st);
}
inline ta_gemm_opts opts_none() { return ta_gemm_opts{nullptr, nullptr, 0, 0, nullptr, nullptr, 0, 0, 0}; }
inline ta_gemm_opts opts_kext(const void* a2, const void* w2) { ta_gemm_opts o = opts_none(); o.a2 = a2; o.w2 = w2; o.k2 = 64; o.lda2 = 64; return o; }
// split-K heuristic: fill the 512 resident workgroup slots (256 CUs x 2) when the tile grid is small
inline int pick_splits(int M, int N, int K) {
  const long tiles = (long)ta_cdiv(M, 128) * ta_cdiv(N, 128);
  int s = 1;
  while (tiles * s < 512 && K / 64 / (s * 2) >= 8 && s < 64) s *= 2;
  return s;
}
#define RC(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)
// Adapter images kept OUTSIDE the training tape (generation packs them once at prefill and reuses them every decode
// step): per layer 4 groups x {s*Acat [64,in], its transpose, Bext [N,64], its transpose}, all bf16.
inline size_t lora_imgs_carve(const ta_lm_weights* w, void* base, ta_i_lora_layer_imgs* out) {
  Carver c(base);
  const int D = w->hidden, F = w->ffn, bq = w->heads * w->head_dim, NQKV = (w->heads + 2 * w->kv_heads) * w->head_dim;
  const int in[4] = {D, bq, D, F}, N[4] = {NQKV, D, 2 * F, D};
  for (int l = 0; l < w->n_layers; ++l)
    for (int g = 0; g < 4; ++g) {
      LoraImg im;
      im.a = c.take<bf16_t>((size_t)64 * in[g]); im.at = c.take<bf16_t>((size_t)64 * in[g]);
      im.b = c.take<bf16_t>((size_t)64 * N[g]); im.bt = c.take<bf16_t>((size_t)64 * N[g]);
      if (out) out[l].g[g] = im;
    }
  return c.total();
}
}  // namespace
