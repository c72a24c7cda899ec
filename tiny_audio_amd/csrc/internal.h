// Internal (non-exported) helpers shared between translation units of libta355.so.
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

// bf16 images of one adapter group: s*Acat [64,in], its transpose [in,64], Bext [N,64], its transpose [64,N]
// a: s*Acat [64, in] and bt: Bext^T [64, N], both in k-step-major blocks [K/32][64][32] (W operands of ta_i_lora_skinny_nt);
// at [in, 64] and b [N, 64] row-major (K extensions of the frozen GEMMs)
struct LoraImg { bf16_t *a, *at, *b, *bt; };
struct ta_i_lora_layer_imgs { LoraImg g[4]; };   // qkv, o, gate|up, down

// `layers` images in one launch when masters / images of consecutive layers are a constant stride apart (else layers = 1)
int ta_i_lora_pack_a(const float* in, float scale, void* out, void* outT, int R, int Cn, int layers, long in_ls, long out_ls,
                     hipStream_t st);
int ta_i_lora_pack_b(const float* in, void* out, void* outT, int N, int r, int b0, int b1, int layers, long in_ls, long out_ls,
                     hipStream_t st);
int ta_i_lora_skinny_tn(const void* X, int Cn, const void* Y, int ldy, int R, float* out, long so_c, long so_j, int M, float post,
                        int r, int b0, int b1, hipStream_t st);
int ta_i_lora_skinny_tn2(const void* X0, int Cn0, const void* Y0, int R0, float* out0, long so_c0, long so_j0, float post0, int r0,
                         int b00, int b10, const void* X1, int Cn1, const void* Y1, int R1, float* out1, long so_c1, long so_j1,
                         float post1, int r1, int b01, int b11, int M, float* part0, long part_cs0, float* part1, long part_cs1,
                         hipStream_t st);   // two TN products, one launch; part != NULL: row chunk c stores at part + c * part_cs (no atomics)
int ta_i_lora_tn2_rows(int M, int Cn0, int Cn1);   // rows per chunk of such a launch (chunks = ceil(M / rows))
// the second level of the adapter-gradient reduction: 8 gradient kinds x layers in one launch
struct LoraReduceDesc { float* out[8]; long out_ls[8]; long size[8]; long part_off[8]; int chunks[8]; const float* part; long part_ls; };
int ta_i_lora_reduce_parts(const LoraReduceDesc& d, int layers, hipStream_t st);
int ta_i_lora_skinny_nt(const void* X, int K, const void* W, void* out, int M, int R, hipStream_t st);   // out[M,64] = X[M,K] W[64,K]^T; rows >= R of W are zero

// fused decode-step kernels for batch <= 32 (csrc/decode_fused.hip); TA_ERR_ARG = outside the envelope, take the unfused path
bool ta_i_dec_fused_serves(int B, int D, int F, int bq, int Hq, int Hkv, int Lmax);
// what the NEXT kernel of the step streams, fetched one kernel early by `wgs` extra workgroups (0 = 96): a contiguous range
// [p0, p0 + bytes) (p1 = null), or the first *slot_p * 256 bytes of `rows` rows of p0 and p1, row_stride bytes apart (K / V cache)
struct ta_i_dec_prefetch { const void* p0; const void* p1; long bytes; const int* slot_p; long row_stride; int rows; int wgs; };
int ta_i_dec_norm_linear(const float* x, const float* lnw, float eps, const void* W, void* out, int M, int N, int K, bool swiglu,
                         const ta_i_dec_prefetch* next, hipStream_t st);   // out = bf16(RMSNorm(x) W^T)  |  swiglu: W = [gate | up] rows, N = F, out = silu(gate) * up
int ta_i_dec_linear_res(const void* x, const void* W, float* out, const float* res, int M, int N, int K, const ta_i_dec_prefetch* next,
                        hipStream_t st);   // out = x W^T + res (f32)
int ta_i_dec_attn(const void* qkv0, const float* qn_w, const float* kn_w, const float* cosT, const float* sinT, const int* pos,
                  const int* slot_dev, const int* kmask, void* kc, void* vc, void* out, int B, int Hq, int Hkv, int Lmax, float eps,
                  float scale, const ta_i_dec_prefetch* next, hipStream_t st);   // q/k norm + rope + cache append + attention of the new position
// The fused step keeps its activations (the f32 stream, attention output, SwiGLU output) in the BLOCKED layout of decode_fused.hip
// ([K/32][32 rows][32]; buffers hold 32 rows whatever B is): x / res / out of the calls above are in that layout.
int ta_i_dec_embed(const long* ids, const float* emb, float* xblk, int B, int D, long vocab, hipStream_t st);
int ta_i_dec_final_norm(const float* xblk, const float* w, void* y, int B, int D, float eps, hipStream_t st);
