// ta355 per-head post-processing around attention (HBM-bound, one wave per token x head).
//
//   enc_qkv_post_kernel   GLM-ASR: partial RoPE (first 32 of 64 dims, rotate-half pairs d <-> d+16,
//                         theta 1e4) on q,k; emits head-major Q,K and the transposed V^T image
//                         (TF:models/glmasr/modeling_glmasr.py:153-168,192-198)
//   lm_qkv_post_fwd_kernel  Qwen3: per-head RMSNorm (q_norm / k_norm over head_dim 128) then full RoPE
//                         (pairs d <-> d+64, theta 1e6); emits head-major Q,K,V plus Q^T,K^T,V^T images and
//                         the per-(token,head) rstd for backward (TF:models/qwen3/modeling_qwen3.py:251-256)
//   lm_qkv_post_bwd_kernel  the backward of that (RoPE^T, RMSNorm backward, frozen norm weights)
//   attn_bwd_prep_kernel  Delta = rowsum(dO o O) and the dO^T image for the dK/dV kernel
//
// Transposed images [B, heads, HD, Lp] (Lp = L rounded up to 64, pad columns ZERO) are written
// through an LDS tile so that global stores stay 16 B per lane.
#include "common.h"

#define TOK_TILE 64

// LDS tile [64 tokens][HD] bf16 with padded stride; rows of tokens >= L must be zero.
template <int HD>
__device__ __forceinline__ void store_transposed(const bf16_t* tile, int tstride, bf16_t* outT /* [HD][Lp] */, int Lp,
                                                 int l0, int tid) {
  // HD rows x 8 chunks (8 tokens = 16 B each)
#pragma unroll
  for (int i = 0; i < HD / 32; ++i) {
    const int c = tid + i * 256;
    const int d = c >> 3, tc = c & 7;
    union { uint4 v; bf16_t e[8]; } u;
#pragma unroll
    for (int j = 0; j < 8; ++j) u.e[j] = tile[(tc * 8 + j) * tstride + d];
    *(uint4*)(outT + (long)d * Lp + l0 + tc * 8) = u.v;
  }
}

// ---------------------------------------------------------------------------- encoder
__global__ __launch_bounds__(256) void enc_qkv_post_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ cosT,
                                                           const float* __restrict__ sinT, bf16_t* __restrict__ Qo,
                                                           bf16_t* __restrict__ Ko, bf16_t* __restrict__ VTo,
                                                           int H, int S, int Sp) {
  constexpr int HD = 64, ROT = 32, TS = HD + 4;
  __shared__ bf16_t tile[TOK_TILE * TS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l0 = blockIdx.x * TOK_TILE, hh = blockIdx.y, b = blockIdx.z;
  const int sec = hh / H, head = hh % H;     // 0 q, 1 k, 2 v
  const long ld = 3L * H * HD;
  for (int i = 0; i < 16; ++i) {
    const int tl = wave * 16 + i, l = l0 + tl;
    float y = 0.f;
    if (l < S) {
      const float x = bf2f(qkv[((long)b * S + l) * ld + (long)sec * H * HD + head * HD + lane]);
      y = x;
      if (sec < 2) {
        const float other = __shfl_xor(x, 16, 64);
        if (lane < ROT) {
          const int fi = lane & 15;
          const float c = cosT[l * (ROT / 2) + fi], s = sinT[l * (ROT / 2) + fi];
          y = (lane < 16) ? (x * c - other * s) : (x * c + other * s);
        }
        bf16_t* dst = (sec == 0 ? Qo : Ko) + (((long)b * H + head) * S + l) * HD;
        dst[lane] = f2bf(y);
      }
    } else if (sec < 2) {
      (void)__shfl_xor(y, 16, 64);
    }
    if (sec == 2) tile[tl * TS + lane] = f2bf(y);
  }
  if (sec == 2) {
    __syncthreads();
    store_transposed<HD>(tile, TS, VTo + ((long)b * H + head) * HD * Sp, Sp, l0, tid);
  }
}

// ---------------------------------------------------------------------------- LM forward
// qkv0 token-major [B*L, (Hq+2Hkv)*128] (pre-norm q | k | v).  Position of token (b,l) = pos ? pos[b*L+l] : l.
__global__ __launch_bounds__(256) void lm_qkv_post_fwd_kernel(const bf16_t* __restrict__ qkv0, const float* __restrict__ qn_w,
                                                              const float* __restrict__ kn_w, const float* __restrict__ cosT,
                                                              const float* __restrict__ sinT, const int* __restrict__ pos,
                                                              bf16_t* __restrict__ Qo, bf16_t* __restrict__ Ko,
                                                              bf16_t* __restrict__ Vo, bf16_t* __restrict__ QTo,
                                                              bf16_t* __restrict__ KTo, bf16_t* __restrict__ VTo,
                                                              float* __restrict__ rq, float* __restrict__ rk,
                                                              int Hq, int Hkv, int L, int Lp, float eps) {
  constexpr int HD = 128, TS = HD + 4;
  __shared__ bf16_t tile[TOK_TILE * TS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l0 = blockIdx.x * TOK_TILE, hh = blockIdx.y, b = blockIdx.z;
  const int sec = hh < Hq ? 0 : (hh < Hq + Hkv ? 1 : 2);
  const int head = sec == 0 ? hh : (sec == 1 ? hh - Hq : hh - Hq - Hkv);
  const int Hs = sec == 0 ? Hq : Hkv;
  const long ld = (long)(Hq + 2 * Hkv) * HD;
  const long coff = (long)hh * HD;
  bf16_t* out = (sec == 0 ? Qo : (sec == 1 ? Ko : Vo)) + ((long)b * Hs + head) * L * HD;
  bf16_t* outT = (sec == 0 ? QTo : (sec == 1 ? KTo : VTo)) + ((long)b * Hs + head) * HD * Lp;
  const float* nw = sec == 0 ? qn_w : kn_w;
  const float w1 = sec < 2 ? nw[lane] : 1.f, w2 = sec < 2 ? nw[lane + 64] : 1.f;
  for (int i = 0; i < 16; ++i) {
    const int tl = wave * 16 + i, l = l0 + tl;
    float y1 = 0.f, y2 = 0.f;
    if (l < L) {
      const bf16_t* src = qkv0 + ((long)b * L + l) * ld + coff;
      const float x1 = bf2f(src[lane]), x2 = bf2f(src[lane + 64]);
      if (sec < 2) {
        const float r = rsqrtf(wave_sum(x1 * x1 + x2 * x2) / (float)HD + eps);
        if (lane == 0) (sec == 0 ? rq : rk)[((long)b * L + l) * Hs + head] = r;
        const float n1 = x1 * r * w1, n2 = x2 * r * w2;
        const int p = pos ? pos[(long)b * L + l] : l;
        const float c = cosT[(long)p * 64 + lane], s = sinT[(long)p * 64 + lane];
        y1 = n1 * c - n2 * s;
        y2 = n2 * c + n1 * s;
      } else { y1 = x1; y2 = x2; }
      out[(long)l * HD + lane] = f2bf(y1);
      out[(long)l * HD + lane + 64] = f2bf(y2);
    }
    tile[tl * TS + lane] = f2bf(y1);
    tile[tl * TS + lane + 64] = f2bf(y2);
  }
  __syncthreads();
  store_transposed<HD>(tile, TS, outT, Lp, l0, tid);
}

// ---------------------------------------------------------------------------- LM backward
// dQ/dK/dV head-major -> dqkv token-major [B*L, (Hq+2Hkv)*128]
__global__ __launch_bounds__(256) void lm_qkv_post_bwd_kernel(const bf16_t* __restrict__ dQ, const bf16_t* __restrict__ dK,
                                                              const bf16_t* __restrict__ dV, const bf16_t* __restrict__ qkv0,
                                                              const float* __restrict__ rq, const float* __restrict__ rk,
                                                              const float* __restrict__ qn_w, const float* __restrict__ kn_w,
                                                              const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                              const int* __restrict__ pos, bf16_t* __restrict__ dqkv,
                                                              int Hq, int Hkv, int L) {
  constexpr int HD = 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l0 = blockIdx.x * TOK_TILE, hh = blockIdx.y, b = blockIdx.z;
  const int sec = hh < Hq ? 0 : (hh < Hq + Hkv ? 1 : 2);
  const int head = sec == 0 ? hh : (sec == 1 ? hh - Hq : hh - Hq - Hkv);
  const int Hs = sec == 0 ? Hq : Hkv;
  const long ld = (long)(Hq + 2 * Hkv) * HD;
  const long coff = (long)hh * HD;
  const bf16_t* din = (sec == 0 ? dQ : (sec == 1 ? dK : dV)) + ((long)b * Hs + head) * L * HD;
  const float* nw = sec == 0 ? qn_w : kn_w;
  const float w1 = sec < 2 ? nw[lane] : 1.f, w2 = sec < 2 ? nw[lane + 64] : 1.f;
  for (int i = 0; i < 16; ++i) {
    const int l = l0 + wave * 16 + i;
    if (l >= L) continue;
    const float g1 = bf2f(din[(long)l * HD + lane]), g2 = bf2f(din[(long)l * HD + lane + 64]);
    float o1 = g1, o2 = g2;
    if (sec < 2) {
      const int p = pos ? pos[(long)b * L + l] : l;
      const float c = cosT[(long)p * 64 + lane], s = sinT[(long)p * 64 + lane];
      const float dn1 = g1 * c + g2 * s, dn2 = g2 * c - g1 * s;          // RoPE^T
      const bf16_t* src = qkv0 + ((long)b * L + l) * ld + coff;
      const float r = (sec == 0 ? rq : rk)[((long)b * L + l) * Hs + head];
      const float xh1 = bf2f(src[lane]) * r, xh2 = bf2f(src[lane + 64]) * r;
      const float a1 = dn1 * w1, a2 = dn2 * w2;
      const float md = wave_sum(a1 * xh1 + a2 * xh2) / (float)HD;
      o1 = r * (a1 - xh1 * md);
      o2 = r * (a2 - xh2 * md);
    }
    bf16_t* dst = dqkv + ((long)b * L + l) * ld + coff;
    dst[lane] = f2bf(o1);
    dst[lane + 64] = f2bf(o2);
  }
}

// ---------------------------------------------------------------------------- attention backward prep
// dO, O token-major [B*L, Hq*128] -> Delta [B,Hq,L], dOT [B,Hq,128,Lp]
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                            float* __restrict__ Delta, bf16_t* __restrict__ dOT,
                                                            int Hq, int L, int Lp) {
  constexpr int HD = 128, TS = HD + 4;
  __shared__ bf16_t tile[TOK_TILE * TS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l0 = blockIdx.x * TOK_TILE, h = blockIdx.y, b = blockIdx.z;
  const long ld = (long)Hq * HD;
  for (int i = 0; i < 16; ++i) {
    const int tl = wave * 16 + i, l = l0 + tl;
    bf16_t d1 = 0, d2 = 0;
    if (l < L) {
      const long off = ((long)b * L + l) * ld + (long)h * HD;
      d1 = dO[off + lane]; d2 = dO[off + lane + 64];
      const float dl = wave_sum(bf2f(d1) * bf2f(O[off + lane]) + bf2f(d2) * bf2f(O[off + lane + 64]));
      if (lane == 0) Delta[((long)b * Hq + h) * L + l] = dl;
    }
    tile[tl * TS + lane] = d1;
    tile[tl * TS + lane + 64] = d2;
  }
  __syncthreads();
  store_transposed<HD>(tile, TS, dOT + ((long)b * Hq + h) * HD * Lp, Lp, l0, tid);
}

// ----------------------------------------------------------------------------- C-ABI
extern "C" int ta_enc_qkv_post(const void* qkv, const float* cosT, const float* sinT, void* Q, void* K, void* VT,
                               int B, int H, int S, int Sp, hipStream_t st) {
  if (B <= 0 || S <= 0) return TA_OK;
  if (Sp % 64 || Sp < S) return TA_ERR_ARG;
  TA_LAUNCH(enc_qkv_post_kernel, dim3(Sp / 64, 3 * H, B), dim3(256), 0, st, (const bf16_t*)qkv, cosT, sinT,
                     (bf16_t*)Q, (bf16_t*)K, (bf16_t*)VT, H, S, Sp);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_lm_qkv_post_fwd(const void* qkv0, const float* qn_w, const float* kn_w, const float* cosT,
                                  const float* sinT, const int* pos, void* Q, void* K, void* V, void* QT, void* KT,
                                  void* VT, float* rq, float* rk, int B, int Hq, int Hkv, int L, int Lp, float eps,
                                  hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  if (Lp % 64 || Lp < L) return TA_ERR_ARG;
  TA_LAUNCH(lm_qkv_post_fwd_kernel, dim3(Lp / 64, Hq + 2 * Hkv, B), dim3(256), 0, st, (const bf16_t*)qkv0, qn_w,
                     kn_w, cosT, sinT, pos, (bf16_t*)Q, (bf16_t*)K, (bf16_t*)V, (bf16_t*)QT, (bf16_t*)KT, (bf16_t*)VT, rq,
                     rk, Hq, Hkv, L, Lp, eps);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_lm_qkv_post_bwd(const void* dQ, const void* dK, const void* dV, const void* qkv0, const float* rq,
                                  const float* rk, const float* qn_w, const float* kn_w, const float* cosT,
                                  const float* sinT, const int* pos, void* dqkv, int B, int Hq, int Hkv, int L,
                                  hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  TA_LAUNCH(lm_qkv_post_bwd_kernel, dim3(ta_cdiv(L, 64), Hq + 2 * Hkv, B), dim3(256), 0, st, (const bf16_t*)dQ,
                     (const bf16_t*)dK, (const bf16_t*)dV, (const bf16_t*)qkv0, rq, rk, qn_w, kn_w, cosT, sinT, pos,
                     (bf16_t*)dqkv, Hq, Hkv, L);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_attn_bwd_prep(const void* dO, const void* O, float* Delta, void* dOT, int B, int Hq, int L, int Lp,
                                hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  if (Lp % 64 || Lp < L) return TA_ERR_ARG;
  TA_LAUNCH(attn_bwd_prep_kernel, dim3(Lp / 64, Hq, B), dim3(256), 0, st, (const bf16_t*)dO, (const bf16_t*)O,
                     Delta, (bf16_t*)dOT, Hq, L, Lp);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
