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

__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
  const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = bf2f((bf16_t)(u[k] & 0xffff)); f[2 * k + 1] = bf2f((bf16_t)(u[k] >> 16)); }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}
__device__ __forceinline__ void load8f(const float* p, float* f) {
  const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
// sum over the 8 lanes that share one (token, head) row
__device__ __forceinline__ float oct_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}

// ---------------------------------------------------------------------------- encoder
// One workgroup = 64 tokens x one (section, head); a thread moves 16-B chunks (8 dims).  RoPE pairs d <-> d+16 are
// chunk c <-> c^2 for c < 4 (the partner chunk is a second 16-B load of the same row); chunks 4..7 pass through.
__global__ __launch_bounds__(256) void enc_qkv_post_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ cosT,
                                                           const float* __restrict__ sinT, bf16_t* __restrict__ Qo,
                                                           bf16_t* __restrict__ Ko, bf16_t* __restrict__ VTo,
                                                           int H, int S, int Sp) {
  constexpr int HD = 64, ROT = 32, TS = HD + 4;
  __shared__ __attribute__((aligned(16))) bf16_t tile[TOK_TILE * TS];
  const int tid = threadIdx.x;
  const int l0 = blockIdx.x * TOK_TILE, hh = blockIdx.y, b = blockIdx.z;
  const int sec = hh / H, head = hh % H;     // 0 q, 1 k, 2 v
  const long ld = 3L * H * HD;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = tid + it * 256, tl = item >> 3, c = item & 7, l = l0 + tl;
    const bf16_t* src = qkv + ((long)b * S + l) * ld + (long)sec * H * HD + head * HD;
    if (sec == 2) {
      const uint4 v = l < S ? *(const uint4*)(src + c * 8) : make_uint4(0, 0, 0, 0);
      *(uint2*)(tile + tl * TS + c * 8) = make_uint2(v.x, v.y);            // rows are 136 B: 8-B aligned
      *(uint2*)(tile + tl * TS + c * 8 + 4) = make_uint2(v.z, v.w);
    } else if (l < S) {
      uint4 v = *(const uint4*)(src + c * 8);
      if (c < ROT / 8) {
        const uint4 o = *(const uint4*)(src + (c ^ 2) * 8);
        const float4* cp = (const float4*)(cosT + (long)l * (ROT / 2) + (c & 1) * 8);
        const float4* sp = (const float4*)(sinT + (long)l * (ROT / 2) + (c & 1) * 8);
        const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
        const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const uint32_t xv[4] = {v.x, v.y, v.z, v.w}, ov[4] = {o.x, o.y, o.z, o.w};
        uint32_t r[4];
        const float sgn = c < 2 ? -1.f : 1.f;                             // first half: x c - other s; second: x c + other s
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xa = bf2f((bf16_t)(xv[k] & 0xffff)), xb = bf2f((bf16_t)(xv[k] >> 16));
          const float oa = bf2f((bf16_t)(ov[k] & 0xffff)), ob = bf2f((bf16_t)(ov[k] >> 16));
          const float ya = xa * cs[2 * k] + sgn * (oa * sn[2 * k]);
          const float yb = xb * cs[2 * k + 1] + sgn * (ob * sn[2 * k + 1]);
          r[k] = pack2bf(ya, yb);
        }
        v = make_uint4(r[0], r[1], r[2], r[3]);
      }
      bf16_t* dst = (sec == 0 ? Qo : Ko) + (((long)b * H + head) * S + l) * HD;
      *(uint4*)(dst + c * 8) = v;
    }
  }
  if (sec == 2) {
    __syncthreads();
    store_transposed<HD>(tile, TS, VTo + ((long)b * H + head) * HD * Sp, Sp, l0, tid);
  }
}

// ---------------------------------------------------------------------------- LM forward
// qkv0 token-major [B*L, (Hq+2Hkv)*128] (pre-norm q | k | v).  Position of token (b,l) = pos ? pos[b*L+l] : l.
// One workgroup = 64 tokens x one (section, head).  Eight lanes share a row: lane j of the octet owns dims
// [8j, 8j+8) and [64+8j, 64+8j+8) -- two 16-B chunks that are each other's RoPE partners (pairs d <-> d+64).
__global__ __launch_bounds__(256) void lm_qkv_post_fwd_kernel(const bf16_t* __restrict__ qkv0, const float* __restrict__ qn_w,
                                                              const float* __restrict__ kn_w, const float* __restrict__ cosT,
                                                              const float* __restrict__ sinT, const int* __restrict__ pos,
                                                              bf16_t* __restrict__ Qo, bf16_t* __restrict__ Ko,
                                                              bf16_t* __restrict__ Vo, bf16_t* __restrict__ QTo,
                                                              bf16_t* __restrict__ KTo, bf16_t* __restrict__ VTo,
                                                              float* __restrict__ rq, float* __restrict__ rk,
                                                              int Hq, int Hkv, int L, int Lp, float eps) {
  constexpr int HD = 128, TS = HD + 4;
  __shared__ __attribute__((aligned(16))) bf16_t tile[TOK_TILE * TS];
  const int tid = threadIdx.x, j = tid & 7;
  const int l0 = blockIdx.x * TOK_TILE, hh = blockIdx.y, b = blockIdx.z;
  const int sec = hh < Hq ? 0 : (hh < Hq + Hkv ? 1 : 2);
  const int head = sec == 0 ? hh : (sec == 1 ? hh - Hq : hh - Hq - Hkv);
  const int Hs = sec == 0 ? Hq : Hkv;
  const long ld = (long)(Hq + 2 * Hkv) * HD;
  const long coff = (long)hh * HD;
  bf16_t* out = (sec == 0 ? Qo : (sec == 1 ? Ko : Vo)) + ((long)b * Hs + head) * L * HD;
  bf16_t* outT_base = sec == 0 ? QTo : (sec == 1 ? KTo : VTo);      // null: that transposed image is not wanted (Q^T / K^T since the
  bf16_t* outT = outT_base + ((long)b * Hs + head) * HD * Lp;       // backward reads them out of its row tiles; V^T feeds the forward)
  float w1[8], w2[8];
  if (sec < 2) { const float* nw = sec == 0 ? qn_w : kn_w; load8f(nw + 8 * j, w1); load8f(nw + 64 + 8 * j, w2); }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int tl = (tid >> 3) + it * 32, l = l0 + tl;
    uint4 v1 = make_uint4(0, 0, 0, 0), v2 = v1;
    if (l < L) {
      const bf16_t* src = qkv0 + ((long)b * L + l) * ld + coff;
      v1 = *(const uint4*)(src + 8 * j); v2 = *(const uint4*)(src + 64 + 8 * j);
    }
    if (sec < 2) {                                  // whole octets take the same branch; rows >= L carry zeros
      float x1[8], x2[8], c[8], sn[8];
      unpack8(v1, x1); unpack8(v2, x2);
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += x1[e] * x1[e] + x2[e] * x2[e];
      const float r = rsqrtf(oct_sum(ss) / (float)HD + eps);
      if (l < L) {
        if (j == 0) (sec == 0 ? rq : rk)[((long)b * L + l) * Hs + head] = r;
        const int p = pos ? pos[(long)b * L + l] : l;
        load8f(cosT + (long)p * 64 + 8 * j, c); load8f(sinT + (long)p * 64 + 8 * j, sn);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float n1 = x1[e] * r * w1[e], n2 = x2[e] * r * w2[e];
          x1[e] = n1 * c[e] - n2 * sn[e];
          x2[e] = n2 * c[e] + n1 * sn[e];
        }
        v1 = pack8(x1); v2 = pack8(x2);
      }
    }
    if (l < L) {
      *(uint4*)(out + (long)l * HD + 8 * j) = v1;
      *(uint4*)(out + (long)l * HD + 64 + 8 * j) = v2;
    }
    if (outT_base) {
      bf16_t* tr = tile + tl * TS + 8 * j;          // rows are 264 B: 8-B aligned
      *(uint2*)(tr) = make_uint2(v1.x, v1.y); *(uint2*)(tr + 4) = make_uint2(v1.z, v1.w);
      *(uint2*)(tr + 64) = make_uint2(v2.x, v2.y); *(uint2*)(tr + 68) = make_uint2(v2.z, v2.w);
    }
  }
  if (!outT_base) return;
  __syncthreads();
  store_transposed<HD>(tile, TS, outT, Lp, l0, tid);
}

// ---------------------------------------------------------------------------- LM backward
// dQ/dK/dV head-major -> dqkv token-major [B*L, (Hq+2Hkv)*128]; same octet-per-row mapping as the forward.
template <bool WGRAD>      // WGRAD: also accumulate the q_norm / k_norm weight gradients (trainable LM)
__global__ __launch_bounds__(256) void lm_qkv_post_bwd_kernel(const bf16_t* __restrict__ dQ, const bf16_t* __restrict__ dK,
                                                              const bf16_t* __restrict__ dV, const bf16_t* __restrict__ qkv0,
                                                              const float* __restrict__ rq, const float* __restrict__ rk,
                                                              const float* __restrict__ qn_w, const float* __restrict__ kn_w,
                                                              const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                              const int* __restrict__ pos, bf16_t* __restrict__ dqkv,
                                                              float* __restrict__ dqn, float* __restrict__ dkn,
                                                              int Hq, int Hkv, int L) {
  constexpr int HD = 128;
  __shared__ float wsum[4][HD];                      // per-wave partial q_norm / k_norm weight gradients (trainable LM)
  float gw1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gw2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int tid = threadIdx.x, j = tid & 7;
  const int l0 = blockIdx.x * TOK_TILE, hh = blockIdx.y, b = blockIdx.z;
  const int sec = hh < Hq ? 0 : (hh < Hq + Hkv ? 1 : 2);
  const int head = sec == 0 ? hh : (sec == 1 ? hh - Hq : hh - Hq - Hkv);
  const int Hs = sec == 0 ? Hq : Hkv;
  const long ld = (long)(Hq + 2 * Hkv) * HD;
  const long coff = (long)hh * HD;
  const bf16_t* din = (sec == 0 ? dQ : (sec == 1 ? dK : dV)) + ((long)b * Hs + head) * L * HD;
  float w1[8], w2[8];
  if (sec < 2) { const float* nw = sec == 0 ? qn_w : kn_w; load8f(nw + 8 * j, w1); load8f(nw + 64 + 8 * j, w2); }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int l = l0 + (tid >> 3) + it * 32;
    const bool live = l < L;
    const int lc = live ? l : L - 1;                // clamped rows compute (octet shuffles stay uniform) but never store
    uint4 g1 = *(const uint4*)(din + (long)lc * HD + 8 * j), g2 = *(const uint4*)(din + (long)lc * HD + 64 + 8 * j);
    if (sec < 2) {
      const bf16_t* src = qkv0 + ((long)b * L + lc) * ld + coff;
      float d1[8], d2[8], x1[8], x2[8], c[8], sn[8];
      unpack8(g1, d1); unpack8(g2, d2);
      unpack8(*(const uint4*)(src + 8 * j), x1); unpack8(*(const uint4*)(src + 64 + 8 * j), x2);
      const int p = pos ? pos[(long)b * L + lc] : lc;
      load8f(cosT + (long)p * 64 + 8 * j, c); load8f(sinT + (long)p * 64 + 8 * j, sn);
      const float r = (sec == 0 ? rq : rk)[((long)b * L + lc) * Hs + head];
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dn1 = d1[e] * c[e] + d2[e] * sn[e], dn2 = d2[e] * c[e] - d1[e] * sn[e];     // RoPE^T
        x1[e] *= r; x2[e] *= r;                                                             // x-hat
        if (WGRAD && live) { gw1[e] += dn1 * x1[e]; gw2[e] += dn2 * x2[e]; }                 // d loss / d norm weight
        d1[e] = dn1 * w1[e]; d2[e] = dn2 * w2[e];
        dot += d1[e] * x1[e] + d2[e] * x2[e];
      }
      const float md = oct_sum(dot) / (float)HD;
#pragma unroll
      for (int e = 0; e < 8; ++e) { d1[e] = r * (d1[e] - x1[e] * md); d2[e] = r * (d2[e] - x2[e] * md); }
      g1 = pack8(d1); g2 = pack8(d2);
    }
    if (live) {
      bf16_t* dst = dqkv + ((long)b * L + l) * ld + coff;
      *(uint4*)(dst + 8 * j) = g1;
      *(uint4*)(dst + 64 + 8 * j) = g2;
    }
  }
  float* dwn = sec == 0 ? dqn : (sec == 1 ? dkn : nullptr);        // block-uniform
  if (WGRAD && dwn) {
    // lanes with the same j (stride 8) hold the same 16 dims: fold them, then the 4 waves through LDS, one atomic per dim
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = gw1[e], c2 = gw2[e];
      a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      c2 += __shfl_xor(c2, 8, 64); c2 += __shfl_xor(c2, 16, 64); c2 += __shfl_xor(c2, 32, 64);
      if ((tid & 63) < 8) { wsum[tid >> 6][8 * j + e] = a; wsum[tid >> 6][64 + 8 * j + e] = c2; }
    }
    __syncthreads();
    if (tid < HD) unsafeAtomicAdd(dwn + tid, wsum[0][tid] + wsum[1][tid] + wsum[2][tid] + wsum[3][tid]);
  }
}

// ---------------------------------------------------------------------------- attention backward prep
// dO, O token-major [B*L, Hq*128] -> Delta [B,Hq,L], dOT [B,Hq,128,Lp]
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                            float* __restrict__ Delta, bf16_t* __restrict__ dOT,
                                                            int Hq, int L, int Lp) {
  constexpr int HD = 128, TS = HD + 4;
  __shared__ __attribute__((aligned(16))) bf16_t tile[TOK_TILE * TS];
  const int tid = threadIdx.x, j = tid & 7;
  const int l0 = blockIdx.x * TOK_TILE, h = blockIdx.y, b = blockIdx.z;
  const long ld = (long)Hq * HD;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int tl = (tid >> 3) + it * 32, l = l0 + tl;
    uint4 d1 = make_uint4(0, 0, 0, 0), d2 = d1, o1 = d1, o2 = d1;
    if (l < L) {
      const long off = ((long)b * L + l) * ld + (long)h * HD;
      d1 = *(const uint4*)(dO + off + 8 * j); d2 = *(const uint4*)(dO + off + 64 + 8 * j);
      o1 = *(const uint4*)(O + off + 8 * j); o2 = *(const uint4*)(O + off + 64 + 8 * j);
    }
    float a[8], c[8], e[8], f[8];
    unpack8(d1, a); unpack8(d2, c); unpack8(o1, e); unpack8(o2, f);
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) dot += a[q] * e[q] + c[q] * f[q];
    dot = oct_sum(dot);
    if (l < L && j == 0) Delta[((long)b * Hq + h) * L + l] = dot;
    if (dOT) {
      bf16_t* tr = tile + tl * TS + 8 * j;
      *(uint2*)(tr) = make_uint2(d1.x, d1.y); *(uint2*)(tr + 4) = make_uint2(d1.z, d1.w);
      *(uint2*)(tr + 64) = make_uint2(d2.x, d2.y); *(uint2*)(tr + 68) = make_uint2(d2.z, d2.w);
    }
  }
  if (!dOT) return;                                  // the backward reads dO^T out of its row tiles (ds_read_b64_tr_b16): no image needed
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
                                  const float* sinT, const int* pos, void* dqkv, float* dqn_accum, float* dkn_accum,
                                  int B, int Hq, int Hkv, int L, hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  if (dqn_accum || dkn_accum)
    TA_LAUNCH(lm_qkv_post_bwd_kernel<true>, dim3(ta_cdiv(L, 64), Hq + 2 * Hkv, B), dim3(256), 0, st, (const bf16_t*)dQ,
              (const bf16_t*)dK, (const bf16_t*)dV, (const bf16_t*)qkv0, rq, rk, qn_w, kn_w, cosT, sinT, pos,
              (bf16_t*)dqkv, dqn_accum, dkn_accum, Hq, Hkv, L);
  else
    TA_LAUNCH(lm_qkv_post_bwd_kernel<false>, dim3(ta_cdiv(L, 64), Hq + 2 * Hkv, B), dim3(256), 0, st, (const bf16_t*)dQ,
              (const bf16_t*)dK, (const bf16_t*)dV, (const bf16_t*)qkv0, rq, rk, qn_w, kn_w, cosT, sinT, pos,
              (bf16_t*)dqkv, dqn_accum, dkn_accum, Hq, Hkv, L);
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
