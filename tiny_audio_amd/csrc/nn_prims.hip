// ta355 primitives for the TRAINABLE transformer-style projectors (QFormer, MOSA):
// everything around their GEMMs that the frozen encoder / LM never needed -- GELU with a saved pre-activation,
// bias gradients (column sums), LayerNorm with residual + dropout mask and its backward (affine gradients), and the
// tiny windowed attention of the QFormer (3 queries x 3 or 15 keys per window, head_dim 80) forward and backward.
// All of them are HBM- or latency-bound; the flops live in gemm.hip.
//   reference: TF:models/blip_2/modeling_blip_2.py (Blip2QFormerMultiHeadAttention / SelfOutput / Intermediate / Output),
//              tiny_audio/projectors.py:88-182, 359-475
#include "common.h"
#include "../../include/ta355.h"

namespace {
// ---------------------------------------------------------------------------- GELU (exact erf form, nn.GELU default)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ a, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const uint4 v = ((const uint4*)h)[i];
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = pack2bf(gelu_erf(bf2f((bf16_t)(u[k] & 0xffff))), gelu_erf(bf2f((bf16_t)(u[k] >> 16))));
    ((uint4*)a)[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ da, const bf16_t* __restrict__ h,
                                                       bf16_t* __restrict__ dh, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const uint4 v = ((const uint4*)h)[i], g = ((const uint4*)da)[i];
    const uint32_t u[4] = {v.x, v.y, v.z, v.w}, w[4] = {g.x, g.y, g.z, g.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = pack2bf(bf2f((bf16_t)(w[k] & 0xffff)) * gelu_erf_grad(bf2f((bf16_t)(u[k] & 0xffff))),
                     bf2f((bf16_t)(w[k] >> 16)) * gelu_erf_grad(bf2f((bf16_t)(u[k] >> 16))));
    ((uint4*)dh)[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ---------------------------------------------------------------------------- ReLU (MOSA router)
__global__ __launch_bounds__(256) void relu_fwd_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ a, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) a[i] = (h[i] & 0x8000) ? (bf16_t)0 : h[i];
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const bf16_t* __restrict__ da, const bf16_t* __restrict__ h,
                                                       bf16_t* __restrict__ dh, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    dh[i] = (bf2f(h[i]) > 0.f) ? da[i] : (bf16_t)0;
}

// ---------------------------------------------------------------------------- dense mixture (MOSA): softmax gate + weighted sum
// rw = softmax(logits) over E <= 16 experts; out[m,:] = sum_e rw[m,e] * o[e][m,:]        (tiny_audio/projectors.py:158-166)
__global__ __launch_bounds__(256) void mix_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ o,
                                                      float* __restrict__ rw, float* __restrict__ out, int M, int D, int E) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float p[16], mx = -INFINITY, sum = 0.f;
  for (int e = 0; e < E; ++e) { p[e] = logits[(long)row * E + e]; mx = fmaxf(mx, p[e]); }
  for (int e = 0; e < E; ++e) { p[e] = __expf(p[e] - mx); sum += p[e]; }
  for (int e = 0; e < E; ++e) p[e] /= sum;
  if (lane < E) rw[(long)row * E + lane] = p[lane];
  for (int c = lane; c < D / 4; c += 64) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = 0; e < E; ++e) {
      const float4 v = ((const float4*)(o + ((long)e * M + row) * D))[c];
      acc.x += p[e] * v.x; acc.y += p[e] * v.y; acc.z += p[e] * v.z; acc.w += p[e] * v.w;
    }
    ((float4*)(out + (long)row * D))[c] = acc;
  }
}
// do[e][m,:] = dout[m,:] * rw[m,e] (bf16, feeds the expert GEMMs); drw[e] = <dout[m,:], o[e][m,:]>;
// dlogits = rw * (drw - sum_e drw rw)   (softmax backward)
__global__ __launch_bounds__(256) void mix_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ o,
                                                      const float* __restrict__ rw, bf16_t* __restrict__ dob,
                                                      float* __restrict__ dlogits, int M, int D, int E) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float p[16], dr[16];
  for (int e = 0; e < E; ++e) { p[e] = rw[(long)row * E + e]; dr[e] = 0.f; }
  for (int c = lane; c < D / 4; c += 64) {
    const float4 g = ((const float4*)(dout + (long)row * D))[c];
    for (int e = 0; e < E; ++e) {
      const float4 v = ((const float4*)(o + ((long)e * M + row) * D))[c];
      dr[e] += g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w;
      uint2 q; q.x = pack2bf(g.x * p[e], g.y * p[e]); q.y = pack2bf(g.z * p[e], g.w * p[e]);
      ((uint2*)(dob + ((long)e * M + row) * D))[c] = q;
    }
  }
  float dot = 0.f;
  for (int e = 0; e < E; ++e) { dr[e] = wave_sum(dr[e]); dot += dr[e] * p[e]; }
  if (lane < E) dlogits[(long)row * E + lane] = p[lane] * (dr[lane] - dot);
}

// ---------------------------------------------------------------------------- column sums (bias gradients)
// out[c] += sum_r x[r, c]; grid (ceil(C/256), row chunks); out must be zeroed
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int R, int C, float* __restrict__ out, int rows_per) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) {
    if constexpr (sizeof(T) == 4) s += ((const float*)x)[(long)r * C + c]; else s += bf2f(((const bf16_t*)x)[(long)r * C + c]);
  }
  unsafeAtomicAdd(out + c, s);
}

// ---------------------------------------------------------------------------- LayerNorm(z * keep + res)
// One wave per row, H <= 64 * 4 * MAXV.  Saves xhat and rstd for the backward.
template <int MAXV>
__global__ __launch_bounds__(256) void ln_res_fwd_kernel(const float* __restrict__ z, const float* __restrict__ keep,
                                                         const float* __restrict__ res, long res_rows,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, float* __restrict__ xhat, float* __restrict__ rstd_o,
                                                         float* __restrict__ yf, bf16_t* __restrict__ yb, int M, int H) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63, nv = H / 4;
  float4 u[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      float4 v = ((const float4*)(z + (long)row * H))[c];
      if (keep) { const float4 k = ((const float4*)(keep + (long)row * H))[c]; v.x *= k.x; v.y *= k.y; v.z *= k.z; v.w *= k.w; }
      if (res) { const float4 r = ((const float4*)(res + (long)(row % res_rows) * H))[c]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
      u[i] = v;
      s += v.x + v.y + v.z + v.w;
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
    if (lane + i * 64 < nv) {
      const float a = u[i].x - mean, b = u[i].y - mean, c = u[i].z - mean, d = u[i].w - mean;
      q += a * a + b * b + c * c + d * d;
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
  if (lane == 0) rstd_o[row] = rstd;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 g = ((const float4*)gamma)[c], b = ((const float4*)beta)[c];
      const float4 xh = make_float4((u[i].x - mean) * rstd, (u[i].y - mean) * rstd, (u[i].z - mean) * rstd, (u[i].w - mean) * rstd);
      ((float4*)(xhat + (long)row * H))[c] = xh;
      const float4 y = make_float4(xh.x * g.x + b.x, xh.y * g.y + b.y, xh.z * g.z + b.z, xh.w * g.w + b.w);
      if (yf) ((float4*)(yf + (long)row * H))[c] = y;
      if (yb) { uint2 o; o.x = pack2bf(y.x, y.y); o.y = pack2bf(y.z, y.w); ((uint2*)(yb + (long)row * H))[c] = o; }
    }
  }
}

// du = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;  dz = du * keep (bf16, for the GEMMs);
// dgamma += sum_rows dy * xhat, dbeta += sum_rows dy (LDS partials per block, then one atomic per column).
template <int MAXV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ keep, float* __restrict__ du,
                                                     bf16_t* __restrict__ dzb, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int M, int H, int rows_per_block) {
  extern __shared__ float part[];                    // [2 * H]
  for (int i = threadIdx.x; i < 2 * H; i += 256) part[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nv = H / 4;
  const int r_begin = blockIdx.x * rows_per_block, r_end = min(M, r_begin + rows_per_block);
  float4 ag[MAXV], ab[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) { ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = ag[i]; }
  for (int row = r_begin + wave; row < r_end; row += 4) {
    float4 g[MAXV], xh[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        const float4 d = ((const float4*)(dy + (long)row * H))[c], w = ((const float4*)gamma)[c];
        xh[i] = ((const float4*)(xhat + (long)row * H))[c];
        g[i] = make_float4(d.x * w.x, d.y * w.y, d.z * w.z, d.w * w.w);
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
        ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
      }
    }
    const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H, r = rstd[row];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        float4 o = make_float4(r * (g[i].x - m1 - xh[i].x * m2), r * (g[i].y - m1 - xh[i].y * m2),
                               r * (g[i].z - m1 - xh[i].z * m2), r * (g[i].w - m1 - xh[i].w * m2));
        if (du) ((float4*)(du + (long)row * H))[c] = o;
        if (dzb) {
          if (keep) { const float4 k = ((const float4*)(keep + (long)row * H))[c]; o.x *= k.x; o.y *= k.y; o.z *= k.z; o.w *= k.w; }
          uint2 p; p.x = pack2bf(o.x, o.y); p.y = pack2bf(o.z, o.w);
          ((uint2*)(dzb + (long)row * H))[c] = p;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      atomicAdd(&part[c * 4 + 0], ag[i].x); atomicAdd(&part[c * 4 + 1], ag[i].y);
      atomicAdd(&part[c * 4 + 2], ag[i].z); atomicAdd(&part[c * 4 + 3], ag[i].w);
      atomicAdd(&part[H + c * 4 + 0], ab[i].x); atomicAdd(&part[H + c * 4 + 1], ab[i].y);
      atomicAdd(&part[H + c * 4 + 2], ab[i].z); atomicAdd(&part[H + c * 4 + 3], ab[i].w);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H; i += 256) {
    if (dgamma) unsafeAtomicAdd(dgamma + i, part[i]);
    if (dbeta) unsafeAtomicAdd(dbeta + i, part[H + i]);
  }
}

// ---------------------------------------------------------------------------- windowed attention, one workgroup per window
// Q [EB*Lq, H], K / V [EB*Lk, H] bf16, H = heads * hd.  P (softmax probabilities BEFORE dropout) is saved for backward.
constexpr int ATT_MAX_S = 4096;                      // heads * Lq * Lk scores per window held in LDS
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                             const bf16_t* __restrict__ V, const float* __restrict__ keep,
                                                             float* __restrict__ P, bf16_t* __restrict__ O, int heads, int hd,
                                                             int Lq, int Lk, float scale) {
  __shared__ float S[ATT_MAX_S];
  const int e = blockIdx.x, tid = threadIdx.x, H = heads * hd, ns = heads * Lq * Lk;
  const bf16_t* q = Q + (long)e * Lq * H;
  const bf16_t* k = K + (long)e * Lk * H;
  const bf16_t* v = V + (long)e * Lk * H;
  for (int i = tid; i < ns; i += 256) {
    const int h = i / (Lq * Lk), qi = (i / Lk) % Lq, ki = i % Lk;
    const bf16_t* a = q + (long)qi * H + h * hd;
    const bf16_t* b = k + (long)ki * H + h * hd;
    float acc = 0.f;
    for (int d = 0; d < hd; d += 8) {
      const uint4 x = *(const uint4*)(a + d), y = *(const uint4*)(b + d);
      const uint32_t xu[4] = {x.x, x.y, x.z, x.w}, yu[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc += bf2f((bf16_t)(xu[t] & 0xffff)) * bf2f((bf16_t)(yu[t] & 0xffff)) + bf2f((bf16_t)(xu[t] >> 16)) * bf2f((bf16_t)(yu[t] >> 16));
    }
    S[i] = acc * scale;
  }
  __syncthreads();
  for (int r = tid; r < heads * Lq; r += 256) {       // softmax of one (head, query) row per thread
    float* s = S + r * Lk;
    float mx = s[0];
    for (int j = 1; j < Lk; ++j) mx = fmaxf(mx, s[j]);
    float sum = 0.f;
    for (int j = 0; j < Lk; ++j) { s[j] = __expf(s[j] - mx); sum += s[j]; }
    const float inv = 1.0f / sum;
    for (int j = 0; j < Lk; ++j) {
      const float p = s[j] * inv;
      const long gi = (long)e * ns + r * Lk + j;
      P[gi] = p;
      s[j] = keep ? p * keep[gi] : p;
    }
  }
  __syncthreads();
  for (int i = tid; i < Lq * H; i += 256) {
    const int qi = i / H, col = i % H, h = col / hd;
    const float* s = S + (h * Lq + qi) * Lk;
    float acc = 0.f;
    for (int j = 0; j < Lk; ++j) acc += s[j] * bf2f(v[(long)j * H + col]);
    O[((long)e * Lq + qi) * H + col] = f2bf(acc);
  }
}

__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ Q,
                                                             const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                             const float* __restrict__ P, const float* __restrict__ keep,
                                                             bf16_t* __restrict__ dQ, bf16_t* __restrict__ dK,
                                                             bf16_t* __restrict__ dV, int heads, int hd, int Lq, int Lk,
                                                             float scale) {
  __shared__ float Pd[ATT_MAX_S];                     // probabilities after dropout (for dV)
  __shared__ float dS[ATT_MAX_S];
  const int e = blockIdx.x, tid = threadIdx.x, H = heads * hd, ns = heads * Lq * Lk;
  const bf16_t* q = Q + (long)e * Lq * H;
  const bf16_t* k = K + (long)e * Lk * H;
  const bf16_t* v = V + (long)e * Lk * H;
  const bf16_t* go = dO + (long)e * Lq * H;
  for (int i = tid; i < ns; i += 256) {               // dPd[h,q,k] = dO[q,h,:] . V[k,h,:]
    const int h = i / (Lq * Lk), qi = (i / Lk) % Lq, ki = i % Lk;
    const bf16_t* a = go + (long)qi * H + h * hd;
    const bf16_t* b = v + (long)ki * H + h * hd;
    float acc = 0.f;
    for (int d = 0; d < hd; d += 8) {
      const uint4 x = *(const uint4*)(a + d), y = *(const uint4*)(b + d);
      const uint32_t xu[4] = {x.x, x.y, x.z, x.w}, yu[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc += bf2f((bf16_t)(xu[t] & 0xffff)) * bf2f((bf16_t)(yu[t] & 0xffff)) + bf2f((bf16_t)(xu[t] >> 16)) * bf2f((bf16_t)(yu[t] >> 16));
    }
    const long gi = (long)e * ns + i;
    const float kp = keep ? keep[gi] : 1.f, p = P[gi];
    Pd[i] = p * kp;
    dS[i] = acc * kp;                                  // dP (w.r.t. the pre-dropout probabilities), finished below
  }
  __syncthreads();
  for (int r = tid; r < heads * Lq; r += 256) {
    float* d = dS + r * Lk;
    const float* p = P + (long)e * ns + r * Lk;
    float dot = 0.f;
    for (int j = 0; j < Lk; ++j) dot += d[j] * p[j];
    for (int j = 0; j < Lk; ++j) d[j] = p[j] * (d[j] - dot) * scale;      // dS * scale (shared by dQ and dK)
  }
  __syncthreads();
  for (int i = tid; i < Lq * H; i += 256) {            // dQ[q, col] = sum_k dS[h,q,k] K[k, col]
    const int qi = i / H, col = i % H, h = col / hd;
    const float* d = dS + (h * Lq + qi) * Lk;
    float acc = 0.f;
    for (int j = 0; j < Lk; ++j) acc += d[j] * bf2f(k[(long)j * H + col]);
    dQ[((long)e * Lq + qi) * H + col] = f2bf(acc);
  }
  for (int i = tid; i < Lk * H; i += 256) {            // dK[k, col] = sum_q dS[h,q,k] Q[q, col];  dV[k, col] = sum_q Pd[h,q,k] dO[q, col]
    const int ki = i / H, col = i % H, h = col / hd;
    float ak = 0.f, av = 0.f;
    for (int qi = 0; qi < Lq; ++qi) {
      ak += dS[(h * Lq + qi) * Lk + ki] * bf2f(q[(long)qi * H + col]);
      av += Pd[(h * Lq + qi) * Lk + ki] * bf2f(go[(long)qi * H + col]);
    }
    dK[((long)e * Lk + ki) * H + col] = f2bf(ak);
    dV[((long)e * Lk + ki) * H + col] = f2bf(av);
  }
}
}  // namespace

extern "C" int ta_gelu_fwd(const void* h, void* a, long n, hipStream_t st) {
  if (n <= 0) return TA_OK;
  if (n % 8) return TA_ERR_ARG;
  int blocks = (int)((n / 8 + 255) / 256); if (blocks > 4096) blocks = 4096;
  TA_LAUNCH(gelu_fwd_kernel, dim3(blocks), dim3(256), 0, st, (const bf16_t*)h, (bf16_t*)a, n / 8);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_gelu_bwd(const void* da, const void* h, void* dh, long n, hipStream_t st) {
  if (n <= 0) return TA_OK;
  if (n % 8) return TA_ERR_ARG;
  int blocks = (int)((n / 8 + 255) / 256); if (blocks > 4096) blocks = 4096;
  TA_LAUNCH(gelu_bwd_kernel, dim3(blocks), dim3(256), 0, st, (const bf16_t*)da, (const bf16_t*)h, (bf16_t*)dh, n / 8);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_colsum(const void* x, int is_f32, int R, int C, float* out, hipStream_t st) {
  if (R <= 0 || C <= 0) return TA_OK;
  if (hipMemsetAsync(out, 0, (size_t)C * 4, st) != hipSuccess) return TA_ERR_LAUNCH;
  const int chunks = R >= 4096 ? 64 : (R >= 256 ? 16 : 1), per = ta_cdiv(R, chunks);
  if (is_f32) TA_LAUNCH((colsum_kernel<float>), dim3(ta_cdiv(C, 256), chunks), dim3(256), 0, st, (const float*)x, R, C, out, per);
  else TA_LAUNCH((colsum_kernel<bf16_t>), dim3(ta_cdiv(C, 256), chunks), dim3(256), 0, st, (const bf16_t*)x, R, C, out, per);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_layernorm_res_fwd(const float* z, const float* keep, const float* res, long res_rows, const float* gamma,
                                    const float* beta, float eps, float* xhat, float* rstd, float* y_f32, void* y_bf16, int M,
                                    int H, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if (H % 4 || H > 64 * 4 * 8 || !xhat || !rstd) return TA_ERR_ARG;
  if (res && res_rows <= 0) res_rows = M;
  if (H <= 1024) TA_LAUNCH((ln_res_fwd_kernel<4>), dim3(ta_cdiv(M, 4)), dim3(256), 0, st, z, keep, res, res_rows, gamma, beta, eps,
                           xhat, rstd, y_f32, (bf16_t*)y_bf16, M, H);
  else TA_LAUNCH((ln_res_fwd_kernel<8>), dim3(ta_cdiv(M, 4)), dim3(256), 0, st, z, keep, res, res_rows, gamma, beta, eps, xhat,
                 rstd, y_f32, (bf16_t*)y_bf16, M, H);
  TA_CHECK_LAUNCH(); return TA_OK;
}
/* dgamma / dbeta are ACCUMULATED (zero them first); du and dz_bf16 are optional outputs. */
extern "C" int ta_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, const float* keep,
                                float* du, void* dz_bf16, float* dgamma, float* dbeta, int M, int H, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if (H % 4 || H > 64 * 4 * 8) return TA_ERR_ARG;
  const int rpb = 32, blocks = ta_cdiv(M, rpb);
  const size_t smem = (size_t)2 * H * 4;
  if (H <= 1024) TA_LAUNCH((ln_bwd_kernel<4>), dim3(blocks), dim3(256), smem, st, dy, xhat, rstd, gamma, keep, du, (bf16_t*)dz_bf16,
                           dgamma, dbeta, M, H, rpb);
  else TA_LAUNCH((ln_bwd_kernel<8>), dim3(blocks), dim3(256), smem, st, dy, xhat, rstd, gamma, keep, du, (bf16_t*)dz_bf16, dgamma,
                 dbeta, M, H, rpb);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_attn_small_fwd(const void* Q, const void* K, const void* V, int EB, int heads, int hd, int Lq, int Lk,
                                 float scale, const float* keep, float* P, void* O, hipStream_t st) {
  if (EB <= 0) return TA_OK;
  if (hd % 8 || heads * Lq * Lk > ATT_MAX_S || !P) return TA_ERR_ARG;
  TA_LAUNCH(attn_small_fwd_kernel, dim3(EB), dim3(256), 0, st, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, keep, P,
            (bf16_t*)O, heads, hd, Lq, Lk, scale);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_attn_small_bwd(const void* dO, const void* Q, const void* K, const void* V, const float* P, const float* keep,
                                 float scale, void* dQ, void* dK, void* dV, int EB, int heads, int hd, int Lq, int Lk,
                                 hipStream_t st) {
  if (EB <= 0) return TA_OK;
  if (hd % 8 || heads * Lq * Lk > ATT_MAX_S) return TA_ERR_ARG;
  TA_LAUNCH(attn_small_bwd_kernel, dim3(EB), dim3(256), 0, st, (const bf16_t*)dO, (const bf16_t*)Q, (const bf16_t*)K,
            (const bf16_t*)V, P, keep, (bf16_t*)dQ, (bf16_t*)dK, (bf16_t*)dV, heads, hd, Lq, Lk, scale);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_relu_fwd(const void* h, void* a, long n, hipStream_t st) {
  if (n <= 0) return TA_OK;
  int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  TA_LAUNCH(relu_fwd_kernel, dim3(blocks), dim3(256), 0, st, (const bf16_t*)h, (bf16_t*)a, n);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_relu_bwd(const void* da, const void* h, void* dh, long n, hipStream_t st) {
  if (n <= 0) return TA_OK;
  int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  TA_LAUNCH(relu_bwd_kernel, dim3(blocks), dim3(256), 0, st, (const bf16_t*)da, (const bf16_t*)h, (bf16_t*)dh, n);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_mix_fwd(const float* logits, const float* o, float* rw, float* out, int M, int D, int E, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if (E > 16 || E <= 0 || D % 4) return TA_ERR_ARG;
  TA_LAUNCH(mix_fwd_kernel, dim3(ta_cdiv(M, 4)), dim3(256), 0, st, logits, o, rw, out, M, D, E);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_mix_bwd(const float* dout, const float* o, const float* rw, void* do_bf16, float* dlogits, int M, int D, int E,
                          hipStream_t st) {
  if (M <= 0) return TA_OK;
  if (E > 16 || E <= 0 || D % 4) return TA_ERR_ARG;
  TA_LAUNCH(mix_bwd_kernel, dim3(ta_cdiv(M, 4)), dim3(256), 0, st, dout, o, rw, (bf16_t*)do_bf16, dlogits, M, D, E);
  TA_CHECK_LAUNCH(); return TA_OK;
}
