// ta355 shared + sparse MoE projector kernels (tiny_audio/projectors.py:185-351, MoEAudioProjector).
//
// The reference loops over experts in Python with `mask.any()` host syncs, `torch.where`, gathers and an atomic
// `index_add_`.  Here routing never leaves the device: the router kernel writes top-2 choices, a single-block
// planning kernel turns them into a STABLE sort of (token, choice) slots by expert (segments padded to the 64-wide
// GEMM K tile, so per-expert weight gradients contract over an aligned slot range), expert GEMMs read their rows
// through the gather list with device-side segment bounds (ta_gemm_bf16_nt_ex), and the combine is a deterministic
// per-token gather (no atomics, bit-reproducible).
#include <cstdlib>
#include "common.h"

#define MOE_MAX_E 8

// ---------------------------------------------------------------------------- frame-stack + RMSNorm (bf16 in -> bf16 out)
// row t = (b, n) of the stacked input starts at b*bs + n*ld in x (ld = k*E: rows overlap nothing, tail frames dropped)
__global__ __launch_bounds__(256) void moe_norm_kernel(const bf16_t* __restrict__ x, long bs, int rpb, long ld,
                                                       const float* __restrict__ w, bf16_t* __restrict__ xn,
                                                       float* __restrict__ rstd, int T, int In, float eps) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int lane = threadIdx.x & 63;
  const bf16_t* xr = x + (long)(t / rpb) * bs + (long)(t % rpb) * ld;
  float q = 0.f;
  for (int c = lane * 8; c < In; c += 512) {
    const uint4 v = *(const uint4*)(xr + c);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float a = bf2f(u[j] & 0xffff), b = bf2f(u[j] >> 16); q += a * a + b * b; }
  }
  const float r = rsqrtf(wave_sum(q) / (float)In + eps);
  if (lane == 0) rstd[t] = r;
  for (int c = lane * 8; c < In; c += 512) {
    const uint4 v = *(const uint4*)(xr + c);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
    const float4 w0 = *(const float4*)(w + c), w1 = *(const float4*)(w + c + 4);
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack2bf(bf2f(u[j] & 0xffff) * r * ww[2 * j], bf2f(u[j] >> 16) * r * ww[2 * j + 1]);
    *(uint4*)(xn + (long)t * In + c) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ---------------------------------------------------------------------------- router (projectors.py:291-325)
// one wave per ROUTER_TPW tokens (the router rows are read once per wave, not once per token: at one token per wave the kernel
// streamed E * In floats of Wr per token through L2 -- 320 MB for 4000 tokens, 232 us): logits = xn . Wr[e] (fp32), optional
// multiplicative jitter, fp32 softmax, top-2, renormalise by (sum + 1e-6).  Accumulates sum_t probs[e] and sum_t logsumexp^2
// for the aux loss.
#define ROUTER_TPW 4
__global__ __launch_bounds__(256) void moe_router_kernel(const bf16_t* __restrict__ xn, const float* __restrict__ wr,
                                                         const float* __restrict__ noise, float* __restrict__ logits_out,
                                                         float* __restrict__ probs_out, int* __restrict__ topi,
                                                         float* __restrict__ topw, float* __restrict__ topraw,
                                                         float* __restrict__ lse_out, float* __restrict__ psum,
                                                         float* __restrict__ zsum, int T, int In, int E) {
  const int t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROUTER_TPW;
  if (t0 >= T) return;
  const int lane = threadIdx.x & 63;
  float acc[ROUTER_TPW][MOE_MAX_E];
#pragma unroll
  for (int q = 0; q < ROUTER_TPW; ++q)
#pragma unroll
    for (int e = 0; e < MOE_MAX_E; ++e) acc[q][e] = 0.f;
  for (int c = lane * 8; c < In; c += 512) {
    float xv[ROUTER_TPW][8];
#pragma unroll
    for (int q = 0; q < ROUTER_TPW; ++q) {
      const int t = min(t0 + q, T - 1);
      const uint4 v = *(const uint4*)(xn + (long)t * In + c);
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { xv[q][2 * j] = bf2f(u[j] & 0xffff); xv[q][2 * j + 1] = bf2f(u[j] >> 16); }
    }
#pragma unroll
    for (int e = 0; e < MOE_MAX_E; ++e) {
      if (e < E) {
        const float4 a = *(const float4*)(wr + (long)e * In + c), b = *(const float4*)(wr + (long)e * In + c + 4);
        // explicit fused multiply-adds in a fixed order: left to the compiler, the contraction of this sum came out differently
        // for different q, and a token's logits depended (in the last bit) on its place in the group -- enough to flip a top-2 tie
        const float wv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < ROUTER_TPW; ++q) {
          float r = acc[q][e];
#pragma unroll
          for (int j = 0; j < 8; ++j) r = __builtin_fmaf(xv[q][j], wv[j], r);
          acc[q][e] = r;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < ROUTER_TPW; ++q)
#pragma unroll
    for (int e = 0; e < MOE_MAX_E; ++e) acc[q][e] = __shfl(wave_sum(acc[q][e]), 0, 64);   // lane 0's sum for every token: the butterfly rounds
                                                                                          // differently per lane, and a token's logits must not depend on its place in the group
  if (lane >= ROUTER_TPW || t0 + lane >= T) return;
  const int t = t0 + lane;                            // lane q finishes token t0 + q
  float lg[MOE_MAX_E], m = -INFINITY;
#pragma unroll
  for (int e = 0; e < MOE_MAX_E; ++e) {
    float a = acc[0][e];
#pragma unroll
    for (int q = 1; q < ROUTER_TPW; ++q) a = lane == q ? acc[q][e] : a;
    lg[e] = a;
  }
  for (int e = 0; e < E; ++e) {
    lg[e] = lg[e] * (noise ? noise[(long)t * E + e] : 1.0f);
    m = fmaxf(m, lg[e]);
  }
  float s = 0.f, p[MOE_MAX_E];
  for (int e = 0; e < E; ++e) { p[e] = __expf(lg[e] - m); s += p[e]; }
  const float lse = m + __logf(s);
  int i0 = 0, i1 = -1;
  for (int e = 0; e < E; ++e) { p[e] /= s; logits_out[(long)t * E + e] = lg[e]; probs_out[(long)t * E + e] = p[e]; }
  for (int e = 1; e < E; ++e) if (p[e] > p[i0]) i0 = e;                 // first maximum wins ties (torch.topk order)
  for (int e = 0; e < E; ++e) if (e != i0 && (i1 < 0 || p[e] > p[i1])) i1 = e;
  const float den = p[i0] + p[i1] + 1e-6f;
  topi[2 * t] = i0; topi[2 * t + 1] = i1;
  topraw[2 * t] = p[i0]; topraw[2 * t + 1] = p[i1];
  topw[2 * t] = p[i0] / den; topw[2 * t + 1] = p[i1] / den;
  lse_out[t] = lse;
  if (psum) { for (int e = 0; e < E; ++e) atomicAdd(psum + e, p[e]); atomicAdd(zsum, lse * lse); }
}

// aux = coef * E * mean_e((pbar_e - 1/E)^2) + zcoef * mean_t(lse^2)      (projectors.py:312-325)
__global__ void moe_aux_kernel(const float* __restrict__ psum, const float* __restrict__ zsum, float* __restrict__ aux, int T,
                               int E, float coef, float zcoef) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float b = 0.f;
    for (int e = 0; e < E; ++e) { const float d = psum[e] / (float)T - 1.0f / (float)E; b += d * d; }
    aux[0] = coef * (b / (float)E) * (float)E + zcoef * zsum[0] / (float)T;
  }
}

// ---------------------------------------------------------------------------- plan: stable sort of 2T slots by expert
// out: seg[e] = {offset (multiple of 64), count}, kr[e] = {offset/64, (offset + pad64(count))/64},
//      perm[pos] = token (or -1 in padding), slot_of[2t+k] = pos.   Single block of 1024 threads.
__global__ __launch_bounds__(1024) void moe_plan_kernel(const int* __restrict__ topi, int T, int E, int Smax,
                                                        int* __restrict__ seg, int* __restrict__ kr, int* __restrict__ perm,
                                                        int* __restrict__ slot_of) {
  __shared__ int scan[1024];
  __shared__ int cnt[MOE_MAX_E], off[MOE_MAX_E], run[MOE_MAX_E];
  const int tid = threadIdx.x, n = 2 * T;
  if (tid < MOE_MAX_E) { cnt[tid] = 0; run[tid] = 0; }
  for (int i = tid; i < Smax; i += 1024) perm[i] = -1;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) atomicAdd(&cnt[topi[i]], 1);      // integer counts: order-independent
  __syncthreads();
  if (tid == 0) {
    int o = 0;
    for (int e = 0; e < E; ++e) {
      off[e] = o; seg[2 * e] = o; seg[2 * e + 1] = cnt[e];
      const int padded = (cnt[e] + 63) / 64 * 64;
      kr[2 * e] = o / 64; kr[2 * e + 1] = (o + padded) / 64;
      o += padded;
    }
  }
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int e_me = i < n ? topi[i] : -1;
    for (int e = 0; e < E; ++e) {
      const int flag = e_me == e ? 1 : 0;
      scan[tid] = flag;
      __syncthreads();
      for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? scan[tid - o] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
      }
      if (flag) {
        const int pos = off[e] + run[e] + scan[tid] - 1;
        perm[pos] = i >> 1;
        slot_of[i] = pos;
      }
      __syncthreads();
      if (tid == 1023) run[e] += scan[1023];
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------- element-wise pieces
// act = gelu(h) (bf16 -> bf16), exact erf: the projector is the trainable part
__global__ void gelu_bf16_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ a, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const uint2 v = ((const uint2*)h)[i];
    uint2 o;
    o.x = pack2bf(gelu_erf(bf2f(v.x & 0xffff)), gelu_erf(bf2f(v.x >> 16)));
    o.y = pack2bf(gelu_erf(bf2f(v.y & 0xffff)), gelu_erf(bf2f(v.y >> 16)));
    ((uint2*)a)[i] = o;
  }
}
// dh = dact * gelu'(h)
__global__ void gelu_bwd_bf16_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ h, bf16_t* __restrict__ dh, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const uint2 d = ((const uint2*)dact)[i], v = ((const uint2*)h)[i];
    uint2 o;
    o.x = pack2bf(bf2f(d.x & 0xffff) * gelu_erf_grad(bf2f(v.x & 0xffff)), bf2f(d.x >> 16) * gelu_erf_grad(bf2f(v.x >> 16)));
    o.y = pack2bf(bf2f(d.y & 0xffff) * gelu_erf_grad(bf2f(v.y & 0xffff)), bf2f(d.y >> 16) * gelu_erf_grad(bf2f(v.y >> 16)));
    ((uint2*)dh)[i] = o;
  }
}
// out[t,:] = shared[t,:] + sum_k topw[t,k] * y[slot_of[t,k],:]      (in place on `out` = shared)
__global__ __launch_bounds__(256) void moe_combine_kernel(float* __restrict__ out, const float* __restrict__ y,
                                                          const int* __restrict__ slot_of, const float* __restrict__ topw,
                                                          int T, int D) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int lane = threadIdx.x & 63;
  const long s0 = slot_of[2 * t], s1 = slot_of[2 * t + 1];
  const float w0 = topw[2 * t], w1 = topw[2 * t + 1];
  for (int c = lane; c < D / 4; c += 64) {
    float4 o = ((float4*)(out + (long)t * D))[c];
    const float4 a = ((const float4*)(y + s0 * D))[c], b = ((const float4*)(y + s1 * D))[c];
    o.x += w0 * a.x + w1 * b.x; o.y += w0 * a.y + w1 * b.y; o.z += w0 * a.z + w1 * b.z; o.w += w0 * a.w + w1 * b.w;
    ((float4*)(out + (long)t * D))[c] = o;
  }
}
// backward of the combine: dy_slot[pos,:] = w * dout[t,:] (bf16), dtopw[t,k] = <dout[t,:], y[pos,:]>; also dout -> bf16
__global__ __launch_bounds__(256) void moe_combine_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                              const int* __restrict__ slot_of, const float* __restrict__ topw,
                                                              bf16_t* __restrict__ dy_slot, float* __restrict__ dtopw,
                                                              bf16_t* __restrict__ dout_b, int T, int D) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const int lane = threadIdx.x & 63;
  const long s0 = slot_of[2 * t], s1 = slot_of[2 * t + 1];
  const float w0 = topw[2 * t], w1 = topw[2 * t + 1];
  float d0 = 0.f, d1 = 0.f;
  for (int c = lane; c < D / 4; c += 64) {
    const float4 g = ((const float4*)(dout + (long)t * D))[c];
    const float4 a = ((const float4*)(y + s0 * D))[c], b = ((const float4*)(y + s1 * D))[c];
    d0 += g.x * a.x + g.y * a.y + g.z * a.z + g.w * a.w;
    d1 += g.x * b.x + g.y * b.y + g.z * b.z + g.w * b.w;
    uint2 o; o.x = pack2bf(g.x * w0, g.y * w0); o.y = pack2bf(g.z * w0, g.w * w0);
    ((uint2*)(dy_slot + s0 * D))[c] = o;
    o.x = pack2bf(g.x * w1, g.y * w1); o.y = pack2bf(g.z * w1, g.w * w1);
    ((uint2*)(dy_slot + s1 * D))[c] = o;
    o.x = pack2bf(g.x, g.y); o.y = pack2bf(g.z, g.w);
    ((uint2*)(dout_b + (long)t * D))[c] = o;
  }
  d0 = wave_sum(d0); d1 = wave_sum(d1);
  if (lane == 0) { dtopw[2 * t] = d0; dtopw[2 * t + 1] = d1; }
}

// column sums of a bf16 matrix over a device-side row segment (bias gradients): out[c] = sum_r X[base + r, c]
__global__ __launch_bounds__(256) void colsum_seg_kernel(const bf16_t* __restrict__ X, int C, const int* __restrict__ seg,
                                                         int rows_if_no_seg, float* __restrict__ out, int row_chunks) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int base = seg ? seg[0] : 0, cnt = seg ? seg[1] : rows_if_no_seg;
  const int per = (cnt + row_chunks - 1) / row_chunks;
  const int r0 = blockIdx.y * per, r1 = min(cnt, r0 + per);
  if (c >= C) return;
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += bf2f(X[(long)(base + r) * C + c]);
  if (r1 > r0) atomicAdd(out + c, s);
}
// slot-ordered transposes for the per-expert weight gradients: out [C, Smax] bf16, column pos = row perm[pos] of X
// (gathered) or row pos of X (slot-ordered input); padding columns (perm = -1) are zero.
__global__ __launch_bounds__(256) void slot_transpose_kernel(const bf16_t* __restrict__ X, int C, const int* __restrict__ perm,
                                                             int gather, bf16_t* __restrict__ out, int Smax) {
  __shared__ bf16_t tile[64][66];
  const int p0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int pos = p0 + i, c = c0 + tx;
    bf16_t v = 0;
    if (pos < Smax && c < C) {
      const int t = perm[pos];
      if (t >= 0) v = X[(long)(gather ? t : pos) * C + c];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, pos = p0 + tx;
    if (c < C && pos < Smax) out[(long)c * Smax + pos] = tile[tx][i];
  }
}

// ---------------------------------------------------------------------------- router backward
// dlogits[t,e] from d(top-2 weights), the renormalisation, the softmax, jitter and the aux losses.
__global__ void moe_router_bwd_kernel(const float* __restrict__ dtopw, const float* __restrict__ probs,
                                      const int* __restrict__ topi, const float* __restrict__ topraw,
                                      const float* __restrict__ lse, const float* __restrict__ noise,
                                      const float* __restrict__ psum, float* __restrict__ dlogits, int T, int E, float d_aux,
                                      const float* __restrict__ d_aux_dev, float coef, float zcoef, int training) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  if (d_aux_dev) d_aux = *d_aux_dev;               // upstream gradient of the auxiliary loss read on the device (no host sync)
  float p[MOE_MAX_E], dp[MOE_MAX_E];
  for (int e = 0; e < E; ++e) { p[e] = probs[(long)t * E + e]; dp[e] = 0.f; }
  const int i0 = topi[2 * t], i1 = topi[2 * t + 1];
  const float r0 = topraw[2 * t], r1 = topraw[2 * t + 1], den = r0 + r1 + 1e-6f;
  const float g0 = dtopw ? dtopw[2 * t] : 0.f, g1 = dtopw ? dtopw[2 * t + 1] : 0.f;   // (null: the auxiliary-loss share alone)
  const float common = (g0 * r0 + g1 * r1) / (den * den);
  dp[i0] = g0 / den - common;
  dp[i1] = g1 / den - common;
  float dl[MOE_MAX_E];
  for (int e = 0; e < E; ++e) dl[e] = 0.f;
  if (training) {
    for (int e = 0; e < E; ++e) {
      dp[e] += d_aux * coef * 2.0f * (psum[e] / (float)T - 1.0f / (float)E) / (float)T;
      dl[e] += d_aux * zcoef * 2.0f * lse[t] * p[e] / (float)T;
    }
  }
  float dot = 0.f;
  for (int e = 0; e < E; ++e) dot += dp[e] * p[e];
  for (int e = 0; e < E; ++e) {
    float v = dl[e] + p[e] * (dp[e] - dot);
    if (noise) v *= noise[(long)t * E + e];
    dlogits[(long)t * E + e] = v;
  }
}
// dWr[e, c] += sum_t dlogits[t,e] * xn[t,c]
__global__ __launch_bounds__(256) void moe_router_dw_kernel(const float* __restrict__ dlogits, const bf16_t* __restrict__ xn,
                                                            float* __restrict__ dwr, int T, int In, int E, int chunks) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int per = (T + chunks - 1) / chunks;
  const int t0 = blockIdx.y * per, t1 = min(T, t0 + per);
  if (c >= In) return;
  float acc[MOE_MAX_E];
#pragma unroll
  for (int e = 0; e < MOE_MAX_E; ++e) acc[e] = 0.f;
  for (int t = t0; t < t1; ++t) {
    const float x = bf2f(xn[(long)t * In + c]);
#pragma unroll
    for (int e = 0; e < MOE_MAX_E; ++e) if (e < E) acc[e] += dlogits[(long)t * E + e] * x;
  }
  for (int e = 0; e < E; ++e) atomicAdd(dwr + (long)e * In + c, acc[e]);
}
// d(norm.weight)[c] += sum_t dxn[t,c] * xs[t,c] * rstd[t], with
// dxn[t,c] = dxn_shared[t,c] + dxn_slot[slot0,c] + dxn_slot[slot1,c] + sum_e dlogits[t,e] * Wr[e,c]
// (the encoder is frozen, so dxn itself is consumed here and never written out)
__global__ __launch_bounds__(256) void moe_norm_bwd_kernel(const float* __restrict__ dxn_sh, const float* __restrict__ dxn_slot,
                                                           const int* __restrict__ slot_of, const float* __restrict__ dlogits,
                                                           const float* __restrict__ wr, const bf16_t* __restrict__ x, long bs,
                                                           int rpb, long ld, const float* __restrict__ rstd,
                                                           float* __restrict__ dg, int T, int In, int E, int chunks) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int per = (T + chunks - 1) / chunks;
  const int t0 = blockIdx.y * per, t1 = min(T, t0 + per);
  if (c >= In) return;
  float wre[MOE_MAX_E];
#pragma unroll
  for (int e = 0; e < MOE_MAX_E; ++e) wre[e] = e < E ? wr[(long)e * In + c] : 0.f;
  float acc = 0.f;
  for (int t = t0; t < t1; ++t) {
    float v = dxn_sh ? dxn_sh[(long)t * In + c] + dxn_slot[(long)slot_of[2 * t] * In + c] + dxn_slot[(long)slot_of[2 * t + 1] * In + c] : 0.f;
#pragma unroll
    for (int e = 0; e < MOE_MAX_E; ++e) if (e < E) v += dlogits[(long)t * E + e] * wre[e];
    const float xs = bf2f(x[(long)(t / rpb) * bs + (long)(t % rpb) * ld + c]);
    acc += v * xs * rstd[t];
  }
  atomicAdd(dg + c, acc);
}

// ============================================================================ composites (C ABI, include/ta355.h)
#include "../../include/ta355.h"

namespace {
struct MoeCarver {
  char* base; size_t off;
  explicit MoeCarver(void* b) : base((char*)b), off(0) {}
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
  size_t total() const { return (off + 255) & ~(size_t)255; }
};
inline int p64(int x) { return (x + 63) / 64 * 64; }
struct MoeDims { int T, N, In, H, D, E, Tp, Smax; };
MoeDims moe_dims(const ta_moe_weights* w, int B, int S) {
  MoeDims d;
  d.N = (S - w->k) / w->k + 1; d.T = B * d.N; d.In = w->k * w->enc_dim; d.H = w->hidden; d.D = w->llm_dim;
  d.E = w->num_experts; d.Tp = p64(d.T); d.Smax = p64(2 * d.T + 64 * d.E);
  return d;
}
struct MoeTape {
  bf16_t *xn, *h_s, *act_s, *h_e, *act_e;
  float *rstd, *logits, *probs, *topw, *topraw, *lse, *psum, *zsum, *y_e;
  int *topi, *seg, *kr, *perm, *slot_of;
  size_t bytes;
};
MoeTape moe_tape(const ta_moe_weights* w, int B, int S, void* base) {
  const MoeDims d = moe_dims(w, B, S);
  MoeCarver c(base); MoeTape t;
  t.xn = c.take<bf16_t>((size_t)d.T * d.In); t.rstd = c.take<float>(d.T);
  t.logits = c.take<float>((size_t)d.T * d.E); t.probs = c.take<float>((size_t)d.T * d.E);
  t.topi = c.take<int>(2 * d.T); t.topw = c.take<float>(2 * d.T); t.topraw = c.take<float>(2 * d.T);
  t.lse = c.take<float>(d.T); t.psum = c.take<float>(MOE_MAX_E); t.zsum = c.take<float>(4);
  t.seg = c.take<int>(2 * MOE_MAX_E); t.kr = c.take<int>(2 * MOE_MAX_E);
  t.perm = c.take<int>(d.Smax); t.slot_of = c.take<int>(2 * d.T);
  t.h_s = c.take<bf16_t>((size_t)d.T * d.H); t.act_s = c.take<bf16_t>((size_t)d.T * d.H);
  t.h_e = c.take<bf16_t>((size_t)d.Smax * d.H); t.act_e = c.take<bf16_t>((size_t)d.Smax * d.H);
  t.y_e = c.take<float>((size_t)d.Smax * d.D);
  t.bytes = c.total();
  return t;
}
struct MoeWs {
  bf16_t *dout_b, *dy_slot, *doutT, *actsT, *dact_s, *dh_s, *dhsT, *xnT, *dyT, *acteT, *dact_e, *dh_e, *dheT, *xngT;
  float *dtopw, *dlogits, *dxn_sh, *dxn_slot, *skws;
  size_t bytes;
};
inline int moe_splits(int M, int N, int K) {
  const long tiles = (long)ta_cdiv(M, 128) * ta_cdiv(N, 128);
  int s = 1;
  while (tiles * s < 512 && K / 64 / (s * 2) >= 8 && s < 64) s *= 2;
  return s;
}
MoeWs moe_ws(const ta_moe_weights* w, int B, int S, void* base) {
  const MoeDims d = moe_dims(w, B, S);
  MoeCarver c(base); MoeWs s;
  s.dout_b = c.take<bf16_t>((size_t)d.T * d.D); s.dy_slot = c.take<bf16_t>((size_t)d.Smax * d.D);
  s.dtopw = c.take<float>(2 * d.T); s.dlogits = c.take<float>((size_t)d.T * d.E);
  s.doutT = c.take<bf16_t>((size_t)d.D * d.Tp); s.actsT = c.take<bf16_t>((size_t)d.H * d.Tp);
  s.dact_s = c.take<bf16_t>((size_t)d.T * d.H); s.dh_s = c.take<bf16_t>((size_t)d.T * d.H);
  s.dhsT = c.take<bf16_t>((size_t)d.H * d.Tp); s.xnT = c.take<bf16_t>((size_t)d.In * d.Tp);
  s.dxn_sh = c.take<float>((size_t)d.T * d.In);
  s.dyT = c.take<bf16_t>((size_t)d.D * d.Smax); s.acteT = c.take<bf16_t>((size_t)d.H * d.Smax);
  s.dact_e = c.take<bf16_t>((size_t)d.Smax * d.H); s.dh_e = c.take<bf16_t>((size_t)d.Smax * d.H);
  s.dheT = c.take<bf16_t>((size_t)d.H * d.Smax); s.xngT = c.take<bf16_t>((size_t)d.In * d.Smax);
  s.dxn_slot = c.take<float>((size_t)d.Smax * d.In);
  const size_t sk1 = (size_t)ta_gemm_splitk_ws_bytes(d.H, d.In, moe_splits(d.H, d.In, d.Tp));
  const size_t sk2 = (size_t)ta_gemm_splitk_ws_bytes(d.D, d.H, moe_splits(d.D, d.H, d.Tp));
  s.skws = c.take<float>((sk1 > sk2 ? sk1 : sk2) / 4 + 4);
  s.bytes = c.total();
  return s;
}
inline int ew(long n) { long b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }
#define MRC(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)
inline int gemm_plain(const void* A, const void* W, void* C, int M, int N, int K, const float* bias, int out_bf16, int splits,
                      float* skws, hipStream_t st) {
  return ta_gemm_bf16_nt(A, W, C, M, N, K, K, 0, 0, N, 0, 0, 0, bias, nullptr, 0, out_bf16, splits, skws, st);
}
inline int gemm_seg(const void* A, const void* W, void* C, int Mmax, int N, int K, const float* bias, int out_bf16,
                    const int* a_idx, const int* seg, const int* krange, hipStream_t st) {
  return ta_gemm_bf16_nt_ex(A, W, C, Mmax, N, K, K, 0, 0, N, 0, 0, 0, bias, nullptr, 0, out_bf16, 1, nullptr, a_idx, seg, krange, st);
}
// Element stride between consecutive routed experts' buffers when they are laid out at a constant distance (the Python
// module packs them into one tensor); 0 = not uniformly strided (then the per-expert launches are used).
template <typename T> long expert_stride(const void* const* ptrs, int E) {
  if (E < 2) return 0;
  const long s = (const T*)ptrs[1] - (const T*)ptrs[0];
  if (s <= 0) return 0;
  for (int e = 1; e + 1 < E; ++e)
    if ((const T*)ptrs[e + 1] - (const T*)ptrs[e] != s) return 0;
  return s;
}
// one grouped launch per matrix whenever the experts' buffers sit at a constant stride (one launch per expert otherwise)
inline bool grouped_enabled() { return true; }
}  // namespace

// ---- bf16 images of all E + 1 adapters in TWO launches (round 4; they were 20 cast / transpose launches + 10 bias copies per step):
// W1 [H, In] and W2 [D, H] f32 masters of every adapter -> the stacked row-major images W1a [n, H, In], W2a [n, D, H] and their
// transposes W1ta [n, In, H], W2ta [n, H, D]; biases -> B1a [n, H], B2a [n, D].  64 x 64 tiles through LDS (both images of a tile are
// written with 16-byte / 8-byte rows).  The master pointers travel by value in the kernel arguments (n <= MOE_MAX_E + 1).
struct MoePackArgs {
  const float* w1[MOE_MAX_E + 1]; const float* w2[MOE_MAX_E + 1]; const float* b1[MOE_MAX_E + 1]; const float* b2[MOE_MAX_E + 1];
  bf16_t *W1a, *W1ta, *W2a, *W2ta; float *B1a, *B2a; int n, H, In, D;
};
__global__ __launch_bounds__(256) void moe_pack_w_kernel(MoePackArgs a) {
  __shared__ float tile[64][65];
  const int which = blockIdx.y, ad = blockIdx.z;
  const int R = which == 0 ? a.H : a.D, Cn = which == 0 ? a.In : a.H;      // master [R, Cn]
  const int tcols = (Cn + 63) / 64, trows = (R + 63) / 64;
  if ((int)blockIdx.x >= tcols * trows) return;
  const int r0 = (blockIdx.x / tcols) * 64, c0 = (blockIdx.x % tcols) * 64;
  const float* src = which == 0 ? a.w1[ad] : a.w2[ad];
  bf16_t* rm = (which == 0 ? a.W1a : a.W2a) + (long)ad * R * Cn;
  bf16_t* tr = (which == 0 ? a.W1ta : a.W2ta) + (long)ad * R * Cn;
  const int tid = threadIdx.x;
  for (int e = tid; e < 64 * 16; e += 256) {                       // 64 rows x 16 float4
    const int r = e >> 4, c4 = (e & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < R && c0 + c4 + 3 < Cn) v = *(const float4*)(src + (long)(r0 + r) * Cn + c0 + c4);
    else if (r0 + r < R) { float t[4] = {0.f, 0.f, 0.f, 0.f}; for (int j = 0; j < 4; ++j) if (c0 + c4 + j < Cn) t[j] = src[(long)(r0 + r) * Cn + c0 + c4 + j]; v = make_float4(t[0], t[1], t[2], t[3]); }
    tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
    if (r0 + r < R) {
      if (c0 + c4 + 3 < Cn) { uint2 o; o.x = pack2bf(v.x, v.y); o.y = pack2bf(v.z, v.w); *(uint2*)(rm + (long)(r0 + r) * Cn + c0 + c4) = o; }
      else { const float t[4] = {v.x, v.y, v.z, v.w}; for (int j = 0; j < 4; ++j) if (c0 + c4 + j < Cn) rm[(long)(r0 + r) * Cn + c0 + c4 + j] = f2bf(t[j]); }
    }
  }
  __syncthreads();
  for (int e = tid; e < 64 * 16; e += 256) {                       // transposed: 64 rows (source columns) x 16 groups of 4 source rows
    const int c = e >> 4, r4 = (e & 15) * 4;
    if (c0 + c >= Cn) continue;
    if (r0 + r4 + 3 < R) { uint2 o; o.x = pack2bf(tile[r4][c], tile[r4 + 1][c]); o.y = pack2bf(tile[r4 + 2][c], tile[r4 + 3][c]); *(uint2*)(tr + (long)(c0 + c) * R + r0 + r4) = o; }
    else for (int j = 0; j < 4; ++j) if (r0 + r4 + j < R) tr[(long)(c0 + c) * R + r0 + r4 + j] = f2bf(tile[r4 + j][c]);
  }
}
__global__ __launch_bounds__(256) void moe_pack_b_kernel(MoePackArgs a) {
  const int which = blockIdx.y, ad = blockIdx.z, n = which == 0 ? a.H : a.D;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  (which == 0 ? a.B1a : a.B2a)[(long)ad * n + i] = (which == 0 ? a.b1[ad] : a.b2[ad])[i];
}
extern "C" int ta_moe_pack_images(const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2, int n,
                                  int H, int In, int D, void* W1a, void* W1ta, void* W2a, void* W2ta, float* B1a, float* B2a,
                                  hipStream_t st) {
  if (n <= 0) return TA_OK;
  if (n > MOE_MAX_E + 1 || !w1 || !b1 || !w2 || !b2 || !W1a || !W1ta || !W2a || !W2ta || !B1a || !B2a || H <= 0 || In <= 0 || D <= 0 ||
      (H % 4) || (In % 4) || (D % 4))
    return TA_ERR_ARG;
  MoePackArgs a;
  for (int i = 0; i < n; ++i) { a.w1[i] = w1[i]; a.w2[i] = w2[i]; a.b1[i] = b1[i]; a.b2[i] = b2[i]; if (!w1[i] || !w2[i] || !b1[i] || !b2[i]) return TA_ERR_ARG; }
  a.W1a = (bf16_t*)W1a; a.W1ta = (bf16_t*)W1ta; a.W2a = (bf16_t*)W2a; a.W2ta = (bf16_t*)W2ta; a.B1a = B1a; a.B2a = B2a;
  a.n = n; a.H = H; a.In = In; a.D = D;
  const int t1 = ta_cdiv(H, 64) * ta_cdiv(In, 64), t2 = ta_cdiv(D, 64) * ta_cdiv(H, 64);
  TA_LAUNCH(moe_pack_w_kernel, dim3(t1 > t2 ? t1 : t2, 2, n), dim3(256), 0, st, a);
  TA_LAUNCH(moe_pack_b_kernel, dim3(ta_cdiv(H > D ? H : D, 256), 2, n), dim3(256), 0, st, a);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

namespace {
}  // namespace (re-opened so that the closing brace below keeps its meaning)

extern "C" long ta_moe_tape_bytes(const ta_moe_weights* w, int B, int S) { return (long)moe_tape(w, B, S, nullptr).bytes; }
extern "C" long ta_moe_bwd_workspace_bytes(const ta_moe_weights* w, int B, int S) { return (long)moe_ws(w, B, S, nullptr).bytes; }

extern "C" int ta_moe_projector_forward(const ta_moe_weights* w, const void* x, int B, int S, const float* noise, int training,
                                        float* y, float* aux, void* tape, hipStream_t st) {
  const MoeDims d = moe_dims(w, B, S);
  if (B <= 0 || d.N <= 0) return TA_OK;
  if (d.E < 2 || d.E > MOE_MAX_E || d.In % 512 || d.H % 64 || d.D % 4) return TA_ERR_ARG;
  MoeTape t = moe_tape(w, B, S, tape);
  const int E = d.E;
  if (hipMemsetAsync(t.psum, 0, (MOE_MAX_E + 4) * 4 + 256, st) != hipSuccess) return TA_ERR_LAUNCH;   // psum and zsum are adjacent carves
  TA_LAUNCH(moe_norm_kernel, dim3(ta_cdiv(d.T, 4)), dim3(256), 0, st, (const bf16_t*)x, (long)S * w->enc_dim, d.N, (long)d.In,
            w->norm_w, t.xn, t.rstd, d.T, d.In, w->eps);
  TA_LAUNCH(moe_router_kernel, dim3(ta_cdiv(d.T, 4 * ROUTER_TPW)), dim3(256), 0, st, t.xn, w->router_w, training ? noise : nullptr, t.logits,
            t.probs, t.topi, t.topw, t.topraw, t.lse, training ? t.psum : nullptr, t.zsum, d.T, d.In, E);
  TA_LAUNCH(moe_plan_kernel, dim3(1), dim3(1024), 0, st, t.topi, d.T, E, d.Smax, t.seg, t.kr, t.perm, t.slot_of);
  TA_CHECK_LAUNCH();
  // shared expert on every token
  MRC(gemm_plain(t.xn, w->w1[E], t.h_s, d.T, d.H, d.In, w->b1[E], 1, 1, nullptr, st));
  TA_LAUNCH(gelu_bf16_kernel, dim3(ew((long)d.T * d.H / 4)), dim3(256), 0, st, t.h_s, t.act_s, (long)d.T * d.H / 4);
  MRC(gemm_plain(t.act_s, w->w2[E], y, d.T, d.D, d.H, w->b2[E], 0, 1, nullptr, st));
  // routed experts over their slot segments (row counts stay on the device): ONE grouped launch per matrix -- the
  // tile -> (expert, row tile) map is resolved in the kernel from the plan's segment table -- when the experts' weights
  // and biases sit at a constant stride (TA355_MOE_GROUPED=0: one launch per expert, 4x fewer workgroups each)
  const long s_w1 = expert_stride<bf16_t>(w->w1, E), s_w2 = expert_stride<bf16_t>(w->w2, E);
  const bool g1 = grouped_enabled() && s_w1 > 0 && expert_stride<float>((const void* const*)w->b1, E) == d.H;
  const bool g2 = grouped_enabled() && s_w2 > 0 && expert_stride<float>((const void* const*)w->b2, E) == d.D;
  if (g1) MRC(ta_gemm_bf16_nt_grouped(t.xn, w->w1[0], t.h_e, 2 * d.T, d.H, d.In, w->b1[0], 0, 1, t.perm, t.seg, nullptr, E, s_w1, 0, st));
  else for (int e = 0; e < E; ++e)
    MRC(gemm_seg(t.xn, w->w1[e], t.h_e, d.T, d.H, d.In, w->b1[e], 1, t.perm, t.seg + 2 * e, nullptr, st));
  TA_LAUNCH(gelu_bf16_kernel, dim3(ew((long)d.Smax * d.H / 4)), dim3(256), 0, st, t.h_e, t.act_e, (long)d.Smax * d.H / 4);
  if (g2) MRC(ta_gemm_bf16_nt_grouped(t.act_e, w->w2[0], t.y_e, 2 * d.T, d.D, d.H, w->b2[0], 0, 0, nullptr, t.seg, nullptr, E, s_w2, 0, st));
  else for (int e = 0; e < E; ++e)
    MRC(gemm_seg(t.act_e, w->w2[e], t.y_e, d.T, d.D, d.H, w->b2[e], 0, nullptr, t.seg + 2 * e, nullptr, st));
  TA_LAUNCH(moe_combine_kernel, dim3(ta_cdiv(d.T, 4)), dim3(256), 0, st, y, t.y_e, t.slot_of, t.topw, d.T, d.D);
  if (aux) {
    if (training) TA_LAUNCH(moe_aux_kernel, dim3(1), dim3(64), 0, st, t.psum, t.zsum, aux, d.T, E, w->aux_coef, w->z_coef);
    else if (hipMemsetAsync(aux, 0, 4, st) != hipSuccess) return TA_ERR_LAUNCH;
  }
  TA_CHECK_LAUNCH();
  return TA_OK;
}

#define CS_CH 128   // row chunks of the bias-gradient column sums: C / 256 x 128 workgroups (16 left a 4-16-workgroup grid: 38 us each)
static int moe_backward_impl(const ta_moe_weights* w, const void* x, int B, int S, const float* dy, float d_aux,
                             const float* d_aux_dev, const float* noise, int training, const void* tape, float* d_norm_w,
                             float* d_router_w, float* const* dW1, float* const* db1, float* const* dW2,
                             float* const* db2, void* ws, long ws_bytes, hipStream_t st) {
  const MoeDims d = moe_dims(w, B, S);
  if (B <= 0 || d.N <= 0) return TA_OK;
  MoeTape t = moe_tape(w, B, S, (void*)tape);
  MoeWs s = moe_ws(w, B, S, ws);
  if ((long)s.bytes > ws_bytes) return TA_ERR_ARG;
  const int E = d.E;
  auto zero = [&](void* p, size_t n) { return hipMemsetAsync(p, 0, n, st) == hipSuccess; };
  if (!zero(d_norm_w, (size_t)d.In * 4) || !zero(d_router_w, (size_t)E * d.In * 4)) return TA_ERR_LAUNCH;
  for (int e = 0; e <= E; ++e)
    if (!zero(db1[e], (size_t)d.H * 4) || !zero(db2[e], (size_t)d.D * 4)) return TA_ERR_LAUNCH;
  TA_LAUNCH(moe_combine_bwd_kernel, dim3(ta_cdiv(d.T, 4)), dim3(256), 0, st, dy, t.y_e, t.slot_of, t.topw, s.dy_slot, s.dtopw,
            s.dout_b, d.T, d.D);
  // ---- shared expert
  TA_LAUNCH(colsum_seg_kernel, dim3(ta_cdiv(d.D, 256), CS_CH), dim3(256), 0, st, s.dout_b, d.D, (const int*)nullptr, d.T, db2[E], CS_CH);
  MRC(ta_transpose_to_bf16(s.dout_b, 0, d.D, 0, 0, s.doutT, d.Tp, d.T, d.D, st));
  MRC(ta_transpose_to_bf16(t.act_s, 0, d.H, 0, 0, s.actsT, d.Tp, d.T, d.H, st));
  MRC(gemm_plain(s.doutT, s.actsT, dW2[E], d.D, d.H, d.Tp, nullptr, 0, moe_splits(d.D, d.H, d.Tp), s.skws, st));
  MRC(gemm_plain(s.dout_b, w->w2_t[E], s.dact_s, d.T, d.H, d.D, nullptr, 1, 1, nullptr, st));
  TA_LAUNCH(gelu_bwd_bf16_kernel, dim3(ew((long)d.T * d.H / 4)), dim3(256), 0, st, s.dact_s, t.h_s, s.dh_s, (long)d.T * d.H / 4);
  TA_LAUNCH(colsum_seg_kernel, dim3(ta_cdiv(d.H, 256), CS_CH), dim3(256), 0, st, s.dh_s, d.H, (const int*)nullptr, d.T, db1[E], CS_CH);
  MRC(ta_transpose_to_bf16(s.dh_s, 0, d.H, 0, 0, s.dhsT, d.Tp, d.T, d.H, st));
  MRC(ta_transpose_to_bf16(t.xn, 0, d.In, 0, 0, s.xnT, d.Tp, d.T, d.In, st));
  MRC(gemm_plain(s.dhsT, s.xnT, dW1[E], d.H, d.In, d.Tp, nullptr, 0, moe_splits(d.H, d.In, d.Tp), s.skws, st));
  MRC(gemm_plain(s.dh_s, w->w1_t[E], s.dxn_sh, d.T, d.In, d.H, nullptr, 0, 1, nullptr, st));
  // ---- routed experts
  dim3 tb(256);
  TA_LAUNCH(slot_transpose_kernel, dim3(ta_cdiv(d.D, 64), d.Smax / 64), tb, 0, st, s.dy_slot, d.D, t.perm, 0, s.dyT, d.Smax);
  TA_LAUNCH(slot_transpose_kernel, dim3(ta_cdiv(d.H, 64), d.Smax / 64), tb, 0, st, t.act_e, d.H, t.perm, 0, s.acteT, d.Smax);
  const long s_w2t = expert_stride<bf16_t>(w->w2_t, E), s_w1t = expert_stride<bf16_t>(w->w1_t, E);
  const long s_dw2 = expert_stride<float>((const void* const*)dW2, E), s_dw1 = expert_stride<float>((const void* const*)dW1, E);
  const bool grp = grouped_enabled();
  for (int e = 0; e < E; ++e)
    TA_LAUNCH(colsum_seg_kernel, dim3(ta_cdiv(d.D, 256), CS_CH), tb, 0, st, s.dy_slot, d.D, t.seg + 2 * e, 0, db2[e], CS_CH);
  // dW2[e] = dy_e^T act_e: the contraction runs over expert e's (64-aligned) slot range -> the K-slice grouped form
  if (grp && s_dw2 > 0) MRC(ta_gemm_bf16_nt_grouped(s.dyT, s.acteT, dW2[0], d.D, d.H, d.Smax, nullptr, 0, 0, nullptr, nullptr, t.kr, E, 0, s_dw2, st));
  else for (int e = 0; e < E; ++e) MRC(gemm_seg(s.dyT, s.acteT, dW2[e], d.D, d.H, d.Smax, nullptr, 0, nullptr, nullptr, t.kr + 2 * e, st));
  if (grp && s_w2t > 0) MRC(ta_gemm_bf16_nt_grouped(s.dy_slot, w->w2_t[0], s.dact_e, 2 * d.T, d.H, d.D, nullptr, 0, 1, nullptr, t.seg, nullptr, E, s_w2t, 0, st));
  else for (int e = 0; e < E; ++e) MRC(gemm_seg(s.dy_slot, w->w2_t[e], s.dact_e, d.T, d.H, d.D, nullptr, 1, nullptr, t.seg + 2 * e, nullptr, st));
  TA_LAUNCH(gelu_bwd_bf16_kernel, dim3(ew((long)d.Smax * d.H / 4)), tb, 0, st, s.dact_e, t.h_e, s.dh_e, (long)d.Smax * d.H / 4);
  TA_LAUNCH(slot_transpose_kernel, dim3(ta_cdiv(d.H, 64), d.Smax / 64), tb, 0, st, s.dh_e, d.H, t.perm, 0, s.dheT, d.Smax);
  TA_LAUNCH(slot_transpose_kernel, dim3(ta_cdiv(d.In, 64), d.Smax / 64), tb, 0, st, t.xn, d.In, t.perm, 1, s.xngT, d.Smax);
  for (int e = 0; e < E; ++e)
    TA_LAUNCH(colsum_seg_kernel, dim3(ta_cdiv(d.H, 256), CS_CH), tb, 0, st, s.dh_e, d.H, t.seg + 2 * e, 0, db1[e], CS_CH);
  if (grp && s_dw1 > 0) MRC(ta_gemm_bf16_nt_grouped(s.dheT, s.xngT, dW1[0], d.H, d.In, d.Smax, nullptr, 0, 0, nullptr, nullptr, t.kr, E, 0, s_dw1, st));
  else for (int e = 0; e < E; ++e) MRC(gemm_seg(s.dheT, s.xngT, dW1[e], d.H, d.In, d.Smax, nullptr, 0, nullptr, nullptr, t.kr + 2 * e, st));
  if (grp && s_w1t > 0) MRC(ta_gemm_bf16_nt_grouped(s.dh_e, w->w1_t[0], s.dxn_slot, 2 * d.T, d.In, d.H, nullptr, 0, 0, nullptr, t.seg, nullptr, E, s_w1t, 0, st));
  else for (int e = 0; e < E; ++e) MRC(gemm_seg(s.dh_e, w->w1_t[e], s.dxn_slot, d.T, d.In, d.H, nullptr, 0, nullptr, t.seg + 2 * e, nullptr, st));
  // ---- router and input norm
  TA_LAUNCH(moe_router_bwd_kernel, dim3(ta_cdiv(d.T, 256)), tb, 0, st, s.dtopw, t.probs, t.topi, t.topraw, t.lse,
            training ? noise : nullptr, t.psum, s.dlogits, d.T, E, d_aux, d_aux_dev, w->aux_coef, w->z_coef, training);
  TA_LAUNCH(moe_router_dw_kernel, dim3(ta_cdiv(d.In, 256), 64), tb, 0, st, s.dlogits, t.xn, d_router_w, d.T, d.In, E, 64);
  TA_LAUNCH(moe_norm_bwd_kernel, dim3(ta_cdiv(d.In, 256), 128), tb, 0, st, s.dxn_sh, s.dxn_slot, t.slot_of, s.dlogits, w->router_w,
            (const bf16_t*)x, (long)S * w->enc_dim, d.N, (long)d.In, t.rstd, d_norm_w, d.T, d.In, E, 128);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

// round 4: the share of d(norm.weight) / d(router.weight) that comes from the AUXILIARY losses alone (d_aux * d aux / d .): the three
// tail kernels of the backward with no expert-path inputs.  The trainer keeps it in a shadow of the flat gradient buffer: HF adds the
// auxiliary term at FULL weight next to a token-normalised CE, so with gradients of CE-sum + aux in the flat buffer and the global
// token count N known only after the all-reduce, the optimizer's division by N is undone for this share alone
// (g += (N - 1) * shadow) -- which is what lets MoE run on ONE collective like every other configuration.
extern "C" int ta_moe_router_aux_grads(const ta_moe_weights* w, const void* x, int B, int S, const float* d_aux_dev, const float* noise,
                                       int training, const void* tape, float* d_norm_w_aux, float* d_router_w_aux, void* ws,
                                       long ws_bytes, hipStream_t st) {
  const MoeDims d = moe_dims(w, B, S);
  if (B <= 0 || d.N <= 0) return TA_OK;
  if (!d_aux_dev || !d_norm_w_aux || !d_router_w_aux) return TA_ERR_ARG;
  MoeTape t = moe_tape(w, B, S, (void*)tape);
  MoeWs s = moe_ws(w, B, S, ws);
  if ((long)s.bytes > ws_bytes) return TA_ERR_ARG;
  const int E = d.E;
  if (hipMemsetAsync(d_norm_w_aux, 0, (size_t)d.In * 4, st) != hipSuccess || hipMemsetAsync(d_router_w_aux, 0, (size_t)E * d.In * 4, st) != hipSuccess)
    return TA_ERR_LAUNCH;
  dim3 tb(256);
  TA_LAUNCH(moe_router_bwd_kernel, dim3(ta_cdiv(d.T, 256)), tb, 0, st, (const float*)nullptr, t.probs, t.topi, t.topraw, t.lse,
            training ? noise : nullptr, t.psum, s.dlogits, d.T, E, 0.f, d_aux_dev, w->aux_coef, w->z_coef, training);
  TA_LAUNCH(moe_router_dw_kernel, dim3(ta_cdiv(d.In, 256), 64), tb, 0, st, s.dlogits, t.xn, d_router_w_aux, d.T, d.In, E, 64);
  TA_LAUNCH(moe_norm_bwd_kernel, dim3(ta_cdiv(d.In, 256), 128), tb, 0, st, (const float*)nullptr, (const float*)nullptr, t.slot_of, s.dlogits,
            w->router_w, (const bf16_t*)x, (long)S * w->enc_dim, d.N, (long)d.In, t.rstd, d_norm_w_aux, d.T, d.In, E, 128);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_moe_projector_backward(const ta_moe_weights* w, const void* x, int B, int S, const float* dy, float d_aux,
                                         const float* noise, int training, const void* tape, float* d_norm_w,
                                         float* d_router_w, float* const* dW1, float* const* db1, float* const* dW2,
                                         float* const* db2, void* ws, long ws_bytes, hipStream_t st) {
  return moe_backward_impl(w, x, B, S, dy, d_aux, nullptr, noise, training, tape, d_norm_w, d_router_w, dW1, db1, dW2, db2, ws,
                           ws_bytes, st);
}
extern "C" int ta_moe_projector_backward_dev(const ta_moe_weights* w, const void* x, int B, int S, const float* dy,
                                             const float* d_aux_dev, const float* noise, int training, const void* tape,
                                             float* d_norm_w, float* d_router_w, float* const* dW1, float* const* db1,
                                             float* const* dW2, float* const* db2, void* ws, long ws_bytes, hipStream_t st) {
  if (!d_aux_dev) return TA_ERR_ARG;
  return moe_backward_impl(w, x, B, S, dy, 0.f, d_aux_dev, noise, training, tape, d_norm_w, d_router_w, dW1, db1, dW2, db2, ws,
                           ws_bytes, st);
}
