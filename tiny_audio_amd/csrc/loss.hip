// ta355 shifted cross-entropy over the (padded) vocabulary, HBM-bound.
// ForCausalLMLoss / fixed_cross_entropy, TF:loss/loss_utils.py:33-71: logits.float(), ignore_index -100,
// reduction "sum" / num_items (== "mean" when num_items is the number of valid targets).
//
// One 256-thread block per row: online (max, sum-exp) in one read pass, then one pass writing
// dlogits = (softmax - onehot) * scale as bf16 (the A operand of the dH = dlogits x E GEMM).
// Columns >= V (padding up to the GEMM-friendly Vp) are excluded from the softmax and get dlogits 0.
#include "common.h"

template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *(const float4*)p; }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
  const uint2 v = *(const uint2*)p;
  return make_float4(bf2f(v.x & 0xffff), bf2f(v.x >> 16), bf2f(v.y & 0xffff), bf2f(v.y >> 16));
}

template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const T* __restrict__ logits, long ldl, const int* __restrict__ rows,
                                                         const long* __restrict__ targets, int V, float scale,
                                                         float* __restrict__ nll_out, float* __restrict__ loss_accum,
                                                         bf16_t* __restrict__ dlogits, long ldd) {
  __shared__ float red_m[4], red_s[4];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long tgt = targets[i];
  const T* row = logits + (long)(rows ? rows[i] : i) * ldl;
  const int nv = V / 4;          // V % 4 handled by the scalar tail below
  float m = -INFINITY, s = 0.f;
  for (int c = tid; c < nv; c += 256) {
    const float4 v = ld4<T>(row + c * 4);
    const float mx = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    if (mx > m) { s *= __expf(m - mx); m = mx; }
    s += __expf(v.x - m) + __expf(v.y - m) + __expf(v.z - m) + __expf(v.w - m);
  }
  for (int c = nv * 4 + tid; c < V; c += 256) {
    const float v = sizeof(T) == 4 ? (float)((const float*)row)[c] : bf2f(((const bf16_t*)row)[c]);
    if (v > m) { s *= __expf(m - v); m = v; }
    s += __expf(v - m);
  }
  // wave then block combine of (m, s)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
    m = mn;
  }
  if (lane == 0) { red_m[wave] = m; red_s[wave] = s; }
  __syncthreads();
  float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
  float S = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) S += red_m[w] == -INFINITY ? 0.f : red_s[w] * __expf(red_m[w] - M);
  const float lse = M + __logf(S);
  const bool valid = tgt >= 0 && tgt < V;
  if (tid == 0) {
    float nll = 0.f;
    if (valid) {
      const float zt = sizeof(T) == 4 ? (float)((const float*)row)[tgt] : bf2f(((const bf16_t*)row)[tgt]);
      nll = lse - zt;
      if (loss_accum) atomicAdd(loss_accum, nll * scale);
    }
    if (nll_out) nll_out[i] = nll;
  }
  if (!dlogits) return;
  bf16_t* drow = dlogits + (long)i * ldd;
  const float sc = valid ? scale : 0.f;
  const int nvd = (int)(ldd / 4);
  for (int c = tid; c < nvd; c += 256) {
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (c * 4 + 3 < V) {
      const float4 v = ld4<T>(row + c * 4);
      o[0] = __expf(v.x - lse) * sc; o[1] = __expf(v.y - lse) * sc; o[2] = __expf(v.z - lse) * sc; o[3] = __expf(v.w - lse) * sc;
    } else {
      for (int j = 0; j < 4; ++j) {
        const int cc = c * 4 + j;
        if (cc < V) {
          const float v = sizeof(T) == 4 ? (float)((const float*)row)[cc] : bf2f(((const bf16_t*)row)[cc]);
          o[j] = __expf(v - lse) * sc;
        }
      }
    }
    if (valid && (long)c * 4 <= tgt && tgt < (long)c * 4 + 4) o[tgt - (long)c * 4] -= sc;
    uint2 w; w.x = pack2bf(o[0], o[1]); w.y = pack2bf(o[2], o[3]);
    *(uint2*)(drow + c * 4) = w;
  }
}

// shifted labels -> compact list of (row index, target) for the valid targets; n_out[0] = count (single block)
// labels [B, L] int64: target of position (b,l) is labels[b,l+1] (l+1 < L), ignore -100.
__global__ __launch_bounds__(1024) void label_rows_kernel(const long* __restrict__ labels, int B, int L,
                                                          int* __restrict__ rows, long* __restrict__ targets,
                                                          int* __restrict__ n_out) {
  __shared__ int scan[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  const int total = B * L;
  for (int base = 0; base < total; base += 1024) {
    const int p = base + tid;
    long t = -100;
    if (p < total) { const int l = p % L; if (l + 1 < L) t = labels[p + 1]; }
    const int flag = (t != -100) ? 1 : 0;
    scan[tid] = flag;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int v = tid >= o ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    if (flag) { const int k = carry + scan[tid] - 1; rows[k] = p; targets[k] = t; }
    __syncthreads();
    if (tid == 1023) carry += scan[1023];
    __syncthreads();
  }
  if (tid == 0) n_out[0] = carry;
}

extern "C" int ta_cross_entropy(const void* logits, int logits_bf16, long ldl, const int* rows, const long* targets,
                                int n, int V, float scale, float* nll, float* loss_accum, void* dlogits_bf16, long ldd,
                                hipStream_t st) {
  if (n <= 0) return TA_OK;
  if ((ldl % 4) || (dlogits_bf16 && (ldd % 4))) return TA_ERR_ARG;
  if (logits_bf16)
    TA_LAUNCH((ce_fwd_bwd_kernel<bf16_t>), dim3(n), dim3(256), 0, st, (const bf16_t*)logits, ldl, rows, targets, V,
                       scale, nll, loss_accum, (bf16_t*)dlogits_bf16, ldd);
  else
    TA_LAUNCH((ce_fwd_bwd_kernel<float>), dim3(n), dim3(256), 0, st, (const float*)logits, ldl, rows, targets, V,
                       scale, nll, loss_accum, (bf16_t*)dlogits_bf16, ldd);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_label_rows(const long* labels, int B, int L, int* rows, long* targets, int* n_out, hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  TA_LAUNCH(label_rows_kernel, dim3(1), dim3(1024), 0, st, labels, B, L, rows, targets, n_out);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
