// ta355 element-wise / data-movement kernels (all HBM-bound; 8-16 B per lane, grid-stride).
#include "common.h"

static inline int ew_blocks(long n, int per_block = 256) {
  long b = (n + per_block - 1) / per_block;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// ---------------------------------------------------------------------------- SwiGLU (Qwen3MLP, modeling_qwen3.py:80-83)
// gu bf16 [M, 2F] = [gate | up]  ->  act bf16 [M, F] = silu(gate) * up
__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act, long M, int F) {
  const long n4 = M * (F / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long m = i / (F / 4); const int c = (int)(i % (F / 4)) * 4;
    const uint2 gv = *(const uint2*)(gu + m * 2 * F + c);
    const uint2 uv = *(const uint2*)(gu + m * 2 * F + F + c);
    const float g[4] = {bf2f(gv.x & 0xffff), bf2f(gv.x >> 16), bf2f(gv.y & 0xffff), bf2f(gv.y >> 16)};
    const float u[4] = {bf2f(uv.x & 0xffff), bf2f(uv.x >> 16), bf2f(uv.y & 0xffff), bf2f(uv.y >> 16)};
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = g[j] / (1.f + __expf(-g[j])) * u[j];
    uint2 o; o.x = pack2bf(a[0], a[1]); o.y = pack2bf(a[2], a[3]);
    *(uint2*)(act + m * F + c) = o;
  }
}
// dact bf16 [M,F], gu bf16 [M,2F] -> dgu bf16 [M,2F]
__global__ void swiglu_bwd_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ gu,
                                  bf16_t* __restrict__ dgu, long M, int F) {
  const long n4 = M * (F / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long m = i / (F / 4); const int c = (int)(i % (F / 4)) * 4;
    const uint2 gv = *(const uint2*)(gu + m * 2 * F + c);
    const uint2 uv = *(const uint2*)(gu + m * 2 * F + F + c);
    const uint2 dv = *(const uint2*)(dact + m * F + c);
    const float g[4] = {bf2f(gv.x & 0xffff), bf2f(gv.x >> 16), bf2f(gv.y & 0xffff), bf2f(gv.y >> 16)};
    const float u[4] = {bf2f(uv.x & 0xffff), bf2f(uv.x >> 16), bf2f(uv.y & 0xffff), bf2f(uv.y >> 16)};
    const float d[4] = {bf2f(dv.x & 0xffff), bf2f(dv.x >> 16), bf2f(dv.y & 0xffff), bf2f(dv.y >> 16)};
    float dg[4], du[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sg = 1.f / (1.f + __expf(-g[j]));
      dg[j] = d[j] * u[j] * (sg * (1.f + g[j] * (1.f - sg)));
      du[j] = d[j] * g[j] * sg;
    }
    uint2 o; o.x = pack2bf(dg[0], dg[1]); o.y = pack2bf(dg[2], dg[3]);
    *(uint2*)(dgu + m * 2 * F + c) = o;
    o.x = pack2bf(du[0], du[1]); o.y = pack2bf(du[2], du[3]);
    *(uint2*)(dgu + m * 2 * F + F + c) = o;
  }
}

// ---------------------------------------------------------------------------- casts / transposes
__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = ((const float4*)x)[i];
    uint2 o; o.x = pack2bf(v.x, v.y); o.y = pack2bf(v.z, v.w);
    ((uint2*)y)[i] = o;
  }
}
// in [R, C] (f32 or bf16, row stride ld_in) -> out bf16 [C, ld_out] (ld_out >= R; columns R..ld_out-1 are zeroed)
template <typename T>
__global__ __launch_bounds__(256) void transpose_to_bf16_kernel(const T* __restrict__ in, long ld_in, long in_bs, int in_rpb,
                                                                bf16_t* __restrict__ out, long ld_out, int R, int C) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    bf16_t v = 0;
    if (r < R && c < C) {
      const long off = (long)(r / in_rpb) * in_bs + (long)(r % in_rpb) * ld_in + c;
      if constexpr (sizeof(T) == 4) v = f2bf(((const float*)in)[off]); else v = ((const bf16_t*)in)[off];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < C && r < ld_out) out[(long)c * ld_out + r] = tile[tx][i];
  }
}

// Same contract, 16-byte accesses on both sides (needs ld_in, in_bs, C, ld_out multiples of 8): a 64 x 64 tile goes
// through LDS; a thread loads 8 consecutive columns of one row and stores 8 consecutive rows of one column.
template <typename T>
__global__ __launch_bounds__(256) void transpose_to_bf16_vec_kernel(const T* __restrict__ in, long ld_in, long in_bs, int in_rpb,
                                                                    bf16_t* __restrict__ out, long ld_out, int R, int C) {
  constexpr int P = 72;                               // LDS row pitch (elements): 144 B, 16-B aligned
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * P];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ch = tid + it * 256, r = ch >> 3, cc = ch & 7;
    const int gr = r0 + r, gc = c0 + cc * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gr < R && gc < C) {
      const long off = (long)(gr / in_rpb) * in_bs + (long)(gr % in_rpb) * ld_in + gc;
      if constexpr (sizeof(T) == 4) {
        const float4 a = *(const float4*)((const float*)in + off), b = *(const float4*)((const float*)in + off + 4);
        v = make_uint4(pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w));
      } else {
        v = *(const uint4*)((const bf16_t*)in + off);
      }
    }
    *(uint4*)(tile + r * P + cc * 8) = v;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ch = tid + it * 256, c = ch >> 3, rc = ch & 7;
    const int gc = c0 + c, gr = r0 + rc * 8;
    if (gc < C && gr < ld_out) {
      union { uint4 v; bf16_t e[8]; } u;
#pragma unroll
      for (int j = 0; j < 8; ++j) u.e[j] = tile[(rc * 8 + j) * P + c];
      *(uint4*)(out + (long)gc * ld_out + gr) = u.v;
    }
  }
}

// ---------------------------------------------------------------------------- encoder input layout
// feats f32 [B, C, T] -> bf16 time-major [B, T+2, C] with zero rows 0 and T+1 (conv padding=1)
__global__ __launch_bounds__(256) void feats_to_tm_kernel(const float* __restrict__ f, bf16_t* __restrict__ out, int C, int T) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < T) ? f[((long)b * C + c) * T + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int t = t0 + i, c = c0 + tx;
    if (t < T && c < C) out[((long)b * (T + 2) + t + 1) * C + c] = f2bf(tile[tx][i]);
  }
  if (blockIdx.x == 0 && ty == 0 && c0 + tx < C) {
    out[((long)b * (T + 2)) * C + c0 + tx] = 0;
    out[((long)b * (T + 2) + T + 1) * C + c0 + tx] = 0;
  }
}
// zero the two padding rows of a [B, T+2, C] bf16 buffer
__global__ void zero_pad_rows_kernel(bf16_t* __restrict__ buf, int T, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    buf[((long)b * (T + 2)) * C + c] = 0;
    buf[((long)b * (T + 2) + T + 1) * C + c] = 0;
  }
}

// ---------------------------------------------------------------------------- <audio> placeholder bookkeeping
// tiny_audio/asr_modeling.py:27-44,511-515: the r-th <audio> position (row-major over [B,L]) receives packed row r,
// packed = first counts[i] projector rows of sample i (zero rows when counts[i] > N).
// src_row[pos] = i*N + n (>=0), -2 -> zero row, -1 -> not an audio position.  Single block.
__global__ __launch_bounds__(1024) void audio_index_kernel(const long* __restrict__ ids, const long* __restrict__ counts,
                                                           int* __restrict__ src_row, int B, int L, int N, long audio_id) {
  __shared__ int scan[1024];
  __shared__ int carry;
  __shared__ long cum[1025];
  const int tid = threadIdx.x;
  if (tid == 0) {
    carry = 0; cum[0] = 0;
    for (int i = 0; i < B && i < 1024; ++i) cum[i + 1] = cum[i] + counts[i];
  }
  __syncthreads();
  const int total = B * L;
  for (int base = 0; base < total; base += 1024) {
    const int p = base + tid;
    const int flag = (p < total && ids[p] == audio_id) ? 1 : 0;
    scan[tid] = flag;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int v = tid >= o ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    if (p < total) {
      int out = -1;
      if (flag) {
        const long rank = carry + scan[tid] - 1;
        int lo = 0, hi = B;                      // largest i with cum[i] <= rank
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (cum[mid] <= rank) lo = mid; else hi = mid; }
        const long n = rank - cum[lo];
        out = (rank < cum[B] && n < N) ? (int)(lo * N + n) : -2;
      }
      src_row[p] = out;
    }
    __syncthreads();
    if (tid == 1023) carry += scan[1023];
    __syncthreads();
  }
}
// x0[p,:] = src_row[p] >= 0 ? audio[src_row[p],:] : (src_row[p] == -2 ? 0 : emb[ids[p],:])     (f32, D % 4 == 0)
__global__ __launch_bounds__(256) void embed_scatter_kernel(const long* __restrict__ ids, const int* __restrict__ src_row,
                                                            const float* __restrict__ emb, const float* __restrict__ audio,
                                                            float* __restrict__ x0, bf16_t* __restrict__ x0b, int n_rows, int D,
                                                            long vocab) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int s = src_row ? src_row[row] : -1;
  long id = ids[row]; if (id < 0) id = 0; if (id >= vocab) id = vocab - 1;
  const float4* src = s >= 0 ? (const float4*)(audio + (long)s * D) : (const float4*)(emb + id * D);
  for (int c = lane; c < D / 4; c += 64) {
    const float4 v = (s == -2) ? make_float4(0.f, 0.f, 0.f, 0.f) : src[c];
    if (x0) ((float4*)(x0 + (long)row * D))[c] = v;
    if (x0b) { uint2 o; o.x = pack2bf(v.x, v.y); o.y = pack2bf(v.z, v.w); ((uint2*)(x0b + (long)row * D))[c] = o; }
  }
}
// backward of the scatter: d_audio[src_row[p],:] = dx0[p,:]  (d_audio pre-zeroed; each row written at most once)
__global__ __launch_bounds__(256) void audio_grad_gather_kernel(const int* __restrict__ src_row, const float* __restrict__ dx0,
                                                                float* __restrict__ d_audio, int n_rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int s = src_row[row];
  if (s < 0) return;
  const int lane = threadIdx.x & 63;
  for (int c = lane; c < D / 4; c += 64) ((float4*)(d_audio + (long)s * D))[c] = ((const float4*)(dx0 + (long)row * D))[c];
}

// generic row gather/scatter with an index list (labelled rows of the LM head)
// out[i,:] = in[idx[i],:]   (bf16 rows, D % 8 == 0)
__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const bf16_t* __restrict__ in, const int* __restrict__ idx,
                                                               bf16_t* __restrict__ out, int n, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const long s = idx[row];
  for (int c = lane; c < D / 8; c += 64) ((uint4*)(out + (long)row * D))[c] = ((const uint4*)(in + s * D))[c];
}
// out[idx[i],:] = in[i,:]   (f32 rows; out pre-zeroed)
__global__ __launch_bounds__(256) void scatter_rows_f32_kernel(const float* __restrict__ in, const int* __restrict__ idx,
                                                               float* __restrict__ out, int n, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const long s = idx[row];
  for (int c = lane; c < D / 4; c += 64) ((float4*)(out + s * D))[c] = ((const float4*)(in + (long)row * D))[c];
}

// frame-keep mask for the encoder output (tiny_audio/asr_modeling.py:458-479): keep[i] = U(seed,i) < keep_prob
__global__ void bernoulli_keep_kernel(float* __restrict__ keep, long n, float keep_prob, unsigned long long seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);   // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    keep[i] = u < keep_prob ? 1.f : 0.f;
  }
}

// ----------------------------------------------------------------------------- C-ABI
extern "C" int ta_swiglu_fwd(const void* gu, void* act, long M, int F, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if (F % 4) return TA_ERR_ARG;
  TA_LAUNCH(swiglu_fwd_kernel, dim3(ew_blocks(M * (F / 4))), dim3(256), 0, st, (const bf16_t*)gu, (bf16_t*)act, M, F);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_swiglu_bwd(const void* dact, const void* gu, void* dgu, long M, int F, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if (F % 4) return TA_ERR_ARG;
  TA_LAUNCH(swiglu_bwd_kernel, dim3(ew_blocks(M * (F / 4))), dim3(256), 0, st, (const bf16_t*)dact,
                     (const bf16_t*)gu, (bf16_t*)dgu, M, F);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_cast_f32_bf16(const float* x, void* y, long n, hipStream_t st) {
  if (n <= 0) return TA_OK;
  if (n % 4) return TA_ERR_ARG;
  TA_LAUNCH(cast_f32_bf16_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, st, x, (bf16_t*)y, n / 4);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_transpose_to_bf16(const void* in, int in_is_f32, long ld_in, long in_bs, int in_rpb, void* out,
                                    long ld_out, int R, int C, hipStream_t st) {
  if (R <= 0 || C <= 0) return TA_OK;
  if (ld_out < R) return TA_ERR_ARG;
  dim3 grid(ta_cdiv(C, 64), ta_cdiv(ld_out, 64));
  if (in_rpb <= 0) in_rpb = R;
  if (!((ld_in | in_bs | (long)C | ld_out) & 7) && !(((uintptr_t)in | (uintptr_t)out) & 15)) {
    if (in_is_f32)
      TA_LAUNCH((transpose_to_bf16_vec_kernel<float>), grid, dim3(256), 0, st, (const float*)in, ld_in, in_bs, in_rpb,
                (bf16_t*)out, ld_out, R, C);
    else
      TA_LAUNCH((transpose_to_bf16_vec_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)in, ld_in, in_bs, in_rpb,
                (bf16_t*)out, ld_out, R, C);
    TA_CHECK_LAUNCH(); return TA_OK;
  }
  if (in_is_f32)
    TA_LAUNCH((transpose_to_bf16_kernel<float>), grid, dim3(256), 0, st, (const float*)in, ld_in, in_bs, in_rpb,
                       (bf16_t*)out, ld_out, R, C);
  else
    TA_LAUNCH((transpose_to_bf16_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)in, ld_in, in_bs, in_rpb,
                       (bf16_t*)out, ld_out, R, C);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_feats_to_time_major(const float* feats, void* out, int B, int C, int T, hipStream_t st) {
  if (B <= 0 || T <= 0) return TA_OK;
  TA_LAUNCH(feats_to_tm_kernel, dim3(ta_cdiv(T, 64), ta_cdiv(C, 64), B), dim3(256), 0, st, feats, (bf16_t*)out, C, T);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_zero_pad_rows(void* buf, int B, int T, int C, hipStream_t st) {
  if (B <= 0) return TA_OK;
  TA_LAUNCH(zero_pad_rows_kernel, dim3(B), dim3(256), 0, st, (bf16_t*)buf, T, C);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_audio_index(const long* ids, const long* counts, int* src_row, int B, int L, int N, long audio_id,
                              hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  if (B > 1024) return TA_ERR_ARG;
  TA_LAUNCH(audio_index_kernel, dim3(1), dim3(1024), 0, st, ids, counts, src_row, B, L, N, audio_id);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_embed_scatter(const long* ids, const int* src_row, const float* emb, const float* audio, float* x0,
                                void* x0_bf16, int n_rows, int D, long vocab, hipStream_t st) {
  if (n_rows <= 0) return TA_OK;
  if (D % 4) return TA_ERR_ARG;
  TA_LAUNCH(embed_scatter_kernel, dim3(ta_cdiv(n_rows, 4)), dim3(256), 0, st, ids, src_row, emb, audio, x0,
                     (bf16_t*)x0_bf16, n_rows, D, vocab);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_audio_grad_gather(const int* src_row, const float* dx0, float* d_audio, int n_rows, int D, hipStream_t st) {
  if (n_rows <= 0) return TA_OK;
  if (D % 4) return TA_ERR_ARG;
  TA_LAUNCH(audio_grad_gather_kernel, dim3(ta_cdiv(n_rows, 4)), dim3(256), 0, st, src_row, dx0, d_audio, n_rows, D);
  TA_CHECK_LAUNCH(); return TA_OK;
}
// input-lookup share of the embedding gradient (trainable LM): one wave per token row, float4 per lane-step
__global__ __launch_bounds__(256) void embed_grad_scatter_kernel(const long* __restrict__ ids, const int* __restrict__ src_row,
                                                                 const float* __restrict__ dx0, float* __restrict__ dembed,
                                                                 int n_rows, int D, long vocab) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  if (src_row && src_row[row] != -1) return;           // an <audio> position (projector row, or the -2 zero row of a count
                                                       // beyond the projector's length): its embedding row was overwritten
  const long id = ids[row];
  if (id < 0 || id >= vocab) return;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 g = *(const float4*)(dx0 + (long)row * D + c);
    float* dst = dembed + id * D + c;
    unsafeAtomicAdd(dst, g.x); unsafeAtomicAdd(dst + 1, g.y); unsafeAtomicAdd(dst + 2, g.z); unsafeAtomicAdd(dst + 3, g.w);
  }
}
extern "C" int ta_embed_grad_scatter(const long* ids, const int* src_row, const float* dx0, float* dembed, int n_rows, int D,
                                     long vocab, hipStream_t st) {
  if (n_rows <= 0) return TA_OK;
  if (D % 4) return TA_ERR_ARG;
  TA_LAUNCH(embed_grad_scatter_kernel, dim3(ta_cdiv(n_rows, 4)), dim3(256), 0, st, ids, src_row, dx0, dembed, n_rows, D, vocab);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_gather_rows_bf16(const void* in, const int* idx, void* out, int n, int D, hipStream_t st) {
  if (n <= 0) return TA_OK;
  if (D % 8) return TA_ERR_ARG;
  TA_LAUNCH(gather_rows_bf16_kernel, dim3(ta_cdiv(n, 4)), dim3(256), 0, st, (const bf16_t*)in, idx, (bf16_t*)out, n, D);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_scatter_rows_f32(const float* in, const int* idx, float* out, int n, int D, hipStream_t st) {
  if (n <= 0) return TA_OK;
  if (D % 4) return TA_ERR_ARG;
  TA_LAUNCH(scatter_rows_f32_kernel, dim3(ta_cdiv(n, 4)), dim3(256), 0, st, in, idx, out, n, D);
  TA_CHECK_LAUNCH(); return TA_OK;
}
extern "C" int ta_bernoulli_keep(float* keep, long n, float keep_prob, unsigned long long seed, hipStream_t st) {
  if (n <= 0) return TA_OK;
  TA_LAUNCH(bernoulli_keep_kernel, dim3(ew_blocks(n)), dim3(256), 0, st, keep, n, keep_prob, seed);
  TA_CHECK_LAUNCH(); return TA_OK;
}

// ---------------------------------------------------------------------------- instrumentation: the footprint of a collective
// `workgroups` x 256 threads stay resident for `micros` microseconds, streaming `buf` (read + write back, 16 B per thread per pass) the
// whole time: what an RCCL ring all-reduce looks like to the kernels beside it (a few channel workgroups, copy traffic, no LDS).
// Used by scripts/allreduce_footprint.py to choose the N > 1 default (overlapped vs synchronous) on a one-GPU box.
__global__ __launch_bounds__(256) void occupy_kernel(uint4* __restrict__ buf, long n16, long ticks) {
  const unsigned long long t0 = wall_clock64();
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  uint4 acc = make_uint4(0, 0, 0, 0);
  while ((long)(wall_clock64() - t0) < ticks) {
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
      if (i >= n16) i -= (i / n16) * n16;
      const uint4 v = buf[i];
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
      buf[i] = v;
      i += stride;
    }
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) buf[0] = acc;     // (keeps the loads alive)
}
extern "C" int ta_debug_occupy(int workgroups, double micros, void* buf, long bytes, hipStream_t st) {
  if (workgroups <= 0 || micros <= 0 || !buf || bytes < 4096) return TA_ERR_ARG;
  int dev = 0, rate_khz = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev);     // kHz of wall_clock64()
  if (rate_khz <= 0) rate_khz = 100000;
  TA_LAUNCH(occupy_kernel, dim3(workgroups), dim3(256), 0, st, (uint4*)buf, bytes / 16, (long)(micros * rate_khz / 1000.0));
  TA_CHECK_LAUNCH(); return TA_OK;
}
