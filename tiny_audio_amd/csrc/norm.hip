// ta355 row-normalisation kernels (HBM-bound): one 64-lane wave per row, the row held in
// registers (float4 per lane, up to 5120 columns), wave-shuffle reductions, 16-byte accesses.
//
//   layernorm_kernel     nn.LayerNorm of the GLM-ASR encoder  (TF:models/glmasr/modeling_glmasr.py:246-247,305)
//   rmsnorm_fwd_kernel   Qwen3RMSNorm / LlamaRMSNorm            (TF:models/qwen3/modeling_qwen3.py:50-64;
//                        tiny_audio/projectors.py:43,50), optionally fused with the projector's erf-GELU
//   rmsnorm_bwd_kernel   its backward (dx, optional dw, optional GELU' prologue, optional residual add)
#include <cstdlib>
#include "common.h"
#include "internal.h"

#define MAXV_LIMIT 20   // float4 per lane -> rows up to 64*4*20 = 5120 columns
// MAXV (float4 per lane held in registers) is a template parameter: 4 (H<=1024), 8 (<=2048), 20 (<=5120)

// IN_BF16: the row is read as bf16 (the encoder's bf16 residual stream, as the reference's bf16 model keeps it)
template <int MAXV, bool OUT_BF16, bool OUT_F32, bool IN_BF16 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, bf16_t* __restrict__ yb,
                                                        float* __restrict__ yf, const float* __restrict__ rowscale,
                                                        int M, int H, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const int nv = H >> 2;
  const float4* xr = (const float4*)(x + (long)row * H);
  const uint2* xb = (const uint2*)((const bf16_t*)x + (long)row * H);
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      if (IN_BF16) {
        const uint2 u = xb[c];
        v[i] = make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16)));
      } else {
        v[i] = xr[c];
      }
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, bq = v[i].y - mean, cq = v[i].z - mean, d = v[i].w - mean;
      q += a * a + bq * bq + cq * cq + d * d;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
  const float rs = rowscale ? rowscale[row] : 1.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 ww = ((const float4*)w)[c], bb = ((const float4*)b)[c];
      float4 o;
      o.x = ((v[i].x - mean) * rstd * ww.x + bb.x) * rs;
      o.y = ((v[i].y - mean) * rstd * ww.y + bb.y) * rs;
      o.z = ((v[i].z - mean) * rstd * ww.z + bb.z) * rs;
      o.w = ((v[i].w - mean) * rstd * ww.w + bb.w) * rs;
      if (OUT_F32) ((float4*)(yf + (long)row * H))[c] = o;
      if (OUT_BF16) {
        uint2 p; p.x = pack2bf(o.x, o.y); p.y = pack2bf(o.z, o.w);
        ((uint2*)(yb + (long)row * H))[c] = p;
      }
    }
  }
}


// bf16 -> bf16 LayerNorm with 16-byte accesses: HALF a wave per row (32 lanes x NCH chunks of 8 columns, H = 256 * NCH),
// two rows per wave, eight per workgroup.  The encoder's two LayerNorms per layer read and write the bf16 residual stream
// (82 MB per call at B = 32): halving the number of memory instructions per byte is what this variant is for.
// Round 3: a half wave walks ROWS rows and keeps its gamma / beta columns in registers.  With one row per half wave every
// lane re-read 8 x NCH floats of gamma and of beta per row from L1 -- 4x the bytes of the row itself (20 KB of L1 reads per
// 2.5-KB row at H = 1280: ~8 of the kernel's 16 us at 64 B/clk/CU); the next row's chunks are requested before the current
// row's arithmetic.
template <int NCH, int ROWS>
__global__ __launch_bounds__(256) void layernorm_bf16x8_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ b, bf16_t* __restrict__ y,
                                                               const float* __restrict__ rowscale, int M, float eps) {
  constexpr int H = NCH * 256;
  const int l = threadIdx.x & 31;
  const int row0 = (blockIdx.x * 8 + (threadIdx.x >> 5)) * ROWS;
  if (row0 >= M) return;
  float ww[NCH][8], bb[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (l + i * 32) * 8;
    const float4 w0 = *(const float4*)(w + c), w1 = *(const float4*)(w + c + 4);
    const float4 b0 = *(const float4*)(b + c), b1 = *(const float4*)(b + c + 4);
    ww[i][0] = w0.x; ww[i][1] = w0.y; ww[i][2] = w0.z; ww[i][3] = w0.w; ww[i][4] = w1.x; ww[i][5] = w1.y; ww[i][6] = w1.z; ww[i][7] = w1.w;
    bb[i][0] = b0.x; bb[i][1] = b0.y; bb[i][2] = b0.z; bb[i][3] = b0.w; bb[i][4] = b1.x; bb[i][5] = b1.y; bb[i][6] = b1.z; bb[i][7] = b1.w;
  }
  uint4 nx[NCH];
  {
    const uint4* xr = (const uint4*)(x + (long)row0 * H);
#pragma unroll
    for (int i = 0; i < NCH; ++i) nx[i] = xr[l + i * 32];
  }
#pragma unroll 1
  for (int rr = 0; rr < ROWS; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;                               // (uniform per half wave: the shuffles below stay inside it)
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const uint32_t q[4] = {nx[i].x, nx[i].y, nx[i].z, nx[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[i][2 * j] = bf2f((bf16_t)(q[j] & 0xffff)); v[i][2 * j + 1] = bf2f((bf16_t)(q[j] >> 16)); s += v[i][2 * j] + v[i][2 * j + 1]; }
    }
    if (rr + 1 < ROWS && row + 1 < M) {
      const uint4* xr = (const uint4*)(x + (long)(row + 1) * H);
#pragma unroll
      for (int i = 0; i < NCH; ++i) nx[i] = xr[l + i * 32];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);          // lanes 0-31 / 32-63 reduce separately
    const float mean = s / (float)H;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q2 += d * d; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q2 += __shfl_xor(q2, o, 64);
    const float rstd = rsqrtf(q2 / (float)H + eps);
    const float rs = rowscale ? rowscale[row] : 1.0f;
    uint4* yr = (uint4*)(y + (long)row * H);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = ((v[i][j] - mean) * rstd * ww[i][j] + bb[i][j]) * rs;
      yr[l + i * 32] = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
    }
  }
}


// f32 -> bf16 LayerNorm, the fp32-stream mode of the encoder (round 6): the same half-wave-per-row walk as layernorm_bf16x8_kernel
// -- gamma / beta columns in registers, the next row's chunks requested before the current row's arithmetic, two 16-byte loads and one
// 16-byte store per chunk and lane.  The generic wave-per-row kernel it replaces there re-read gamma and beta (10 KB) for every
// 5-KB row: 24.4 us per launch at B = 32 (123 MB: 5.0 TB/s).
template <int NCH, int ROWS>
__global__ __launch_bounds__(256) void layernorm_f32x8_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, bf16_t* __restrict__ y,
                                                              const float* __restrict__ rowscale, int M, float eps) {
  constexpr int H = NCH * 256;
  const int l = threadIdx.x & 31;
  const int row0 = (blockIdx.x * 8 + (threadIdx.x >> 5)) * ROWS;
  if (row0 >= M) return;
  float ww[NCH][8], bb[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (l + i * 32) * 8;
    const float4 w0 = *(const float4*)(w + c), w1 = *(const float4*)(w + c + 4);
    const float4 b0 = *(const float4*)(b + c), b1 = *(const float4*)(b + c + 4);
    ww[i][0] = w0.x; ww[i][1] = w0.y; ww[i][2] = w0.z; ww[i][3] = w0.w; ww[i][4] = w1.x; ww[i][5] = w1.y; ww[i][6] = w1.z; ww[i][7] = w1.w;
    bb[i][0] = b0.x; bb[i][1] = b0.y; bb[i][2] = b0.z; bb[i][3] = b0.w; bb[i][4] = b1.x; bb[i][5] = b1.y; bb[i][6] = b1.z; bb[i][7] = b1.w;
  }
  float4 nx[NCH][2];
  {
    const float4* xr = (const float4*)(x + (long)row0 * H);
#pragma unroll
    for (int i = 0; i < NCH; ++i) { nx[i][0] = xr[(l + i * 32) * 2]; nx[i][1] = xr[(l + i * 32) * 2 + 1]; }
  }
#pragma unroll 1
  for (int rr = 0; rr < ROWS; ++rr) {
    const int row = row0 + rr;
    if (row >= M) break;                               // (uniform per half wave: the shuffles below stay inside it)
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      v[i][0] = nx[i][0].x; v[i][1] = nx[i][0].y; v[i][2] = nx[i][0].z; v[i][3] = nx[i][0].w;
      v[i][4] = nx[i][1].x; v[i][5] = nx[i][1].y; v[i][6] = nx[i][1].z; v[i][7] = nx[i][1].w;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    }
    if (rr + 1 < ROWS && row + 1 < M) {
      const float4* xr = (const float4*)(x + (long)(row + 1) * H);
#pragma unroll
      for (int i = 0; i < NCH; ++i) { nx[i][0] = xr[(l + i * 32) * 2]; nx[i][1] = xr[(l + i * 32) * 2 + 1]; }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);          // lanes 0-31 / 32-63 reduce separately
    const float mean = s / (float)H;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q2 += d * d; }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q2 += __shfl_xor(q2, o, 64);
    const float rstd = rsqrtf(q2 / (float)H + eps);
    const float rs = rowscale ? rowscale[row] : 1.0f;
    uint4* yr = (uint4*)(y + (long)row * H);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = ((v[i][j] - mean) * rstd * ww[i][j] + bb[i][j]) * rs;
      yr[l + i * 32] = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
    }
  }
}


// bf16 -> bf16 RMSNorm (the LM's residual stream) with 16-byte accesses, half a wave per row: same layout idea as
// layernorm_bf16x8_kernel.  H = 256 * NCH.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_bf16x8_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                                 bf16_t* __restrict__ y, float* __restrict__ rstd_out, int M,
                                                                 float eps) {
  constexpr int H = NCH * 256;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= M) return;
  const int l = threadIdx.x & 31;
  const uint4* xr = (const uint4*)(x + (long)row * H);
  float v[NCH][8];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const uint4 u = xr[l + i * 32];
    const uint32_t t[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[i][2 * j] = bf2f((bf16_t)(t[j] & 0xffff)); v[i][2 * j + 1] = bf2f((bf16_t)(t[j] >> 16));
      q += v[i][2 * j] * v[i][2 * j] + v[i][2 * j + 1] * v[i][2 * j + 1];
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)H + eps);
  if (rstd_out && l == 0) rstd_out[row] = rstd;
  uint4* yr = (uint4*)(y + (long)row * H);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (l + i * 32) * 8;
    const float4 w0 = *(const float4*)(w + c), w1 = *(const float4*)(w + c + 4);
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v[i][j] * rstd * ww[j];
    yr[l + i * 32] = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
  }
}

// f32 -> bf16 RMSNorm (the LM's residual stream in the fp32-stream mode, round 6): the half-wave-per-row layout of
// rmsnorm_fwd_bf16x8_kernel with two 16-byte loads per chunk and lane.  H = 256 * NCH.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_f32x8_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                bf16_t* __restrict__ y, float* __restrict__ rstd_out, int M,
                                                                float eps) {
  constexpr int H = NCH * 256;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= M) return;
  const int l = threadIdx.x & 31;
  const float4* xr = (const float4*)(x + (long)row * H);
  float v[NCH][8];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const float4 a = xr[(l + i * 32) * 2], b = xr[(l + i * 32) * 2 + 1];
    v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w; v[i][4] = b.x; v[i][5] = b.y; v[i][6] = b.z; v[i][7] = b.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) q += v[i][j] * v[i][j];
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)H + eps);
  if (rstd_out && l == 0) rstd_out[row] = rstd;
  uint4* yr = (uint4*)(y + (long)row * H);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = (l + i * 32) * 8;
    const float4 w0 = *(const float4*)(w + c), w1 = *(const float4*)(w + c + 4);
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v[i][j] * rstd * ww[j];
    yr[l + i * 32] = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
  }
}

// y = w * (x * rstd)   [ACT==1: y = gelu(y)];  x f32 [M,H]
template <int MAXV, int ACT, bool IN_BF16 = false>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          bf16_t* __restrict__ yb, float* __restrict__ yf,
                                                          float* __restrict__ rstd_out, int M, int H, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const int nv = H >> 2;
  const float4* xr = (const float4*)(x + (long)row * H);
  float4 v[MAXV];
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      if (IN_BF16) { const uint2 u = ((const uint2*)((const bf16_t*)x + (long)row * H))[c]; v[i] = make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16))); }
      else v[i] = xr[c];
      q += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
  if (rstd_out && lane == 0) rstd_out[row] = rstd;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 ww = ((const float4*)w)[c];
      float4 o;
      o.x = v[i].x * rstd * ww.x; o.y = v[i].y * rstd * ww.y;
      o.z = v[i].z * rstd * ww.z; o.w = v[i].w * rstd * ww.w;
      if (ACT == 1) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
      if (yf) ((float4*)(yf + (long)row * H))[c] = o;
      if (yb) {
        uint2 p; p.x = pack2bf(o.x, o.y); p.y = pack2bf(o.z, o.w);
        ((uint2*)(yb + (long)row * H))[c] = p;
      }
    }
  }
}

// Backward of y = act(w * x * rstd):
//   dn = dy * act'(n)           (ACT==1, n = w*x*rstd recomputed)
//   dx = rstd * (dn*w - xh * mean(dn*w*xh)),  xh = x*rstd        (+ dres if given)
//   dw += sum_rows dn * xh      (if dw != null; LDS partials + one atomicAdd per column per block)
// DRES_BF16 (round 4): the residual gradient `dres` is bf16 and may BE the output image dxb (in place: a lane reads its four
// elements before it writes them) -- the LM's d(x) stream kept in bf16, the dtype the reference's bf16 model back-propagates in
template <int MAXV, int ACT, bool X_BF16 = false, bool DY_BF16 = false, bool DRES_BF16 = false>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ rstd_in,
                                                          const float* __restrict__ w, const float* dres,
                                                          float* dxf, bf16_t* dxb,
                                                          float* __restrict__ dw, int M, int H) {
  extern __shared__ float dw_part[];   // [H] when dw != null
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nv = H >> 2;
  if (dw) {
    for (int i = threadIdx.x; i < H; i += 256) dw_part[i] = 0.f;
    __syncthreads();
  }
  // each block handles rows_per_block consecutive groups of 4 rows (grid-stride) so dw atomics stay few
  for (int row0 = blockIdx.x * 4; row0 < M; row0 += gridDim.x * 4) {
    const int row = row0 + wv;
    if (row < M) {
      const float r = rstd_in[row];
      const float4* xr = (const float4*)(x + (long)row * H);
      const float4* dr = (const float4*)(dy + (long)row * H);
      float4 xh[MAXV], dn[MAXV];
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
          float4 xv;
          if (X_BF16) { const uint2 u = ((const uint2*)((const bf16_t*)x + (long)row * H))[c]; xv = make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16))); }
          else xv = xr[c];
          float4 dv;
          if (DY_BF16) { const uint2 u = ((const uint2*)((const bf16_t*)dy + (long)row * H))[c]; dv = make_float4(bf2f((bf16_t)(u.x & 0xffff)), bf2f((bf16_t)(u.x >> 16)), bf2f((bf16_t)(u.y & 0xffff)), bf2f((bf16_t)(u.y >> 16))); }
          else dv = dr[c];
          const float4 ww = ((const float4*)w)[c];
          xh[i] = make_float4(xv.x * r, xv.y * r, xv.z * r, xv.w * r);
          float4 d = dv;
          if (ACT == 1) {
            d.x *= gelu_erf_grad(xh[i].x * ww.x); d.y *= gelu_erf_grad(xh[i].y * ww.y);
            d.z *= gelu_erf_grad(xh[i].z * ww.z); d.w *= gelu_erf_grad(xh[i].w * ww.w);
          }
          if (dw) {
            atomicAdd(&dw_part[c * 4 + 0], d.x * xh[i].x); atomicAdd(&dw_part[c * 4 + 1], d.y * xh[i].y);
            atomicAdd(&dw_part[c * 4 + 2], d.z * xh[i].z); atomicAdd(&dw_part[c * 4 + 3], d.w * xh[i].w);
          }
          dn[i] = make_float4(d.x * ww.x, d.y * ww.y, d.z * ww.z, d.w * ww.w);
          dot += dn[i].x * xh[i].x + dn[i].y * xh[i].y + dn[i].z * xh[i].z + dn[i].w * xh[i].w;
        }
      }
      const float mdot = wave_sum(dot) / (float)H;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
          float4 o;
          o.x = r * (dn[i].x - xh[i].x * mdot); o.y = r * (dn[i].y - xh[i].y * mdot);
          o.z = r * (dn[i].z - xh[i].z * mdot); o.w = r * (dn[i].w - xh[i].w * mdot);
          if (dres) {
            if (DRES_BF16) {
              const uint2 u = ((const uint2*)((const bf16_t*)dres + (long)row * H))[c];
              o.x += bf2f((bf16_t)(u.x & 0xffff)); o.y += bf2f((bf16_t)(u.x >> 16)); o.z += bf2f((bf16_t)(u.y & 0xffff)); o.w += bf2f((bf16_t)(u.y >> 16));
            } else {
              const float4 e = ((const float4*)(dres + (long)row * H))[c];
              o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
            }
          }
          if (dxf) ((float4*)(dxf + (long)row * H))[c] = o;
          if (dxb) {
            uint2 p; p.x = pack2bf(o.x, o.y); p.y = pack2bf(o.z, o.w);
            ((uint2*)(dxb + (long)row * H))[c] = p;
          }
        }
      }
    }
  }
  if (dw) {
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += 256) atomicAdd(&dw[i], dw_part[i]);
  }
}

// RMSNorm backward, half a wave per row with 16-byte accesses (round 6): the all-bf16 form the LM backward runs 56 times per step in the
// bf16-stream mode -- x, the incoming gradient, dres and the d(x) image bf16 (dres may BE dxb: the stream updated in place), optional
// f32 copy; no GELU, no weight gradient.  H = 256 * NCH.  11.3 -> 10.3 us per launch at M = 6144, H = 1024 (profiles/r06_g_*).  (The
// fp32-stream counterpart -- five streams, 100 MB per launch -- measured SLOWER in this layout, 19.5 against 18.6 us, and stays on the
// wave-per-row kernel.)
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_bf16x8_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                 const float* __restrict__ rstd_in, const float* __restrict__ w,
                                                                 const bf16_t* dres, float* dxf, bf16_t* dxb, int M) {
  constexpr int H = NCH * 256;
  const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= M) return;
  const int l = threadIdx.x & 31;
  const float r = rstd_in[row];
  float xh[NCH][8], dn[NCH][8];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c8 = l + i * 32;                                   // 8-column chunk index inside the row
    const uint4 u = ((const uint4*)(x + (long)row * H))[c8];
    const uint4 du = ((const uint4*)(dy + (long)row * H))[c8];
    const uint32_t t[4] = {u.x, u.y, u.z, u.w}, dt[4] = {du.x, du.y, du.z, du.w};
    const float4 w0 = *(const float4*)(w + c8 * 8), w1 = *(const float4*)(w + c8 * 8 + 4);
    const float ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xv = bf2f((bf16_t)((j & 1) ? (t[j >> 1] >> 16) : (t[j >> 1] & 0xffff)));
      const float d = bf2f((bf16_t)((j & 1) ? (dt[j >> 1] >> 16) : (dt[j >> 1] & 0xffff)));
      xh[i][j] = xv * r;
      dn[i][j] = d * ww[j];
      dot += dn[i][j] * xh[i][j];
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);          // lanes 0-31 / 32-63 reduce separately
  const float mdot = dot / (float)H;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c8 = l + i * 32;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = r * (dn[i][j] - xh[i][j] * mdot);
    if (dres) {
      const uint4 e = ((const uint4*)(dres + (long)row * H))[c8];
      const uint32_t et[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[2 * j] += bf2f((bf16_t)(et[j] & 0xffff)); o[2 * j + 1] += bf2f((bf16_t)(et[j] >> 16)); }
    }
    if (dxf) {
      float4* fr = (float4*)(dxf + (long)row * H);
      fr[c8 * 2] = make_float4(o[0], o[1], o[2], o[3]); fr[c8 * 2 + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
    if (dxb) ((uint4*)(dxb + (long)row * H))[c8] = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
  }
}

// ----------------------------------------------------------------------------- C-ABI
#define DISPATCH_MAXV(H, CALL)            \
  do {                                    \
    if ((H) <= 1024) { CALL(4); }         \
    else if ((H) <= 2048) { CALL(8); }    \
    else { CALL(20); }                    \
  } while (0)

extern "C" int ta_layernorm_bf16(const void* x_bf16, const float* w, const float* b, void* y_bf16, float* y_f32,
                                 const float* rowscale, int M, int H, float eps, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if ((H & 3) || H > 64 * 4 * MAXV_LIMIT || (!y_bf16 && !y_f32)) return TA_ERR_ARG;
  dim3 grid(ta_cdiv(M, 4)), blk(256);
  const float* x = (const float*)x_bf16;
  // bf16 -> bf16 only, H a multiple of 256 up to 2048: the 16-byte half-wave-per-row variant (other shapes: the generic kernel)
  if (y_bf16 && !y_f32 && (H % 256) == 0 && H <= 2048) {
    // rows per half wave: 4 once that still leaves >= 1 workgroup per CU (M = 16000 at B = 32: 500 workgroups; measured 41.62 /
    // 41.63 / 41.88 ms per step at 4 / 2 / 1, profiles/r03_k_ab_ln_rows.txt), 2 from 4096 rows, else 1
    const int rows = M >= 8 * 4 * 256 ? 4 : (M >= 8 * 2 * 256 ? 2 : 1);
    dim3 g8(ta_cdiv(M, 8 * rows));
    switch (H / 256) {
#define LNW(N) case N: if (rows == 4) TA_LAUNCH((layernorm_bf16x8_kernel<N, 4>), g8, blk, 0, st, (const bf16_t*)x_bf16, w, b, (bf16_t*)y_bf16, rowscale, M, eps); \
               else if (rows == 2) TA_LAUNCH((layernorm_bf16x8_kernel<N, 2>), g8, blk, 0, st, (const bf16_t*)x_bf16, w, b, (bf16_t*)y_bf16, rowscale, M, eps); \
               else TA_LAUNCH((layernorm_bf16x8_kernel<N, 1>), g8, blk, 0, st, (const bf16_t*)x_bf16, w, b, (bf16_t*)y_bf16, rowscale, M, eps); break;
      LNW(1) LNW(2) LNW(3) LNW(4) LNW(5) LNW(6) LNW(7) LNW(8)
#undef LNW
    }
    TA_CHECK_LAUNCH();
    return TA_OK;
  }
#define LNB_CALL(V)                                                                                              \
  if (y_bf16 && y_f32)                                                                                           \
    TA_LAUNCH((layernorm_kernel<V, true, true, true>), grid, blk, 0, st, x, w, b, (bf16_t*)y_bf16, y_f32, rowscale, M, H, eps);  \
  else if (y_bf16)                                                                                               \
    TA_LAUNCH((layernorm_kernel<V, true, false, true>), grid, blk, 0, st, x, w, b, (bf16_t*)y_bf16, y_f32, rowscale, M, H, eps); \
  else                                                                                                           \
    TA_LAUNCH((layernorm_kernel<V, false, true, true>), grid, blk, 0, st, x, w, b, (bf16_t*)y_bf16, y_f32, rowscale, M, H, eps);
  DISPATCH_MAXV(H, LNB_CALL);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_layernorm_f32(const float* x, const float* w, const float* b, void* y_bf16, float* y_f32,
                                const float* rowscale, int M, int H, float eps, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if ((H & 3) || H > 64 * 4 * MAXV_LIMIT || (!y_bf16 && !y_f32)) return TA_ERR_ARG;
  dim3 grid(ta_cdiv(M, 4)), blk(256);
  // f32 -> bf16 only, H a multiple of 256 up to 2048: the half-wave-per-row variant (as ta_layernorm_bf16; other shapes: the generic kernel)
  if (y_bf16 && !y_f32 && (H % 256) == 0 && H <= 2048) {
    const int rows = M >= 8 * 4 * 256 ? 4 : (M >= 8 * 2 * 256 ? 2 : 1);
    dim3 g8(ta_cdiv(M, 8 * rows));
    switch (H / 256) {
#define LNF(N) case N: if (rows == 4) TA_LAUNCH((layernorm_f32x8_kernel<N, 4>), g8, blk, 0, st, x, w, b, (bf16_t*)y_bf16, rowscale, M, eps); \
               else if (rows == 2) TA_LAUNCH((layernorm_f32x8_kernel<N, 2>), g8, blk, 0, st, x, w, b, (bf16_t*)y_bf16, rowscale, M, eps); \
               else TA_LAUNCH((layernorm_f32x8_kernel<N, 1>), g8, blk, 0, st, x, w, b, (bf16_t*)y_bf16, rowscale, M, eps); break;
      LNF(1) LNF(2) LNF(3) LNF(4) LNF(5) LNF(6) LNF(7) LNF(8)
#undef LNF
    }
    TA_CHECK_LAUNCH();
    return TA_OK;
  }
#define LN_CALL(V)                                                                                               \
  if (y_bf16 && y_f32)                                                                                           \
    TA_LAUNCH((layernorm_kernel<V, true, true>), grid, blk, 0, st, x, w, b, (bf16_t*)y_bf16, y_f32, rowscale, M, H, eps);  \
  else if (y_bf16)                                                                                               \
    TA_LAUNCH((layernorm_kernel<V, true, false>), grid, blk, 0, st, x, w, b, (bf16_t*)y_bf16, y_f32, rowscale, M, H, eps); \
  else                                                                                                           \
    TA_LAUNCH((layernorm_kernel<V, false, true>), grid, blk, 0, st, x, w, b, (bf16_t*)y_bf16, y_f32, rowscale, M, H, eps);
  DISPATCH_MAXV(H, LN_CALL);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_rmsnorm_fwd(const float* x, const float* w, void* y_bf16, float* y_f32, float* rstd,
                              int M, int H, float eps, int act_gelu, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if ((H & 3) || H > 64 * 4 * MAXV_LIMIT) return TA_ERR_ARG;
  dim3 grid(ta_cdiv(M, 4)), blk(256);
  if (!act_gelu && y_bf16 && !y_f32 && (H % 256) == 0 && H <= 2048) {      // f32 -> bf16 only: the half-wave-per-row variant
    dim3 g8(ta_cdiv(M, 8));
    switch (H / 256) {
#define RFF(N) case N: TA_LAUNCH((rmsnorm_fwd_f32x8_kernel<N>), g8, blk, 0, st, x, w, (bf16_t*)y_bf16, rstd, M, eps); break;
      RFF(1) RFF(2) RFF(3) RFF(4) RFF(5) RFF(6) RFF(7) RFF(8)
#undef RFF
    }
    TA_CHECK_LAUNCH();
    return TA_OK;
  }
#define RF_CALL(V)                                                                                                 \
  if (act_gelu)                                                                                                    \
    TA_LAUNCH((rmsnorm_fwd_kernel<V, 1>), grid, blk, 0, st, x, w, (bf16_t*)y_bf16, y_f32, rstd, M, H, eps); \
  else                                                                                                             \
    TA_LAUNCH((rmsnorm_fwd_kernel<V, 0>), grid, blk, 0, st, x, w, (bf16_t*)y_bf16, y_f32, rstd, M, H, eps);
  DISPATCH_MAXV(H, RF_CALL);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

// x read as bf16 (the LM's residual stream in the reference's model dtype)
extern "C" int ta_rmsnorm_fwd_bf16(const void* x_bf16, const float* w, void* y_bf16, float* y_f32, float* rstd, int M, int H,
                                   float eps, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if ((H & 3) || H > 64 * 4 * MAXV_LIMIT) return TA_ERR_ARG;
  dim3 grid(ta_cdiv(M, 4)), blk(256);
  const float* x = (const float*)x_bf16;
  if (y_bf16 && !y_f32 && (H % 256) == 0 && H <= 2048) {
    dim3 g8(ta_cdiv(M, 8));
    switch (H / 256) {
#define RFW(N) case N: TA_LAUNCH((rmsnorm_fwd_bf16x8_kernel<N>), g8, blk, 0, st, (const bf16_t*)x_bf16, w, (bf16_t*)y_bf16, rstd, M, eps); break;
      RFW(1) RFW(2) RFW(3) RFW(4) RFW(5) RFW(6) RFW(7) RFW(8)
#undef RFW
    }
    TA_CHECK_LAUNCH();
    return TA_OK;
  }
#define RFB_CALL(V) TA_LAUNCH((rmsnorm_fwd_kernel<V, 0, true>), grid, blk, 0, st, x, w, (bf16_t*)y_bf16, y_f32, rstd, M, H, eps);
  DISPATCH_MAXV(H, RFB_CALL);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
extern "C" int ta_rmsnorm_bwd_bf16(const void* dy, int dy_is_bf16, const void* x_bf16, const float* rstd, const float* w,
                                   const float* dres, float* dx_f32, void* dx_bf16, int M, int H, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if ((H & 3) || H > 64 * 4 * MAXV_LIMIT) return TA_ERR_ARG;
  const float* x = (const float*)x_bf16;
  const float* dyf = (const float*)dy;
#define RBB_CALL(V)                                                                                                        \
  if (dy_is_bf16) TA_LAUNCH((rmsnorm_bwd_kernel<V, 0, true, true>), dim3(ta_cdiv(M, 4)), dim3(256), 0, st, dyf, x, rstd, w, dres, \
                            dx_f32, (bf16_t*)dx_bf16, (float*)nullptr, M, H);                                              \
  else TA_LAUNCH((rmsnorm_bwd_kernel<V, 0, true, false>), dim3(ta_cdiv(M, 4)), dim3(256), 0, st, dyf, x, rstd, w, dres, dx_f32,  \
                 (bf16_t*)dx_bf16, (float*)nullptr, M, H);
  DISPATCH_MAXV(H, RBB_CALL);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

// the same with the residual gradient read as bf16 (dres_bf16 may alias dx_bf16: the bf16 d(x) stream updated in place)
extern "C" int ta_rmsnorm_bwd_bf16s(const void* dy, int dy_is_bf16, const void* x_bf16, const float* rstd, const float* w,
                                    const void* dres_bf16, float* dx_f32, void* dx_bf16, int M, int H, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if ((H & 3) || H > 64 * 4 * MAXV_LIMIT) return TA_ERR_ARG;
  if (dy_is_bf16 && (H % 256) == 0 && H <= 2048) {
    dim3 g8(ta_cdiv(M, 8)), blk8(256);
    switch (H / 256) {
#define RBX(N) case N: TA_LAUNCH((rmsnorm_bwd_bf16x8_kernel<N>), g8, blk8, 0, st, (const bf16_t*)dy, (const bf16_t*)x_bf16, rstd, w, (const bf16_t*)dres_bf16, dx_f32, (bf16_t*)dx_bf16, M); break;
      RBX(1) RBX(2) RBX(3) RBX(4) RBX(5) RBX(6) RBX(7) RBX(8)
#undef RBX
    }
    TA_CHECK_LAUNCH();
    return TA_OK;
  }
  const float* x = (const float*)x_bf16;
  const float* dyf = (const float*)dy;
  const float* dres = (const float*)dres_bf16;
#define RBS_CALL(V)                                                                                                                       \
  if (dy_is_bf16) TA_LAUNCH((rmsnorm_bwd_kernel<V, 0, true, true, true>), dim3(ta_cdiv(M, 4)), dim3(256), 0, st, dyf, x, rstd, w, dres,      \
                            dx_f32, (bf16_t*)dx_bf16, (float*)nullptr, M, H);                                                             \
  else TA_LAUNCH((rmsnorm_bwd_kernel<V, 0, true, false, true>), dim3(ta_cdiv(M, 4)), dim3(256), 0, st, dyf, x, rstd, w, dres, dx_f32,       \
                 (bf16_t*)dx_bf16, (float*)nullptr, M, H);
  DISPATCH_MAXV(H, RBS_CALL);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

// fp32 residual stream, bf16 incoming gradient (ta_lm_backward in the fp32-stream mode -- under the recipe's bf16 autocast
// the gradient of a Linear's bf16 input IS a bf16 tensor, cast up afterwards, so nothing is lost by keeping its 2 bytes)
extern "C" int ta_rmsnorm_bwd_dyb(const void* dy_bf16, const float* x, const float* rstd, const float* w, const float* dres, float* dx_f32,
                         void* dx_bf16, int M, int H, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if ((H & 3) || H > 64 * 4 * MAXV_LIMIT) return TA_ERR_ARG;
  const float* dyf = (const float*)dy_bf16;
#define RBD_CALL(V)                                                                                                              \
  TA_LAUNCH((rmsnorm_bwd_kernel<V, 0, false, true, false>), dim3(ta_cdiv(M, 4)), dim3(256), 0, st, dyf, x, rstd, w, dres, dx_f32, \
            (bf16_t*)dx_bf16, (float*)nullptr, M, H);
  DISPATCH_MAXV(H, RBD_CALL);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_rmsnorm_bwd(const float* dy, const float* x, const float* rstd, const float* w,
                              const float* dres, float* dx_f32, void* dx_bf16, float* dw_accum,
                              int M, int H, int act_gelu, hipStream_t st) {
  if (M <= 0) return TA_OK;
  if ((H & 3) || H > 64 * 4 * MAXV_LIMIT) return TA_ERR_ARG;
  int blocks = ta_cdiv(M, 4);
  if (dw_accum && blocks > 512) blocks = 512;
  const size_t lds = dw_accum ? (size_t)H * 4 : 0;
#define RB_CALL(V)                                                                                             \
  if (act_gelu)                                                                                                \
    TA_LAUNCH((rmsnorm_bwd_kernel<V, 1>), dim3(blocks), dim3(256), lds, st, dy, x, rstd, w, dres,     \
                       dx_f32, (bf16_t*)dx_bf16, dw_accum, M, H);                                              \
  else                                                                                                         \
    TA_LAUNCH((rmsnorm_bwd_kernel<V, 0>), dim3(blocks), dim3(256), lds, st, dy, x, rstd, w, dres,     \
                       dx_f32, (bf16_t*)dx_bf16, dw_accum, M, H);
  DISPATCH_MAXV(H, RB_CALL);
  TA_CHECK_LAUNCH();
  return TA_OK;
}


// ---------------------------------------------------------------------------- RMSNorm weight gradient (trainable LM)
// dw[h] += sum_m dy[m,h] * x[m,h] * rstd[m].  One workgroup = 32 rows x 256 columns: 4 row slots x 64 lanes of 4 columns,
// 8 rows per thread with the loads of a row pair in flight together; the 4 slots fold through LDS, one atomic per column.
template <bool DY_BF16, bool X_BF16>
__global__ __launch_bounds__(256) void rmsnorm_dw_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                         const float* __restrict__ rstd, float* __restrict__ dw, int M, int H) {
  __shared__ float part[4][256];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + tx * 4, r0 = blockIdx.x * 32;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  auto ld4 = [&](const void* p, bool bf, long off, float* o) {
    if (bf) { const uint2 u = *(const uint2*)((const bf16_t*)p + off); o[0] = bf2f((bf16_t)(u.x & 0xffff)); o[1] = bf2f((bf16_t)(u.x >> 16)); o[2] = bf2f((bf16_t)(u.y & 0xffff)); o[3] = bf2f((bf16_t)(u.y >> 16)); }
    else { const float4 u = *(const float4*)((const float*)p + off); o[0] = u.x; o[1] = u.y; o[2] = u.z; o[3] = u.w; }
  };
  if (c < H) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const int ra = r0 + ty + 4 * i, rb = ra + 4;
      float g0[4], v0[4], g1[4], v1[4];
      const bool ia = ra < M, ib = rb < M;
      if (ia) { ld4(dy, DY_BF16, (long)ra * H + c, g0); ld4(x, X_BF16, (long)ra * H + c, v0); }
      if (ib) { ld4(dy, DY_BF16, (long)rb * H + c, g1); ld4(x, X_BF16, (long)rb * H + c, v1); }
      if (ia) { const float rs = rstd[ra]; for (int k = 0; k < 4; ++k) a[k] += g0[k] * v0[k] * rs; }
      if (ib) { const float rs = rstd[rb]; for (int k = 0; k < 4; ++k) a[k] += g1[k] * v1[k] * rs; }
    }
  }
  for (int k = 0; k < 4; ++k) part[ty][tx * 4 + k] = a[k];
  __syncthreads();
  const int cc = blockIdx.y * 256 + threadIdx.x;
  if (cc < H) unsafeAtomicAdd(dw + cc, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

extern "C" int ta_rmsnorm_dw(const void* dy, int dy_is_bf16, const void* x, int x_is_bf16, const float* rstd, float* dw_accum,
                             int M, int H, hipStream_t st) {
  if (M <= 0 || H <= 0) return TA_OK;
  if (H & 3) return TA_ERR_ARG;
  dim3 grid(ta_cdiv(M, 32), ta_cdiv(H, 256)), blk(256);
  if (dy_is_bf16 && x_is_bf16) TA_LAUNCH((rmsnorm_dw_kernel<true, true>), grid, blk, 0, st, dy, x, rstd, dw_accum, M, H);
  else if (dy_is_bf16) TA_LAUNCH((rmsnorm_dw_kernel<true, false>), grid, blk, 0, st, dy, x, rstd, dw_accum, M, H);
  else if (x_is_bf16) TA_LAUNCH((rmsnorm_dw_kernel<false, true>), grid, blk, 0, st, dy, x, rstd, dw_accum, M, H);
  else TA_LAUNCH((rmsnorm_dw_kernel<false, false>), grid, blk, 0, st, dy, x, rstd, dw_accum, M, H);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
