// ta355 greedy decoding, round 4: a decode step of one layer as FIVE launches instead of nine (generate.hip holds the step
// itself and the unfused kernels, which remain the path for LoRA adapters and for shapes outside this file's envelope).
//
// Reference: one Qwen3 decoder layer on ONE new position per clip against the KV cache (TF:models/qwen3/modeling_qwen3.py:211-280
// with past_key_values; driven by tiny_audio/asr_modeling.py:562-646 through HF greedy search).
//
// Why: at batch <= 32 every kernel of the step is a dependent link of a chain, and what a link costs is not its bytes but its fixed
// latency (dispatch, one DRAM round trip, the cross-wave reduction, the drain: ~5.5 us with 4-12 MB of weights behind it): the round-3
// step ran 9 kernels per layer at 4.9-12 us each = 69 us per layer for 57 MB of weights + cache rows
// (profiles/r04_o_decode_kernel_stats_before.md), and its linears covered 64-96 of the 256 CUs.  Here
//   dec_linear_kernel<NORM, bf16>      RMSNorm(x) folded into q|k|v: every workgroup recomputes the 32 row norms (32 columns each)
//   dec_attn_kernel                    per-head q/k RMSNorm + RoPE + cache append + attention, one workgroup per (clip, kv head):
//                                      K and V rows are read ONCE for the q heads of the group, both streams requested up front
//   dec_linear_kernel<PLAIN, f32+res>  o_proj + residual, 8 output columns per workgroup
//   dec_linear_kernel<NORM, SwiGLU>    RMSNorm(x1) + gate|up + SiLU(gate) * up: a workgroup owns 16 gate rows and the same 16 up rows
//   dec_linear_kernel<PLAIN, f32+res>  down_proj + residual
// Every linear workgroup is 8 waves that split K (wave w takes k-steps w, w + 8, ...), MFMA fragments come straight from global
// memory (both operands are K-major), all loads of a workgroup are in flight before its first MFMA, weights are read with the
// non-temporal policy (each byte is used once per token), and the partial sums meet in LDS in a fixed order: results do not
// depend on the launch geometry or on timing.  The activations of the step live in a BLOCKED layout (blk_off below), and every
// kernel carries extra workgroups that touch what the NEXT kernel will stream (DecPf below): DESIGN.md section 3, "Greedy decoding",
// has the measurements behind both (scripts/probe/dec_probe.hip).
#include <cstdlib>
#include "host_util.h"
#include "internal.h"

namespace {
constexpr int HD = 128;
typedef __attribute__((ext_vector_type(8))) short dbf16x8;
typedef __attribute__((ext_vector_type(4))) float df32x4;
typedef __attribute__((ext_vector_type(4))) unsigned du32x4;

__device__ __forceinline__ dbf16x8 ld_nt(const bf16_t* p) {
#ifdef TA355_DEC_PLAIN_LOADS
  return *(const dbf16x8*)p;
#else
  const du32x4 v = __builtin_nontemporal_load((const du32x4*)p);
  return __builtin_bit_cast(dbf16x8, v);
#endif
}

enum { EPI_BF16 = 0, EPI_F32_RES = 1, EPI_SWIGLU = 2 };

// What the NEXT kernel of the step will stream, touched by extra workgroups of this one (blockIdx past the compute grid): the
// kernels of a step are links of a dependent chain, each one launch + one DRAM round trip + a reduction long (5-6 us with 4-12 MB
// of weights, against 1-3 us of HBM time), and HBM idles through every boundary.  Weights and cache rows do not depend on the
// activations, so their fetch can start one kernel early: they then wait in the memory-side cache (256 MB), or in the XCD's L2.
//   contiguous:  [p0, p0 + bytes)
//   cache rows:  `rows` rows of p0 and of p1, `row_stride` bytes apart, the first *slot_p * 256 bytes of each (K and V of one layer)
struct DecPf { const char* p0; const char* p1; long bytes; const int* slot_p; long row_stride; int rows; };
// one dword per 128-byte line: the line is what travels from HBM, the register traffic stays negligible
__device__ __forceinline__ void dec_prefetch(const DecPf& pf, int wg, int nwg, int tid, int nthr) {
  unsigned acc = 0;
  if (pf.p1 == nullptr) {
    const long step = (long)nwg * nthr * 128;
    long off = ((long)wg * nthr + tid) * 128;
    for (; off + 3 * step < pf.bytes; off += 4 * step) {
      unsigned v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *(const unsigned*)(pf.p0 + off + j * step);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc ^= v[j];
    }
    for (; off < pf.bytes; off += step) acc ^= *(const unsigned*)(pf.p0 + off);
  } else {
    const long rb = (long)(*pf.slot_p) * 256;                      // bytes per row prefix
    for (int r = wg; r < pf.rows; r += nwg) {
      const char* a = pf.p0 + (long)r * pf.row_stride;
      const char* b = pf.p1 + (long)r * pf.row_stride;
      for (long off = (long)tid * 128; off < rb; off += (long)nthr * 128) { acc ^= *(const unsigned*)(a + off); acc ^= *(const unsigned*)(b + off); }
    }
  }
  asm volatile("" ::"v"(acc));
}

// BLOCKED activation layout of the step (BLK): element (row, k) of a [32, K] activation lives at ((k >> 5) * 32 + row) * 32 + (k & 31),
// i.e. k-step-major blocks [K/32][32 rows][32].  Every workgroup of a decode linear reads ALL of X, so X is a broadcast out of
// each XCD's L2; row-major rows are K*2 or K*4 bytes apart -- a multiple of the 4 KB channel period for every K of the model -- and
// the 16 rows of one fragment load then sit on ONE L2 channel (measured, scripts/probe/dec_probe.hip: the X loads were 2.5-6.8 of the
// 8-13 us of a kernel, the 4-12 MB weight stream 1.3-3.4).  Blocked, a wave's fragment load of 16 rows is 1 KB contiguous (bf16; 2 KB
// for the f32 stream) and consecutive k-steps walk through all channels.
__device__ __forceinline__ long blk_off(int row, int k) { return ((long)(k >> 5) * 32 + row) * 32 + (k & 31); }

// out = epilogue( prologue(X)[M, K] W[., K]^T ),  M <= 32.
//   NORM:  X is the f32 residual stream; prologue = bf16(x * rstd(x) * lnw) (Qwen3RMSNorm, as rmsnorm_fwd_kernel rounds it)
//   else:  X is bf16
//   EPI_BF16:    COLS (16 / 32) columns per workgroup, out bf16 [M, N] row-major (the attention kernel reads heads out of it)
//   EPI_F32_RES: COLS (4 / 8 / 16) columns per workgroup (the 16 MFMA columns repeat them), out f32 = acc + res (both in X's layout)
//   EPI_SWIGLU:  COLS (8 / 16) gate rows f0.. and the same up rows F + f0.. per workgroup, out bf16 [M, F] = silu(bf16(gate)) * bf16(up)
// KS = k-steps per wave = K / 256; UN of them are requested at a time (double-buffered when KS > UN).
// DBG (scripts/probe/dec_probe.hip only; the library instantiates 0): 1 = no W loads, 2 = no X loads, 4 = no epilogue arithmetic,
// 8 = return at once
template <bool NORM, int EPI, int UN, int COLS, bool BLK, int DBG = 0>
__global__ __launch_bounds__(512) void dec_linear_kernel(const void* __restrict__ Xv, const float* __restrict__ lnw, float eps,
                                                         const bf16_t* __restrict__ W, void* __restrict__ out,
                                                         const float* __restrict__ res, int M, int N, int K, int n_main, DecPf pf) {
  constexpr int NB = EPI == EPI_SWIGLU ? (COLS == 16 ? 2 : 1) : (COLS + 15) / 16;      // 16-column MFMA blocks per workgroup
  constexpr int CPB = EPI == EPI_SWIGLU ? 16 : COLS / NB;                              // distinct W rows per block
  if constexpr ((DBG & 8) != 0) return;
  if ((int)blockIdx.x >= n_main) { dec_prefetch(pf, blockIdx.x - n_main, gridDim.x - n_main, threadIdx.x, 512); return; }
  __shared__ float red[8][32][NB * 16 + 1];
  __shared__ float ssq[8][32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * COLS;
  const int wmax = EPI == EPI_SWIGLU ? 2 * N - 1 : N - 1;
  const bf16_t* wp[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    int wrow;
    if (EPI == EPI_SWIGLU) wrow = COLS == 16 ? nb * N + n0 + i : (i < 8 ? n0 + i : N + n0 + (i - 8));   // N = F: gate rows [0, F), up rows [F, 2F)
    else wrow = n0 + nb * CPB + (i % CPB);
    wp[nb] = W + (long)min(wrow, wmax) * K + g * 8;
  }
  const int ra = BLK ? i : min(i, M - 1), rb = BLK ? 16 + i : min(16 + i, M - 1);   // (blocked buffers always hold 32 rows; rows >= M are never stored)
  const int KS = K >> 8;
  df32x4 acc[2][NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) { acc[0][nb] = (df32x4){0.f, 0.f, 0.f, 0.f}; acc[1][nb] = (df32x4){0.f, 0.f, 0.f, 0.f}; }
  if constexpr (NORM) {
    // K = hidden: KS <= UN by construction of the launcher (UN = 4 or 8): everything is requested before anything is used
    const float* X = (const float*)Xv;
    float4 fa[UN][2], fb[UN][2], lw[UN][2];
    dbf16x8 wf[UN][NB];
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (u < KS) {
        const int kk = (wave + 8 * u) * 32;
        const float* xa = BLK ? X + blk_off(ra, kk) + g * 8 : X + (long)ra * K + kk + g * 8;
        const float* xb = BLK ? X + blk_off(rb, kk) + g * 8 : X + (long)rb * K + kk + g * 8;
        if constexpr ((DBG & 2) != 0) { fa[u][0] = fa[u][1] = fb[u][0] = fb[u][1] = make_float4(0.5f, 0.25f, 1.f, 2.f); }
        else {
          fa[u][0] = *(const float4*)xa; fa[u][1] = *(const float4*)(xa + 4);
          fb[u][0] = *(const float4*)xb; fb[u][1] = *(const float4*)(xb + 4);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if constexpr ((DBG & 1) != 0) wf[u][nb] = (dbf16x8){1, 2, 3, 4, 5, 6, 7, 8}; else wf[u][nb] = ld_nt(wp[nb] + kk);
        }
        lw[u][0] = *(const float4*)(lnw + kk + g * 8); lw[u][1] = *(const float4*)(lnw + kk + g * 8 + 4);
      }
    float qa = 0.f, qb = 0.f;
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (u < KS) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          qa += fa[u][h].x * fa[u][h].x + fa[u][h].y * fa[u][h].y + fa[u][h].z * fa[u][h].z + fa[u][h].w * fa[u][h].w;
          qb += fb[u][h].x * fb[u][h].x + fb[u][h].y * fb[u][h].y + fb[u][h].z * fb[u][h].z + fb[u][h].w * fb[u][h].w;
        }
      }
    qa += __shfl_xor(qa, 16, 64); qa += __shfl_xor(qa, 32, 64);
    qb += __shfl_xor(qb, 16, 64); qb += __shfl_xor(qb, 32, 64);
    if (g == 0) { ssq[wave][i] = qa; ssq[wave][16 + i] = qb; }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { ta += ssq[w][i]; tb += ssq[w][16 + i]; }
    const float rsa = rsqrtf(ta / (float)K + eps), rsb = rsqrtf(tb / (float)K + eps);
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (u < KS) {
        const float4 w0 = lw[u][0], w1 = lw[u][1];
        du32x4 pa, pb;
        pa.x = pack2bf(fa[u][0].x * rsa * w0.x, fa[u][0].y * rsa * w0.y); pa.y = pack2bf(fa[u][0].z * rsa * w0.z, fa[u][0].w * rsa * w0.w);
        pa.z = pack2bf(fa[u][1].x * rsa * w1.x, fa[u][1].y * rsa * w1.y); pa.w = pack2bf(fa[u][1].z * rsa * w1.z, fa[u][1].w * rsa * w1.w);
        pb.x = pack2bf(fb[u][0].x * rsb * w0.x, fb[u][0].y * rsb * w0.y); pb.y = pack2bf(fb[u][0].z * rsb * w0.z, fb[u][0].w * rsb * w0.w);
        pb.z = pack2bf(fb[u][1].x * rsb * w1.x, fb[u][1].y * rsb * w1.y); pb.w = pack2bf(fb[u][1].z * rsb * w1.z, fb[u][1].w * rsb * w1.w);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dbf16x8, pa), wf[u][nb], acc[0][nb], 0, 0, 0);
          acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dbf16x8, pb), wf[u][nb], acc[1][nb], 0, 0, 0);
        }
      }
  } else {
    const bf16_t* X = (const bf16_t*)Xv;
    dbf16x8 a0[2][UN], a1[2][UN], wf[2][UN][NB];
    auto fetch = [&](int c, int buf) {
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int ks = c * UN + u;
        if (ks < KS) {
          const int kk = (wave + 8 * ks) * 32;
          if constexpr ((DBG & 2) != 0) { a0[buf][u] = a1[buf][u] = (dbf16x8){8, 7, 6, 5, 4, 3, 2, 1}; }
          else {
            a0[buf][u] = *(const dbf16x8*)(BLK ? X + blk_off(ra, kk) + g * 8 : X + (long)ra * K + kk + g * 8);
            a1[buf][u] = *(const dbf16x8*)(BLK ? X + blk_off(rb, kk) + g * 8 : X + (long)rb * K + kk + g * 8);
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            if constexpr ((DBG & 1) != 0) wf[buf][u][nb] = (dbf16x8){1, 2, 3, 4, 5, 6, 7, 8}; else wf[buf][u][nb] = ld_nt(wp[nb] + kk);
          }
        }
      }
    };
    auto mul = [&](int c, int buf) {
#pragma unroll
      for (int u = 0; u < UN; ++u)
        if (c * UN + u < KS) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[buf][u], wf[buf][u][nb], acc[0][nb], 0, 0, 0);
            acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[buf][u], wf[buf][u][nb], acc[1][nb], 0, 0, 0);
          }
        }
    };
    const int NC = (KS + UN - 1) / UN;
    fetch(0, 0);
    for (int c = 0; c < NC; c += 2) {
      if (c + 1 < NC) fetch(c + 1, 1);
      mul(c, 0);
      if (c + 2 < NC) fetch(c + 2, 0);
      if (c + 1 < NC) mul(c + 1, 1);
    }
  }
  if constexpr ((DBG & 4) != 0) {                                       // probe: no cross-wave reduction, no epilogue arithmetic
    if (wave == 0 && acc[0][0][0] + acc[1][0][0] == 123.456f) ((float*)out)[tid] = acc[0][0][1];
    return;
  }
  // the residual of this thread's output element is requested before the reduction barrier
  constexpr int OUTS = 32 * COLS;                                       // output elements of the workgroup (<= 1024)
  const int e0 = tid, e1 = tid + 512;
  float r0 = 0.f, r1 = 0.f;
  if (EPI == EPI_F32_RES) {
    if (e0 < OUTS) { const int row = e0 / COLS, n = n0 + e0 % COLS; if (row < M && n < N) r0 = res[BLK ? blk_off(row, n) : (long)row * N + n]; }
    if (OUTS > 512 && e1 < OUTS) { const int row = e1 / COLS, n = n0 + e1 % COLS; if (row < M && n < N) r1 = res[BLK ? blk_off(row, n) : (long)row * N + n]; }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[wave][g * 4 + q][nb * 16 + i] = acc[0][nb][q]; red[wave][16 + g * 4 + q][nb * 16 + i] = acc[1][nb][q]; }
  __syncthreads();
  auto finish = [&](int e, float rr) {
    const int row = e / COLS, col = e % COLS, n = n0 + col;
    if (row >= M || n >= N) return;
    if (EPI == EPI_SWIGLU) {
      const int cg = COLS == 16 ? col : col, cu = COLS == 16 ? 16 + col : 8 + col;
      float gv = 0.f, uv = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { gv += red[w][row][cg]; uv += red[w][row][cu]; }
      const float gq = bf2f(f2bf(gv)), uq = bf2f(f2bf(uv));              // the unfused path stores gate|up as bf16 first
      ((bf16_t*)out)[BLK ? blk_off(row, n) : (long)row * N + n] = f2bf(gq / (1.f + __expf(-gq)) * uq);
    } else {
      const int c = (col / CPB) * 16 + col % CPB;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[w][row][c];
      if (EPI == EPI_F32_RES) ((float*)out)[BLK ? blk_off(row, n) : (long)row * N + n] = v + rr;
      else ((bf16_t*)out)[(long)row * N + n] = f2bf(v);
    }
  };
  if (e0 < OUTS) finish(e0, r0);
  if (OUTS > 512 && e1 < OUTS) finish(e1, r1);
}

// the token embeddings of the step into the blocked f32 stream (one wave per clip)
__global__ __launch_bounds__(256) void dec_embed_kernel(const long* __restrict__ ids, const float* __restrict__ emb, float* __restrict__ xblk,
                                                        int B, int D, long vocab) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  long id = ids[row]; if (id < 0) id = 0; if (id >= vocab) id = vocab - 1;
  const float4* src = (const float4*)(emb + id * D);
  for (int c = lane; c < D / 4; c += 64) *(float4*)(xblk + blk_off(row, c * 4)) = src[c];
}
// final RMSNorm of the blocked stream -> bf16 [B, D] row-major (the LM head's input); rmsnorm_fwd_kernel's arithmetic and order
template <int MAXV>
__global__ __launch_bounds__(256) void dec_final_norm_kernel(const float* __restrict__ xblk, const float* __restrict__ w, bf16_t* __restrict__ y,
                                                             int B, int D, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  const int nv = D >> 2;
  float4 v[MAXV];
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < MAXV; ++t) {
    const int c = lane + t * 64;
    if (c < nv) {
      v[t] = *(const float4*)(xblk + blk_off(row, c * 4));
      q += v[t].x * v[t].x + v[t].y * v[t].y + v[t].z * v[t].z + v[t].w * v[t].w;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int t = 0; t < MAXV; ++t) {
    const int c = lane + t * 64;
    if (c < nv) {
      const float4 ww = ((const float4*)w)[c];
      uint2 p; p.x = pack2bf(v[t].x * rstd * ww.x, v[t].y * rstd * ww.y); p.y = pack2bf(v[t].z * rstd * ww.z, v[t].w * rstd * ww.w);
      ((uint2*)(y + (long)row * D))[c] = p;
    }
  }
}

// One workgroup per (kv head, clip), GRP q heads per kv head.  Waves 0..GRP-1 normalise + rotate one q head each, wave GRP the new
// key (appended to the cache), wave GRP+1 copies the new value row (likewise) -- the arithmetic of lm_qkv_post_decode_kernel.  Then
// 16 lanes share a key row (16 B each): scores of all GRP heads from ONE read of K, fp32 softmax, P V from one read of V.
// LDS: q [GRP][128] f32 | new k, v [2][128] bf16-rounded f32 | scores [GRP][Lcap] | partial outputs [16][GRP][128].
template <int GRP>
__global__ __launch_bounds__(256) void dec_attn_kernel(const bf16_t* __restrict__ qkv0, const float* __restrict__ qn_w,
                                                       const float* __restrict__ kn_w, const float* __restrict__ cosT,
                                                       const float* __restrict__ sinT, const int* __restrict__ pos,
                                                       const int* __restrict__ slot_p, const int* __restrict__ kmask,
                                                       bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, bf16_t* __restrict__ out,
                                                       int Hq, int Hkv, int Lmax, int Lcap, float eps, float scale, int out_blocked,
                                                       int n_main_y, DecPf pf) {
  if ((int)blockIdx.y >= n_main_y) {
    dec_prefetch(pf, (blockIdx.y - n_main_y) * gridDim.x + blockIdx.x, (gridDim.y - n_main_y) * gridDim.x, threadIdx.x, 256);
    return;
  }
  extern __shared__ float sm[];
  float* qs = sm;                         // [GRP][128]
  float* nk = qs + GRP * HD;              // [128] new key, [128] new value
  float* nv = nk + HD;
  float* red = nv + HD;                   // [2][GRP][4]
  float* sc = red + 2 * GRP * 4;          // [GRP][Lcap]
  float* oh = sc + GRP * Lcap;            // [16][GRP][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hk = blockIdx.x, b = blockIdx.y;
  const int slot = *slot_p, n = slot + 1;
  const long ld = (long)(Hq + 2 * Hkv) * HD;
  bf16_t* Kb = kc + ((long)b * Hkv + hk) * Lmax * HD;
  bf16_t* Vb = vc + ((long)b * Hkv + hk) * Lmax * HD;
  // ---- the new position: q heads, key, value
  for (int job = wave; job < GRP + 2; job += 4) {
    const int sec = job < GRP ? 0 : (job == GRP ? 1 : 2);
    const int hh = sec == 0 ? hk * GRP + job : (sec == 1 ? Hq + hk : Hq + Hkv + hk);
    const bf16_t* src = qkv0 + (long)b * ld + (long)hh * HD;
    float y1 = bf2f(src[lane]), y2 = bf2f(src[lane + 64]);
    if (sec < 2) {
      const float* nw = sec == 0 ? qn_w : kn_w;
      const float r = rsqrtf(wave_sum(y1 * y1 + y2 * y2) / (float)HD + eps);
      const float n1 = y1 * r * nw[lane], n2 = y2 * r * nw[lane + 64];
      const int p = pos[b];
      const float c = cosT[(long)p * 64 + lane], s = sinT[(long)p * 64 + lane];
      y1 = n1 * c - n2 * s;
      y2 = n2 * c + n1 * s;
    }
    const bf16_t o1 = f2bf(y1), o2 = f2bf(y2);
    if (sec == 0) { qs[job * HD + lane] = bf2f(o1); qs[job * HD + lane + 64] = bf2f(o2); }
    else {
      bf16_t* dst = (sec == 1 ? Kb : Vb) + (long)slot * HD;
      dst[lane] = o1; dst[lane + 64] = o2;
      float* l = sec == 1 ? nk : nv;
      l[lane] = bf2f(o1); l[lane + 64] = bf2f(o2);
    }
  }
  // K rows of the cached positions are requested before the barrier: 16 lanes per row, 16 rows per pass
  const int dc = tid & 15, kg = tid >> 4;
  constexpr int PRE = 16;                                         // passes held in registers (256 keys)
  uint4 kr[PRE];
#pragma unroll
  for (int t = 0; t < PRE; ++t) {
    const int j = kg + 16 * t;
    if (j < slot) kr[t] = *(const uint4*)(Kb + (long)j * HD + dc * 8);
  }
  uint4 vr[PRE];                                                  // ... and the V rows behind them: both streams are in flight at once
#pragma unroll
  for (int t = 0; t < PRE; ++t) {
    const int j = kg + 16 * t;
    if (j < slot) vr[t] = *(const uint4*)(Vb + (long)j * HD + dc * 8);
  }
  __syncthreads();
  float qv[GRP][8];
#pragma unroll
  for (int h = 0; h < GRP; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[h][e] = qs[h * HD + dc * 8 + e];
  auto dots = [&](const uint4& v, float* d) {
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int h = 0; h < GRP; ++h) {
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) a += bf2f((bf16_t)(u[e] & 0xffff)) * qv[h][2 * e] + bf2f((bf16_t)(u[e] >> 16)) * qv[h][2 * e + 1];
      a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 1, 64);
      d[h] = a;
    }
  };
  float mx[GRP];
#pragma unroll
  for (int h = 0; h < GRP; ++h) mx[h] = -INFINITY;
  for (int j0 = 0; j0 < n; j0 += 16 * PRE) {
#pragma unroll
    for (int t = 0; t < PRE; ++t) {
      const int j = j0 + kg + 16 * t;
      if (j < n) {                                                  // uniform over the 16 lanes of a row
        uint4 v;
        if (j == slot) {
          v.x = pack2bf(nk[dc * 8 + 0], nk[dc * 8 + 1]); v.y = pack2bf(nk[dc * 8 + 2], nk[dc * 8 + 3]);
          v.z = pack2bf(nk[dc * 8 + 4], nk[dc * 8 + 5]); v.w = pack2bf(nk[dc * 8 + 6], nk[dc * 8 + 7]);
        } else if (j0 == 0) v = kr[t];
        else v = *(const uint4*)(Kb + (long)j * HD + dc * 8);
        float d[GRP];
        dots(v, d);
        const bool live = kmask[(long)b * Lmax + j] != 0;
#pragma unroll
        for (int h = 0; h < GRP; ++h) {
          const float s = live ? d[h] * scale : -INFINITY;
          if (dc == 0) sc[h * Lcap + j] = s;
          mx[h] = fmaxf(mx[h], s);
        }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < GRP; ++h) {
    const float m = wave_max(mx[h]);
    if (lane == 0) red[h * 4 + wave] = m;
  }
  __syncthreads();
  float sum[GRP];
#pragma unroll
  for (int h = 0; h < GRP; ++h) {
    const float m = fmaxf(fmaxf(red[h * 4], red[h * 4 + 1]), fmaxf(red[h * 4 + 2], red[h * 4 + 3]));
    float s = 0.f;
    for (int j = tid; j < n; j += 256) {
      const float p = sc[h * Lcap + j] == -INFINITY ? 0.f : __expf(sc[h * Lcap + j] - m);
      sc[h * Lcap + j] = p;
      s += p;
    }
    sum[h] = wave_sum(s);
    if (lane == 0) red[GRP * 4 + h * 4 + wave] = sum[h];
  }
  __syncthreads();
  float o[GRP][8];
#pragma unroll
  for (int h = 0; h < GRP; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[h][e] = 0.f;
  for (int j0 = 0; j0 < n; j0 += 16 * PRE) {
#pragma unroll
    for (int t = 0; t < PRE; ++t) {
      const int j = j0 + kg + 16 * t;
      if (j < n) {
        float vf[8];
        if (j == slot) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vf[e] = nv[dc * 8 + e];
        } else {
          const uint4 v = j0 == 0 ? vr[t] : *(const uint4*)(Vb + (long)j * HD + dc * 8);
          const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { vf[2 * e] = bf2f((bf16_t)(u[e] & 0xffff)); vf[2 * e + 1] = bf2f((bf16_t)(u[e] >> 16)); }
        }
#pragma unroll
        for (int h = 0; h < GRP; ++h) {
          const float p = sc[h * Lcap + j];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[h][e] += p * vf[e];
        }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < GRP; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) oh[(kg * GRP + h) * HD + dc * 8 + e] = o[h][e];
  __syncthreads();
  for (int e = tid; e < GRP * HD; e += 256) {
    const int h = e / HD, d = e % HD;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += oh[(k * GRP + h) * HD + d];
    const float inv = 1.0f / (red[GRP * 4 + h * 4] + red[GRP * 4 + h * 4 + 1] + red[GRP * 4 + h * 4 + 2] + red[GRP * 4 + h * 4 + 3]);
    const int col = (hk * GRP + h) * HD + d;
    out[out_blocked ? blk_off(b, col) : (long)b * Hq * HD + col] = f2bf(t * inv);
  }
}

size_t dec_attn_smem(int grp, int Lcap) { return ((size_t)grp * HD + 2 * HD + 2 * grp * 4 + (size_t)grp * Lcap + 16 * grp * HD) * sizeof(float); }
}  // namespace

// ---- launchers (generate.hip).  Each answers TA_ERR_ARG for a shape outside its envelope; the caller then takes the unfused path.
bool ta_i_dec_fused_serves(int B, int D, int F, int bq, int Hq, int Hkv, int Lmax) {
  if (B <= 0 || B > 32) return false;
  if (D % 256 || F % 256 || bq % 256 || D > 2048) return false;            // K split over 8 waves x 32; the norm holds <= 8 k-steps
  if (D % 32 || F % 32) return false;
  const int grp = Hkv > 0 && Hq % Hkv == 0 ? Hq / Hkv : 0;
  if (grp != 1 && grp != 2 && grp != 4) return false;
  return dec_attn_smem(grp, Lmax) <= 64 * 1024;
}

// COLS per workgroup (scripts/probe/dec_probe.hip, profiles/r04_s_dec_probe.txt: q|k|v 8.5 us at 16 columns / 9.2 at 32; gate|up 15.2 at
// 8 (384 workgroups: two per CU on half the chip) / 9.5 at 16; o_proj and down 6.6-7.1 / 8.8-9.7 whatever the count)
#ifndef TA355_DEC_COLS_QKV
#define TA355_DEC_COLS_QKV 16
#define TA355_DEC_COLS_GU 16
#define TA355_DEC_COLS_RES 8
#endif
static DecPf to_pf(const ta_i_dec_prefetch* q) {
  DecPf pf = {nullptr, nullptr, 0, nullptr, 0, 0};
  if (q) { pf.p0 = (const char*)q->p0; pf.p1 = (const char*)q->p1; pf.bytes = q->bytes; pf.slot_p = q->slot_p; pf.row_stride = q->row_stride; pf.rows = q->rows; }
  return pf;
}
// prefetch workgroups: 96 (in situ 1.455 ms per token against 1.464 at 128, 1.59 at 256-512, 1.66 at 64, 1.94 at 32 and 1.50 with "as many
// as fit beside the compute grid in one round"; 1.54 without prefetch: profiles/r04_v_*, r04_w_*, r04_x_*)
static int pf_wgs(const ta_i_dec_prefetch* q, int n_main) {
  (void)n_main;
  if (!(q && q->p0 && (q->bytes > 0 || q->rows > 0))) return 0;
  return q->wgs > 0 ? q->wgs : 96;
}

int ta_i_dec_norm_linear(const float* x, const float* lnw, float eps, const void* W, void* out, int M, int N, int K, bool swiglu,
                         const ta_i_dec_prefetch* next, hipStream_t st) {
  const int ks = K / 256;
  if (M > 32 || K % 256 || ks > 8) return TA_ERR_ARG;
  const dim3 blk(512);
  const DecPf pf = to_pf(next);
#define DL(EPI_, UN_, COLS_) TA_LAUNCH((dec_linear_kernel<true, EPI_, UN_, COLS_, true>), dim3(ta_cdiv(N, COLS_) + pf_wgs(next, ta_cdiv(N, COLS_))), blk, 0, st, (const void*)x, lnw, eps, (const bf16_t*)W, out, (const float*)nullptr, M, N, K, ta_cdiv(N, COLS_), pf)
  constexpr bool qkv32 = true;   // 32 columns = 128 workgroups: with the cache prefetchers beside them 1.455 ms per token against 1.59 at 16 columns (r04_w)
  if (swiglu) { if (ks <= 4) DL(EPI_SWIGLU, 4, TA355_DEC_COLS_GU); else DL(EPI_SWIGLU, 8, TA355_DEC_COLS_GU); }
  else if (qkv32 && N % 32 == 0) { if (ks <= 4) DL(EPI_BF16, 4, 32); else DL(EPI_BF16, 8, 32); }
  else { if (ks <= 4) DL(EPI_BF16, 4, TA355_DEC_COLS_QKV); else DL(EPI_BF16, 8, TA355_DEC_COLS_QKV); }
#undef DL
  TA_CHECK_LAUNCH();
  return TA_OK;
}

int ta_i_dec_linear_res(const void* x, const void* W, float* out, const float* res, int M, int N, int K, const ta_i_dec_prefetch* next,
                        hipStream_t st) {
  const int ks = K / 256;
  if (M > 32 || K % 256 || !res) return TA_ERR_ARG;
  const DecPf pf = to_pf(next);
  const int nm = ta_cdiv(N, TA355_DEC_COLS_RES);
  const dim3 grid(nm + pf_wgs(next, nm)), blk(512);
#define DL(UN_) TA_LAUNCH((dec_linear_kernel<false, EPI_F32_RES, UN_, TA355_DEC_COLS_RES, true>), grid, blk, 0, st, x, (const float*)nullptr, 0.f, (const bf16_t*)W, (void*)out, res, M, N, K, nm, pf)
  if (ks % 6 == 0) DL(6); else if (ks % 4 == 0) DL(4); else if (ks % 2 == 0) DL(2); else DL(1);
#undef DL
  TA_CHECK_LAUNCH();
  return TA_OK;
}

int ta_i_dec_attn(const void* qkv0, const float* qn_w, const float* kn_w, const float* cosT, const float* sinT, const int* pos,
                  const int* slot_dev, const int* kmask, void* kc, void* vc, void* out, int B, int Hq, int Hkv, int Lmax, float eps,
                  float scale, const ta_i_dec_prefetch* next, hipStream_t st) {
  const int grp = Hq / Hkv;
  const size_t smem = dec_attn_smem(grp, Lmax);
  if (smem > 64 * 1024) return TA_ERR_ARG;
  const DecPf pf = to_pf(next);
  const dim3 grid(Hkv, B + ta_cdiv(pf_wgs(next, Hkv * B), Hkv)), blk(256);
#define DA(G_) TA_LAUNCH((dec_attn_kernel<G_>), grid, blk, smem, st, (const bf16_t*)qkv0, qn_w, kn_w, cosT, sinT, pos, slot_dev, kmask, (bf16_t*)kc, (bf16_t*)vc, (bf16_t*)out, Hq, Hkv, Lmax, Lmax, eps, scale, 1, B, pf)
  if (grp == 1) DA(1); else if (grp == 2) DA(2); else if (grp == 4) DA(4); else return TA_ERR_ARG;
#undef DA
  TA_CHECK_LAUNCH();
  return TA_OK;
}

// ids -> blocked f32 stream; blocked stream -> final RMSNorm, bf16 row-major
int ta_i_dec_embed(const long* ids, const float* emb, float* xblk, int B, int D, long vocab, hipStream_t st) {
  if (D % 32) return TA_ERR_ARG;
  TA_LAUNCH(dec_embed_kernel, dim3(ta_cdiv(B, 4)), dim3(256), 0, st, ids, emb, xblk, B, D, vocab);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
int ta_i_dec_final_norm(const float* xblk, const float* w, void* y, int B, int D, float eps, hipStream_t st) {
  if (D % 32 || D > 2048) return TA_ERR_ARG;
  if (D <= 1024) TA_LAUNCH((dec_final_norm_kernel<4>), dim3(ta_cdiv(B, 4)), dim3(256), 0, st, xblk, w, (bf16_t*)y, B, D, eps);
  else TA_LAUNCH((dec_final_norm_kernel<8>), dim3(ta_cdiv(B, 4)), dim3(256), 0, st, xblk, w, (bf16_t*)y, B, D, eps);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
