// ta355 greedy decoding: one new token per clip against the KV cache.
//
// Reference: ASRModel.generate (tiny_audio/asr_modeling.py:562-646) hands inputs_embeds to HF GenerationMixin greedy
// search (TF:generation/utils.py `_sample`, do_sample=False) with a DynamicCache; every step runs Qwen3 on ONE new
// position per clip (TF:models/qwen3/modeling_qwen3.py:211-280 with past_key_values).  Here the prompt pass is
// ta_lm_prefill (api.hip); this file holds the per-token step and the greedy bookkeeping.
//
// Everything a step depends on that changes from token to token (cache slot, RoPE position, key mask, token ids,
// finished flags) lives in DEVICE memory and is advanced by greedy_advance_kernel, so the launch sequence of a step
// is identical every time: no host round trip inside the loop, and the step is hipGraph-capturable.
#include <cstdlib>
#include "host_util.h"
#include "internal.h"

static bool read_decode_fused() { const char* e = getenv("TA355_DECODE_FUSED"); return !(e && *e == '0'); }
static bool g_decode_fused = read_decode_fused();                 // read at load; ta_gemm_reload_knobs() re-reads it (the tests compare both sequences)
static const bool g_decode_prefetch = true;                       // next-kernel prefetch workgroups (1.54 -> 1.455 ms per token, r04)
static const int g_decode_pf_wgs = 0;                             // 0 = decode_fused.hip's default (96: 1.94 ms at 32, 1.46 at 96-128, 1.59 at 256+)
void ta_i_reload_decode_knobs() { g_decode_fused = read_decode_fused(); }

namespace {
constexpr int HD = 128;

// One wave per (clip, q|k|v head): per-head RMSNorm + RoPE at the clip's position (as lm_qkv_post_fwd_kernel), q to a
// dense [B, Hq, 128] buffer, k / v straight into cache slot *slot_p.
__global__ __launch_bounds__(64) void lm_qkv_post_decode_kernel(const bf16_t* __restrict__ qkv0, const float* __restrict__ qn_w,
                                                                const float* __restrict__ kn_w, const float* __restrict__ cosT,
                                                                const float* __restrict__ sinT, const int* __restrict__ pos,
                                                                const int* __restrict__ slot_p, bf16_t* __restrict__ q_out,
                                                                bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, int Hq, int Hkv,
                                                                int Lmax, float eps) {
  const int lane = threadIdx.x, hh = blockIdx.x, b = blockIdx.y;
  const int sec = hh < Hq ? 0 : (hh < Hq + Hkv ? 1 : 2);
  const int head = sec == 0 ? hh : (sec == 1 ? hh - Hq : hh - Hq - Hkv);
  const long ld = (long)(Hq + 2 * Hkv) * HD;
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
  bf16_t* dst;
  if (sec == 0) dst = q_out + ((long)b * Hq + head) * HD;
  else dst = (sec == 1 ? kc : vc) + (((long)b * Hkv + head) * Lmax + *slot_p) * HD;
  dst[lane] = f2bf(y1);
  dst[lane + 64] = f2bf(y2);
}

// One workgroup per (q head, clip): scores of the single query against cache slots 0..*slot_p (key mask applied),
// softmax in fp32, then the probability-weighted sum of the V rows.  HBM-bound: reads the (b, kv head) K and V rows once.
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                          const bf16_t* __restrict__ vc, const int* __restrict__ kmask,
                                                          const int* __restrict__ slot_p, bf16_t* __restrict__ out, int Hq,
                                                          int Hkv, int Lmax, float scale) {
  extern __shared__ float sm[];                    // [Lmax] scores, then [128] q, [8] partials, [16 x 128] output partials
  float* sc = sm;
  float* qs = sm + Lmax;
  float* red = qs + HD;
  float* oh = red + 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y, hk = h / (Hq / Hkv);
  const int n = *slot_p + 1;
  if (tid < HD) qs[tid] = bf2f(q[((long)b * Hq + h) * HD + tid]);
  __syncthreads();
  const bf16_t* K = kc + ((long)b * Hkv + hk) * Lmax * HD;
  const bf16_t* V = vc + ((long)b * Hkv + hk) * Lmax * HD;
  float mx = -INFINITY;
  for (int j = tid; j < n; j += 256) {
    float s = -INFINITY;
    if (kmask[(long)b * Lmax + j]) {
      const uint4* kr = (const uint4*)(K + (long)j * HD);
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) {
        const uint4 v = kr[c];
        const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc += bf2f((bf16_t)(u[e] & 0xffff)) * qs[c * 8 + 2 * e] + bf2f((bf16_t)(u[e] >> 16)) * qs[c * 8 + 2 * e + 1];
      }
      s = acc * scale;
    }
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int j = tid; j < n; j += 256) {
    const float p = sc[j] == -INFINITY ? 0.f : __expf(sc[j] - mx);
    sc[j] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  // 16 key groups x 16 lanes; a lane owns 8 dims (one 16-B load per key row), partial sums meet in LDS
  const int dc = tid & 15, kg = tid >> 4;
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int j = kg; j < n; j += 16) {
    const float p = sc[j];
    const uint4 v = *(const uint4*)(V + (long)j * HD + dc * 8);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] += p * bf2f((bf16_t)(u[e] & 0xffff));
      o[2 * e + 1] += p * bf2f((bf16_t)(u[e] >> 16));
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) oh[kg * HD + dc * 8 + e] = o[e];
  __syncthreads();
  if (tid < HD) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += oh[k * HD + tid];
    out[((long)b * Hq + h) * HD + tid] = f2bf(t * inv);
  }
}

// row-wise argmax over the first n columns (lowest index wins ties, as torch.argmax on a CPU tensor does).
// One workgroup of 1024 lanes per row, float4 loads: a 151 680-column fp32 row is 37 load rounds.
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const float* __restrict__ x, long ld, int n, long* __restrict__ out) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const float* row = x + (long)blockIdx.x * ld;
  float best = -INFINITY; int idx = 0x7fffffff;
  auto take = [&](float v, int j) { if (v > best || (v == best && j < idx)) { best = v; idx = j; } };
  const bool vec = (((size_t)row) & 15) == 0;
  const int n4 = vec ? n / 4 : 0;
  for (int q = threadIdx.x; q < n4; q += 1024) {
    const float4 v = ((const float4*)row)[q];
    take(v.x, 4 * q); take(v.y, 4 * q + 1); take(v.z, 4 * q + 2); take(v.w, 4 * q + 3);
  }
  for (int j = n4 * 4 + threadIdx.x; j < n; j += 1024) take(row[j], j);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    out[blockIdx.x] = idx == 0x7fffffff ? 0 : idx;
  }
}

// HF logits processors of the reference's generation settings (tiny_audio/asr_config.py:84-86,155-160, forwarded to
// language_model.generate at tiny_audio/asr_modeling.py:627-633 together with input_ids, so the processors see the PROMPT
// ids followed by the generated tokens), applied to the f32 logits of the step before the argmax:
//   RepetitionPenaltyLogitsProcessor (TF:generation/logits_process.py): every token id of the sequence so far gets
//       score < 0 ? score * penalty : score / penalty -- once, however often it occurs (gather, then scatter);
//   NoRepeatNGramLogitsProcessor(n): with the last n-1 tokens as the prefix, every token that followed an earlier occurrence
//       of that prefix is banned (-inf); nothing is banned while the sequence is shorter than n-1 tokens.
// One workgroup per clip.  The sequence length comes from *step_p in device memory (prompt length + tokens generated so
// far), so the launch arguments never change and the step stays hipGraph-capturable.
#define LP_MAXSEQ 8192
__global__ __launch_bounds__(1024) void logits_process_kernel(float* __restrict__ logits, long ld, int V, const long* __restrict__ prompt,
                                                              int L, const long* __restrict__ out_seq, int max_new,
                                                              const int* __restrict__ step_p, float penalty, int ngram) {
  __shared__ float val[LP_MAXSEQ];
  const int b = blockIdx.x, t = *step_p, n = L + (t < max_new ? t : max_new);
  float* row = logits + (long)b * ld;
  const long* pr = prompt + (long)b * L;
  const long* gen = out_seq + (long)b * max_new;
  auto tok = [&](int i) -> long { return i < L ? pr[i] : gen[i - L]; };
  if (penalty != 1.0f) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {              // gather the ORIGINAL scores first ...
      const long id = tok(i);
      float sc = 0.f;
      if (id >= 0 && id < V) { sc = row[id]; sc = sc < 0.f ? sc * penalty : sc / penalty; }
      val[i] = sc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {              // ... then scatter: a repeated id is penalised once
      const long id = tok(i);
      if (id >= 0 && id < V) row[id] = val[i];
    }
    __syncthreads();
  }
  if (ngram > 0 && n + 1 >= ngram) {
    const int pre = ngram - 1;                                        // prefix = tokens n-pre .. n-1
    for (int i = threadIdx.x; i + pre < n; i += blockDim.x) {        // the n-gram starting at i ends at i + pre <= n - 1
      bool same = true;
      for (int j = 0; j < pre; ++j) same &= tok(i + j) == tok(n - pre + j);
      const long id = tok(i + pre);
      if (same && id >= 0 && id < V) row[id] = -INFINITY;
    }
  }
}

// HF greedy bookkeeping (TF:generation/utils.py _sample): finished clips emit pad; a clip finishes when it emits an
// eos id; record the token, make it the next input, advance its position, open the next cache slot in the key mask.
__global__ void greedy_advance_kernel(const long* __restrict__ amax, const long* __restrict__ eos, int n_eos, long pad_id,
                                      int* __restrict__ finished, long* __restrict__ next_ids, long* __restrict__ out_seq,
                                      int max_new, int* __restrict__ step_p, int* __restrict__ slot_p, int* __restrict__ pos,
                                      int* __restrict__ kmask, int Lmax, int B, int* __restrict__ n_unfinished) {
  __shared__ int alive;
  const int t = *step_p, slot = *slot_p;
  if (threadIdx.x == 0) alive = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    long tok = finished[b] ? pad_id : amax[b];
    if (t < max_new) out_seq[(long)b * max_new + t] = tok;
    int fin = finished[b];
    for (int e = 0; e < n_eos; ++e) fin |= (tok == eos[e]);
    finished[b] = fin;
    next_ids[b] = tok;
    // the token just recorded is the input of the next decode step: it sits at position pos[b] in cache slot `ns`.
    // After the prompt pass (t == 0) pos / slot already point there (prompt length / L); afterwards both advance.
    if (t > 0) pos[b] += 1;
    const int ns = t > 0 ? slot + 1 : slot;
    if (ns < Lmax) kmask[(long)b * Lmax + ns] = 1;
    if (!fin) atomicAdd(&alive, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) { *step_p = t + 1; *slot_p = t > 0 ? slot + 1 : slot; *n_unfinished = alive; }
}

// out[M, N] = X[M, K] W[N, K]^T (+ A2[M, 64] W2[N, 64]^T) (+ residual) for M <= 32: the decode-step linears.
// A 128x128 tile grid leaves 8..48 workgroups each walking all of K serially (the step was latency-bound at
// ~5 ms/token); here a workgroup owns 64 output columns, its eight waves split K, fragments come straight from
// global memory (both operands are K-major) and the partial sums meet in LDS.  The kernel streams W once: HBM-bound.
typedef __attribute__((ext_vector_type(8))) short gbf16x8;
typedef __attribute__((ext_vector_type(4))) float gf32x4;
template <bool OUT_BF16, int CB>      // CB = 16-column blocks per workgroup (4: wide layers and the LM head; 1: N = hidden)
__global__ __launch_bounds__(512) void linear_small_m_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W,
                                                             void* __restrict__ out, const float* __restrict__ res, int M,
                                                             int N, int K, const bf16_t* __restrict__ A2,
                                                             const bf16_t* __restrict__ W2) {
  constexpr int COLS = CB * 16, UN = 4;                                   // UN k-steps of loads in flight per wave
  __shared__ float red[8][32][COLS + 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * COLS;
  gf32x4 acc[2][CB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < CB; ++b) acc[a][b] = (gf32x4){0.f, 0.f, 0.f, 0.f};
  const int ra = min(i, M - 1), rb = min(16 + i, M - 1);              // clamped rows are never stored
  const bf16_t* xa = X + (long)ra * K + g * 8;
  const bf16_t* xb = X + (long)rb * K + g * 8;
  const bf16_t* wp[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) wp[cb] = W + (long)min(n0 + cb * 16 + i, N - 1) * K + g * 8;
  const bool two = M > 16;
  for (int k0 = wave * 32; k0 < K; k0 += 256 * UN) {
    gbf16x8 a0[UN], a1[UN], b[UN][CB];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kk = k0 + u * 256;
      if (kk < K) {
        a0[u] = *(const gbf16x8*)(xa + kk);
        a1[u] = two ? *(const gbf16x8*)(xb + kk) : a0[u];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) b[u][cb] = *(const gbf16x8*)(wp[cb] + kk);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (k0 + u * 256 < K) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[u], b[u][cb], acc[0][cb], 0, 0, 0);
          if (two) acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[u], b[u][cb], acc[1][cb], 0, 0, 0);
        }
      }
    }
  }
  if (A2 && wave < 2) {                                                   // the adapter's 64-wide K tile: waves 0, 1
    const int kk = wave * 32 + g * 8;
    const gbf16x8 a0 = *(const gbf16x8*)(A2 + (long)ra * 64 + kk), a1 = *(const gbf16x8*)(A2 + (long)rb * 64 + kk);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const gbf16x8 b = *(const gbf16x8*)(W2 + (long)min(n0 + cb * 16 + i, N - 1) * 64 + kk);
      acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc[0][cb], 0, 0, 0);
      acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc[1][cb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int rbk = 0; rbk < 2; ++rbk)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[wave][rbk * 16 + g * 4 + q][cb * 16 + i] = acc[rbk][cb][q];
  __syncthreads();
  for (int e = tid; e < M * COLS; e += 512) {
    const int row = e / COLS, col = e % COLS, n = n0 + col;
    if (n >= N) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[w][row][col];
    if (res) v += res[(long)row * N + n];
    if (OUT_BF16) ((bf16_t*)out)[(long)row * N + n] = f2bf(v);
    else ((float*)out)[(long)row * N + n] = v;
  }
}

int linear_small_m(const bf16_t* X, const bf16_t* W, void* out, int M, int N, int K, const float* res, bool out_bf16,
                   const bf16_t* A2, const bf16_t* W2, hipStream_t st) {
  if (M > 32 || K % 32) return TA_ERR_ARG;
  const bool narrow = N <= 2048;           // few columns: 16 per workgroup keeps >= 64 workgroups streaming
  const dim3 grid(ta_cdiv(N, narrow ? 16 : 64));
  if (out_bf16) {
    if (narrow) TA_LAUNCH((linear_small_m_kernel<true, 1>), grid, dim3(512), 0, st, X, W, out, res, M, N, K, A2, W2);
    else TA_LAUNCH((linear_small_m_kernel<true, 4>), grid, dim3(512), 0, st, X, W, out, res, M, N, K, A2, W2);
  } else {
    if (narrow) TA_LAUNCH((linear_small_m_kernel<false, 1>), grid, dim3(512), 0, st, X, W, out, res, M, N, K, A2, W2);
    else TA_LAUNCH((linear_small_m_kernel<false, 4>), grid, dim3(512), 0, st, X, W, out, res, M, N, K, A2, W2);
  }
  TA_CHECK_LAUNCH();
  return TA_OK;
}

struct DecodeWs {
  float *x, *x1, *r, *logits_unused;
  bf16_t *xn, *qkv0, *q, *ao, *gu, *act, *xa, *hn;
  size_t bytes;
};
DecodeWs decode_ws(const ta_lm_weights* w, int B, void* base) {
  const int D = w->hidden, F = w->ffn, bq = w->heads * HD, NQKV = (w->heads + 2 * w->kv_heads) * HD;
  Carver c(base);
  DecodeWs s;
  const size_t Bp = B < 32 ? 32 : B;                 // the fused step's blocked activations always hold 32 rows
  s.x = c.take<float>(Bp * D); s.x1 = c.take<float>(Bp * D); s.r = c.take<float>((size_t)B);
  s.xn = c.take<bf16_t>((size_t)B * D); s.qkv0 = c.take<bf16_t>((size_t)B * NQKV); s.q = c.take<bf16_t>((size_t)B * bq);
  s.ao = c.take<bf16_t>(Bp * bq); s.gu = c.take<bf16_t>((size_t)B * 2 * F); s.act = c.take<bf16_t>(Bp * F);
  s.xa = c.take<bf16_t>((size_t)B * 64); s.hn = c.take<bf16_t>((size_t)B * D);
  s.bytes = c.total();
  return s;
}
}  // namespace

extern "C" long ta_lm_decode_workspace_bytes(const ta_lm_weights* w, int B) { return (long)decode_ws(w, B, nullptr).bytes; }

// One decoding step: ids [B] (device) are the tokens emitted by the previous step; their keys / values are appended
// at cache slot *slot_dev, and logits [B, vocab_pad] (fp32) of the NEXT token come back.
extern "C" int ta_lm_decode_step(const ta_lm_weights* w, const long* ids, const int* pos, const int* kmask,
                                 const int* slot_dev, int B, void* kcache, void* vcache, int Lmax, float* logits,
                                 const void* lora_img, void* ws, long ws_bytes, hipStream_t st) {
  if (B <= 0) return TA_OK;
  if (w->head_dim != HD || w->n_layers > 64 || !ids || !pos || !kmask || !slot_dev || !kcache || !vcache || !logits ||
      w->heads % w->kv_heads)
    return TA_ERR_ARG;
  const bool lora = w->lora_rank > 0;
  if (lora && !lora_img) return TA_ERR_ARG;
  const int D = w->hidden, F = w->ffn, Hq = w->heads, Hkv = w->kv_heads, bq = Hq * HD, NQKV = (Hq + 2 * Hkv) * HD;
  DecodeWs s = decode_ws(w, B, ws);
  if ((long)s.bytes > ws_bytes) return TA_ERR_ARG;
  ta_i_lora_layer_imgs imgs[64];
  if (lora) lora_imgs_carve(w, (void*)lora_img, imgs);
  const int lgm = lora ? (w->lora_groups ? w->lora_groups : 15) : 0;
  const float scale = 1.0f / sqrtf((float)HD);
  const size_t layer_elems = (size_t)B * Hkv * Lmax * HD;
  const size_t smem = ((size_t)Lmax + HD + 8 + 16 * HD) * sizeof(float);
  if (smem > 64 * 1024) return TA_ERR_ARG;          // Lmax up to ~16000 cache slots
  // one adapted linear: y = x W^T (+ xa Bext^T, xa = x (s Acat)^T) (+ residual); small batches take the K-split
  // streaming kernel, larger ones the tile GEMM with its K extension
  const bool small = B <= 32;
  auto linear = [&](const bf16_t* x, const void* Wm, void* y, int N, int K, const float* res, bool out_bf16,
                    const LoraImg* g) -> int {
    const bf16_t *a2 = nullptr, *w2 = nullptr;
    if (g) {
      RC(ta_i_lora_skinny_nt(x, K, g->a, s.xa, B, 64, st));   // (decode: M = batch rows; the full 64-wide image)
      a2 = s.xa; w2 = g->b;
    }
    if (small) return linear_small_m(x, (const bf16_t*)Wm, y, B, N, K, res, out_bf16, a2, w2, st);
    return gemm_opt(x, Wm, y, B, N, K, nullptr, res, 0, out_bf16 ? 1 : 0, g ? opts_kext(a2, w2) : opts_none(), st);
  };
  // round 4: five launches per layer instead of nine (csrc/decode_fused.hip) when no adapter is attached and the shapes fit;
  // TA355_DECODE_FUSED=0 keeps the round-3 sequence (the A/B of profiles/r04_decode_*)
  const bool fused = g_decode_fused && !lora && ta_i_dec_fused_serves(B, D, F, bq, Hq, Hkv, Lmax);
  if (fused) RC(ta_i_dec_embed(ids, w->embed_f32, s.x, B, D, w->vocab, st));
  else RC(ta_embed_scatter(ids, nullptr, w->embed_f32, nullptr, s.x, nullptr, B, D, w->vocab, st));
  for (int l = 0; l < w->n_layers; ++l) {
    const ta_lm_layer& Lw = w->layers[l];
    bf16_t* kc = (bf16_t*)kcache + (size_t)l * layer_elems;
    bf16_t* vc = (bf16_t*)vcache + (size_t)l * layer_elems;
    if (fused) {
      // every kernel also fetches what the next one streams (weights / this layer's cache rows): TA355_DECODE_PREFETCH=0 turns it off
      const int pw = g_decode_pf_wgs;
      const ta_i_dec_prefetch p_kv = {kc, vc, 0, slot_dev, (long)Lmax * HD * 2, B * Hkv, pw};
      const ta_i_dec_prefetch p_o = {Lw.wo, nullptr, (long)D * bq * 2, nullptr, 0, 0, pw};
      const ta_i_dec_prefetch p_gu = {Lw.wgu, nullptr, (long)2 * F * D * 2, nullptr, 0, 0, pw};
      const ta_i_dec_prefetch p_d = {Lw.wd, nullptr, (long)D * F * 2, nullptr, 0, 0, pw};
      const bool last = l + 1 == w->n_layers;
      const long head_bytes = (long)w->vocab_pad * D * 2;              // (the first 16 MB of the LM head behind the last layer)
      const ta_i_dec_prefetch p_next = {last ? w->embed_bf16 : w->layers[l + 1].wqkv, nullptr,
                                        last ? (head_bytes < ((long)16 << 20) ? head_bytes : (long)16 << 20) : (long)NQKV * D * 2, nullptr, 0, 0, pw};
      const bool pfon = g_decode_prefetch;
      RC(ta_i_dec_norm_linear(s.x, Lw.ln_in_w, w->eps, Lw.wqkv, s.qkv0, B, NQKV, D, false, pfon ? &p_kv : nullptr, st));
      RC(ta_i_dec_attn(s.qkv0, Lw.qn_w, Lw.kn_w, w->rope_cos, w->rope_sin, pos, slot_dev, kmask, kc, vc, s.ao, B, Hq, Hkv, Lmax,
                       w->eps, scale, pfon ? &p_o : nullptr, st));
      RC(ta_i_dec_linear_res(s.ao, Lw.wo, s.x1, s.x, B, D, bq, pfon ? &p_gu : nullptr, st));
      RC(ta_i_dec_norm_linear(s.x1, Lw.ln_post_w, w->eps, Lw.wgu, s.act, B, F, D, true, pfon ? &p_d : nullptr, st));
      RC(ta_i_dec_linear_res(s.act, Lw.wd, s.x, s.x1, B, D, F, pfon ? &p_next : nullptr, st));
      continue;
    }
    RC(ta_rmsnorm_fwd(s.x, Lw.ln_in_w, s.xn, nullptr, s.r, B, D, w->eps, 0, st));
    RC(linear(s.xn, Lw.wqkv, s.qkv0, NQKV, D, nullptr, true, (lgm & 1) ? &imgs[l].g[0] : nullptr));
    TA_LAUNCH(lm_qkv_post_decode_kernel, dim3(Hq + 2 * Hkv, B), dim3(64), 0, st, s.qkv0, Lw.qn_w, Lw.kn_w, w->rope_cos,
              w->rope_sin, pos, slot_dev, s.q, kc, vc, Hq, Hkv, Lmax, w->eps);
    TA_CHECK_LAUNCH();
    TA_LAUNCH(attn_decode_kernel, dim3(Hq, B), dim3(256), smem, st, s.q, kc, vc, kmask, slot_dev, s.ao, Hq, Hkv, Lmax, scale);
    TA_CHECK_LAUNCH();
    RC(linear(s.ao, Lw.wo, s.x1, D, bq, s.x, false, (lgm & 2) ? &imgs[l].g[1] : nullptr));
    RC(ta_rmsnorm_fwd(s.x1, Lw.ln_post_w, s.xn, nullptr, s.r, B, D, w->eps, 0, st));
    RC(linear(s.xn, Lw.wgu, s.gu, 2 * F, D, nullptr, true, (lgm & 4) ? &imgs[l].g[2] : nullptr));
    RC(ta_swiglu_fwd(s.gu, s.act, B, F, st));
    RC(linear(s.act, Lw.wd, s.x, D, F, s.x1, false, (lgm & 8) ? &imgs[l].g[3] : nullptr));
  }
  if (fused) RC(ta_i_dec_final_norm(s.x, w->norm_w, s.hn, B, D, w->eps, st));
  else RC(ta_rmsnorm_fwd(s.x, w->norm_w, s.hn, nullptr, s.r, B, D, w->eps, 0, st));
  RC(linear(s.hn, w->embed_bf16, logits, w->vocab_pad, D, nullptr, false, nullptr));
  return TA_OK;
}

extern "C" int ta_argmax_f32(const float* x, long ld, int n, int rows, long* out, hipStream_t st) {
  if (rows <= 0) return TA_OK;
  if (n <= 0) return TA_ERR_ARG;
  TA_LAUNCH(argmax_rows_kernel, dim3(rows), dim3(1024), 0, st, x, ld, n, out);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_logits_process(float* logits, long ld, int V, const long* prompt_ids, int L, const long* out_seq, int max_new,
                                 const int* step_dev, int B, float repetition_penalty, int no_repeat_ngram_size, hipStream_t st) {
  if (B <= 0 || (repetition_penalty == 1.0f && no_repeat_ngram_size <= 0)) return TA_OK;
  if (L + max_new > LP_MAXSEQ || repetition_penalty <= 0.f || no_repeat_ngram_size < 0) return TA_ERR_ARG;
  TA_LAUNCH(logits_process_kernel, dim3(B), dim3(1024), 0, st, logits, ld, V, prompt_ids, L, out_seq, max_new, step_dev,
            repetition_penalty, no_repeat_ngram_size);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

// HF MinNewTokensLengthLogitsProcessor (TF:generation/logits_process.py; generation_config.min_new_tokens, tiny_audio/asr_config.py:83,
// forwarded at tiny_audio/asr_modeling.py:631-637): while fewer than `min_new` tokens have been generated, every eos id scores -inf.
// The count comes from *step_p in device memory, so the launch arguments never change (hipGraph-capturable).
__global__ void logits_suppress_kernel(float* __restrict__ logits, long ld, int V, const long* __restrict__ ids, int n_ids, int min_new,
                                       const int* __restrict__ step_p, int B) {
  if (*step_p >= min_new) return;
  for (int i = threadIdx.x; i < B * n_ids; i += blockDim.x) {
    const long id = ids[i % n_ids];
    if (id >= 0 && id < V) logits[(long)(i / n_ids) * ld + id] = -INFINITY;
  }
}
extern "C" int ta_logits_suppress_until(float* logits, long ld, int V, const long* ids, int n_ids, int min_new, const int* step_dev, int B,
                                        hipStream_t st) {
  if (B <= 0 || n_ids <= 0 || min_new <= 0) return TA_OK;
  if (!logits || !ids || !step_dev) return TA_ERR_ARG;
  TA_LAUNCH(logits_suppress_kernel, dim3(1), dim3(256), 0, st, logits, ld, V, ids, n_ids, min_new, step_dev, B);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

// ---- sampling (generation_config.do_sample / temperature / top_k / top_p: tiny_audio/asr_config.py:78-81, forwarded to HF at
// tiny_audio/asr_modeling.py:631-637).  HF order (TF:generation/utils.py _get_logits_processor + _sample): the processors above, then the
// warpers TemperatureLogitsWarper (scores / T), TopKLogitsWarper (scores below the k-th largest -> -inf, ties kept), TopPLogitsWarper
// (ascending cumulative probability <= 1 - top_p -> -inf, the largest always kept), then softmax + multinomial.
// One 1024-thread workgroup per row.  The two thresholds are found by bisection on the order-preserving integer image of the floats
// (<= 32 counting / mass passes over the row each; the row stays in L2), so no sort is needed and ties behave as in HF.
__device__ __forceinline__ int w_f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float block_sum_1024(float v, float* red) {      // red: [17] floats of LDS
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += red[w];
  return t;
}
__device__ __forceinline__ float block_max_1024(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) t = fmaxf(t, red[w]);
  return t;
}
// WCAP: survivors of the top-k cut that fit the LDS candidate list (the bisection narrows the row to <= WCAP candidates with
// full-row counting passes, then finishes on the list: ~12 + 2 passes over the 600 KB row instead of 64)
#define WCAP 4096
__global__ __launch_bounds__(1024) void logits_warp_kernel(float* __restrict__ logits, long ld, int V, float temperature, int top_k,
                                                           float top_p) {
  __shared__ float red[17];
  __shared__ float cand[WCAP];
  __shared__ int ncand;
  float* row = logits + (long)blockIdx.x * ld;
  const int tid = threadIdx.x;
  float mx = -INFINITY, mn = INFINITY;
  for (int j = tid; j < V; j += 1024) {
    float v = row[j];
    if (temperature != 1.0f) { v = v / temperature; row[j] = v; }
    if (v > -INFINITY) { mx = fmaxf(mx, v); mn = fminf(mn, v); }
  }
  mx = block_max_1024(mx, red);
  mn = -block_max_1024(-mn, red);
  if (!(mx > -INFINITY)) return;                                  // an all -inf row: nothing to do
  bool listed = false;                                            // the survivors of the top-k cut are in cand[0, ncand)
  if (top_k > 0 && top_k < V) {
    // the k-th largest value = the largest ordered integer o with count(x >= o) >= k.  Full-row passes until the candidates fit the list
    long lo = w_f2ord(mn), hi = w_f2ord(mx);
    float c_lo = (float)V;                                        // count(x >= lo) (an upper bound before the first pass)
    while (lo < hi && (c_lo > (float)WCAP || top_k > WCAP)) {
      const long mid = lo + (hi - lo + 1) / 2;
      float c = 0.f;
      for (int j = tid; j < V; j += 1024) { const float v = row[j]; c += (v > -INFINITY && w_f2ord(v) >= mid) ? 1.f : 0.f; }
      c = block_sum_1024(c, red);
      if (c >= (float)top_k) { lo = mid; c_lo = c; } else hi = mid - 1;
    }
    if (lo < hi) {                                                // <= WCAP values >= lo: list them, finish on the list
      if (tid == 0) ncand = 0;
      __syncthreads();
      for (int j = tid; j < V; j += 1024) { const float v = row[j]; if (v > -INFINITY && w_f2ord(v) >= lo) { const int q = atomicAdd(&ncand, 1); if (q < WCAP) cand[q] = v; } }
      __syncthreads();
      const int nc = ncand < WCAP ? ncand : WCAP;
      while (lo < hi) {
        const long mid = lo + (hi - lo + 1) / 2;
        float c = 0.f;
        for (int q = tid; q < nc; q += 1024) c += w_f2ord(cand[q]) >= mid ? 1.f : 0.f;
        c = block_sum_1024(c, red);
        if (c >= (float)top_k) lo = mid; else hi = mid - 1;
      }
      listed = true;
    }
    __syncthreads();
    for (int j = tid; j < V; j += 1024) if (w_f2ord(row[j]) < lo) row[j] = -INFINITY;
    if (listed) {                                                 // keep only the survivors in the list (values; order is irrelevant)
      const int nc = ncand < WCAP ? ncand : WCAP;
      for (int q = tid; q < nc; q += 1024) if (w_f2ord(cand[q]) < lo) cand[q] = -INFINITY;
    }
    __syncthreads();
    mn = INFINITY;
    if (listed) { const int nc = ncand < WCAP ? ncand : WCAP; for (int q = tid; q < nc; q += 1024) { const float v = cand[q]; if (v > -INFINITY) mn = fminf(mn, v); } }
    else for (int j = tid; j < V; j += 1024) { const float v = row[j]; if (v > -INFINITY) mn = fminf(mn, v); }
    mn = -block_max_1024(-mn, red);
  }
  if (top_p < 1.0f) {
    // mass over the surviving scores: from the list when there is one, else over the row
    const int nc = listed ? (ncand < WCAP ? ncand : WCAP) : 0;
    auto mass_le = [&](long bound, bool all) {
      float m = 0.f;
      if (listed) { for (int q = tid; q < nc; q += 1024) { const float v = cand[q]; m += (v > -INFINITY && (all || w_f2ord(v) <= bound)) ? __expf(v - mx) : 0.f; } }
      else for (int j = tid; j < V; j += 1024) { const float v = row[j]; m += (v > -INFINITY && (all || w_f2ord(v) <= bound)) ? __expf(v - mx) : 0.f; }
      return block_sum_1024(m, red);
    };
    const float z = mass_le(0, true);
    // smallest ordered integer o with mass(x <= o) > 1 - top_p: everything below it is removed (the maximum has mass 1: always kept)
    const float cut = (1.0f - top_p) * z;
    long lo = w_f2ord(mn), hi = w_f2ord(mx);
    while (lo < hi) {
      const long mid = lo + (hi - lo) / 2;
      if (mass_le(mid, false) > cut) hi = mid; else lo = mid + 1;
    }
    __syncthreads();
    for (int j = tid; j < V; j += 1024) if (w_f2ord(row[j]) < hi) row[j] = -INFINITY;
  }
}
extern "C" int ta_logits_warp(float* logits, long ld, int V, int B, float temperature, int top_k, float top_p, hipStream_t st) {
  if (B <= 0) return TA_OK;
  if (!logits || V <= 0 || !(temperature > 0.f) || top_k < 0 || !(top_p > 0.f) || top_p > 1.0f) return TA_ERR_ARG;
  if (temperature == 1.0f && (top_k == 0 || top_k >= V) && top_p >= 1.0f) return TA_OK;
  TA_LAUNCH(logits_warp_kernel, dim3(B), dim3(1024), 0, st, logits, ld, V, temperature, top_k, top_p);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

// Philox4x32-10 (Salmon et al. 2011): counter (step, row, 0, 0), key = the 64-bit seed -> one uniform in [0, 1) per row and step
__device__ __forceinline__ float philox_uniform(unsigned long long seed, unsigned step, unsigned row) {
  unsigned c0 = step, c1 = row, c2 = 0u, c3 = 0u, k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}
// out[b] ~ softmax(logits[b, :V]) (multinomial, one draw): inverse CDF in INDEX order -- thread t owns the contiguous chunk
// [t * per, (t + 1) * per), an exclusive scan of the chunk masses finds the chunk, its owner walks it.
__global__ __launch_bounds__(1024) void sample_rows_kernel(const float* __restrict__ logits, long ld, int V, unsigned long long seed,
                                                           const int* __restrict__ step_p, long* __restrict__ out) {
  __shared__ float red[17];
  __shared__ float part[1024];
  __shared__ int pick;
  const float* row = logits + (long)blockIdx.x * ld;
  const int tid = threadIdx.x, per = (V + 1023) / 1024, j0 = tid * per, j1 = min(V, j0 + per);
  float mx = -INFINITY;
  for (int j = tid; j < V; j += 1024) mx = fmaxf(mx, row[j]);
  mx = block_max_1024(mx, red);
  float s = 0.f;
  int last = -1;                                                 // last index of this chunk with a non-zero probability
  for (int j = j0; j < j1; ++j) { const float p = __expf(row[j] - mx); s += p; if (p > 0.f) last = j; }
  part[tid] = s;
  if (tid == 0) pick = -1;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {                            // inclusive scan of the chunk masses
    const float v = tid >= o ? part[tid - o] : 0.f;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  const float total = part[1023];
  const float u = philox_uniform(seed, (unsigned)*step_p, blockIdx.x) * total;
  // The claim must be UNIQUE: part[] comes from a Hillis-Steele scan whose prefixes are summed in different association orders per
  // index, so it is not guaranteed non-decreasing and the intervals [part[t-1], part[t]) can overlap by a rounding when a chunk's
  // mass is tiny -- two owners would then race on `pick`.  The owner is the SMALLEST chunk index t with mass whose
  // inclusive prefix exceeds u (a block-wide minimum), and only that thread walks its chunk.
  __shared__ int owner;
  if (tid == 0) owner = 1024;
  __syncthreads();
  {
    int mine = (s > 0.f && u < part[tid]) ? tid : 1024;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine = min(mine, __shfl_xor(mine, o, 64));
    if ((tid & 63) == 0 && mine < 1024) atomicMin(&owner, mine);
  }
  __syncthreads();
  if (tid == owner) {
    // the walk starts from the previous chunk's inclusive prefix as read from the scan (the same float its owner compared with u)
    float c = tid ? part[tid - 1] : 0.f; int choice = last;
    for (int j = j0; j < j1; ++j) { const float p = __expf(row[j] - mx); c += p; if (p > 0.f && u < c) { choice = j; break; } }
    pick = choice;
  }
  __syncthreads();
  if (pick < 0) {                                                 // rounding left the draw behind the last mass: the last token that has any
    int cand = last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = max(cand, __shfl_xor(cand, o, 64));
    if ((tid & 63) == 0) atomicMax(&pick, cand);
    __syncthreads();
  }
  if (tid == 0) out[blockIdx.x] = pick < 0 ? 0 : pick;
}
extern "C" int ta_sample_f32(const float* logits, long ld, int V, int B, unsigned long long seed, const int* step_dev, long* out,
                             hipStream_t st) {
  if (B <= 0) return TA_OK;
  if (!logits || V <= 0 || !step_dev || !out) return TA_ERR_ARG;
  TA_LAUNCH(sample_rows_kernel, dim3(B), dim3(1024), 0, st, logits, ld, V, seed, step_dev, out);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_greedy_advance(const long* amax, const long* eos_ids, int n_eos, long pad_id, int* finished, long* next_ids,
                                 long* out_seq, int max_new, int* step_dev, int* slot_dev, int* pos, int* kmask, int Lmax,
                                 int B, int* n_unfinished, hipStream_t st) {
  if (B <= 0) return TA_OK;
  TA_LAUNCH(greedy_advance_kernel, dim3(1), dim3(256), 0, st, amax, eos_ids, n_eos, pad_id, finished, next_ids, out_seq,
            max_new, step_dev, slot_dev, pos, kmask, Lmax, B, n_unfinished);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
