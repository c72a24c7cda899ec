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
#include "host_util.h"

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
  extern __shared__ float sm[];                    // [Lmax] scores, then [128] q, [8] partials, [256] output halves
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
  // threads 0..127 take even keys, 128..255 odd keys, dim d = tid & 127 (coalesced V rows)
  const int d = tid & 127, par = tid >> 7;
  float o = 0.f;
  for (int j = par; j < n; j += 2) o += sc[j] * bf2f(V[(long)j * HD + d]);
  oh[tid] = o;
  __syncthreads();
  if (tid < HD) out[((long)b * Hq + h) * HD + tid] = f2bf((oh[tid] + oh[tid + 128]) * inv);
}

// row-wise argmax over the first n columns (lowest index wins ties, as torch.argmax on a CPU tensor does)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long ld, int n, long* __restrict__ out) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const float* row = x + (long)blockIdx.x * ld;
  float best = -INFINITY; int idx = 0x7fffffff;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float v = row[j];
    if (v > best || (v == best && j < idx)) { best = v; idx = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    out[blockIdx.x] = idx == 0x7fffffff ? 0 : idx;
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

struct DecodeWs {
  float *x, *x1, *r, *logits_unused;
  bf16_t *xn, *qkv0, *q, *ao, *gu, *act, *xa, *hn;
  size_t bytes;
};
DecodeWs decode_ws(const ta_lm_weights* w, int B, void* base) {
  const int D = w->hidden, F = w->ffn, bq = w->heads * HD, NQKV = (w->heads + 2 * w->kv_heads) * HD;
  Carver c(base);
  DecodeWs s;
  s.x = c.take<float>((size_t)B * D); s.x1 = c.take<float>((size_t)B * D); s.r = c.take<float>((size_t)B);
  s.xn = c.take<bf16_t>((size_t)B * D); s.qkv0 = c.take<bf16_t>((size_t)B * NQKV); s.q = c.take<bf16_t>((size_t)B * bq);
  s.ao = c.take<bf16_t>((size_t)B * bq); s.gu = c.take<bf16_t>((size_t)B * 2 * F); s.act = c.take<bf16_t>((size_t)B * F);
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
  const float scale = 1.0f / sqrtf((float)HD);
  const size_t layer_elems = (size_t)B * Hkv * Lmax * HD;
  const size_t smem = ((size_t)Lmax + HD + 8 + 256) * sizeof(float);
  if (smem > 64 * 1024) return TA_ERR_ARG;          // Lmax up to ~16000 cache slots
  auto lora_fwd = [&](const bf16_t* x, int in, const LoraImg& g) -> int {
    RC(ta_i_lora_skinny_nt(x, in, g.a, s.xa, B, st));
    return ta_gemm_set_k_extension(s.xa, g.b, 64, 64);
  };
  RC(ta_embed_scatter(ids, nullptr, w->embed_f32, nullptr, s.x, nullptr, B, D, w->vocab, st));
  for (int l = 0; l < w->n_layers; ++l) {
    const ta_lm_layer& Lw = w->layers[l];
    bf16_t* kc = (bf16_t*)kcache + (size_t)l * layer_elems;
    bf16_t* vc = (bf16_t*)vcache + (size_t)l * layer_elems;
    RC(ta_rmsnorm_fwd(s.x, Lw.ln_in_w, s.xn, nullptr, s.r, B, D, w->eps, 0, st));
    if (lora) RC(lora_fwd(s.xn, D, imgs[l].g[0]));
    RC(gemm(s.xn, Lw.wqkv, s.qkv0, B, NQKV, D, nullptr, nullptr, 0, 1, st));
    TA_LAUNCH(lm_qkv_post_decode_kernel, dim3(Hq + 2 * Hkv, B), dim3(64), 0, st, s.qkv0, Lw.qn_w, Lw.kn_w, w->rope_cos,
              w->rope_sin, pos, slot_dev, s.q, kc, vc, Hq, Hkv, Lmax, w->eps);
    TA_CHECK_LAUNCH();
    TA_LAUNCH(attn_decode_kernel, dim3(Hq, B), dim3(256), smem, st, s.q, kc, vc, kmask, slot_dev, s.ao, Hq, Hkv, Lmax, scale);
    TA_CHECK_LAUNCH();
    if (lora) RC(lora_fwd(s.ao, bq, imgs[l].g[1]));
    RC(gemm(s.ao, Lw.wo, s.x1, B, D, bq, nullptr, s.x, 0, 0, st));
    RC(ta_rmsnorm_fwd(s.x1, Lw.ln_post_w, s.xn, nullptr, s.r, B, D, w->eps, 0, st));
    if (lora) RC(lora_fwd(s.xn, D, imgs[l].g[2]));
    RC(gemm(s.xn, Lw.wgu, s.gu, B, 2 * F, D, nullptr, nullptr, 0, 1, st));
    RC(ta_swiglu_fwd(s.gu, s.act, B, F, st));
    if (lora) RC(lora_fwd(s.act, F, imgs[l].g[3]));
    RC(gemm(s.act, Lw.wd, s.x, B, D, F, nullptr, s.x1, 0, 0, st));
  }
  RC(ta_rmsnorm_fwd(s.x, w->norm_w, s.hn, nullptr, s.r, B, D, w->eps, 0, st));
  RC(gemm(s.hn, w->embed_bf16, logits, B, w->vocab_pad, D, nullptr, nullptr, 0, 0, st));
  return TA_OK;
}

extern "C" int ta_argmax_f32(const float* x, long ld, int n, int rows, long* out, hipStream_t st) {
  if (rows <= 0) return TA_OK;
  if (n <= 0) return TA_ERR_ARG;
  TA_LAUNCH(argmax_rows_kernel, dim3(rows), dim3(256), 0, st, x, ld, n, out);
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
