// ta355 composite ops: the layer loops of the hot path, orchestrated on the host side of the C ABI so a
// binding makes one call per reference nn.Module boundary (see include/ta355.h).  Pure kernel launches on
// the caller's stream: no allocation, no synchronisation, graph-capturable.
#include <cstdlib>
#include "common.h"
#include "internal.h"
#include "../../include/ta355.h"

extern "C" int ta_version(void) { return 4; }

#include <algorithm>
#include "host_util.h"

// ============================================================================ residual-stream dtypes (the numerics contract)
// Which dtype the encoder's residual stream, the LM's forward residual stream (+ tape) and the LM's backward d(x) stream are STORED
// in travels with the weights handle of each call (ta_encoder_weights.res_f32, ta_lm_weights.res_f32 / dx_f32; ABI 4) -- rounds 1-5
// kept it in process-wide state, which two models of different model_dtype in one process raced on.  bf16 (0) is what a bf16-module
// reference keeps (ASRConfig model_dtype="bfloat16"); fp32 (1) is what the training recipe keeps (fp32 modules under bf16 autocast:
// configs/config.yaml:14-18 + configs/training/production.yaml:49).  MFMA operands (bf16) and accumulators (fp32) are the same in both.

// ============================================================================ encoder
namespace {
struct EncWs {
  bf16_t *x0, *x1, *xn, *qkv, *q, *k, *vt, *ao, *hf;
  float* xr;
  size_t bytes;
};
EncWs enc_carve(const ta_encoder_weights* w, int B, int T, void* base) {
  const int H = w->hidden, S = (T - 1) / 2 + 1, Sp = pad64(S);
  const long M = (long)B * S;
  Carver c(base);
  EncWs e;
  e.x0 = c.take<bf16_t>((size_t)B * (T + 2) * w->n_mels);
  e.x1 = c.take<bf16_t>((size_t)B * (T + 2) * H);
  e.xr = (float*)c.take<float>((size_t)M * H);      // residual stream (bf16 by default: the fp32 size is reserved)
  e.xn = c.take<bf16_t>((size_t)M * H);
  e.qkv = c.take<bf16_t>((size_t)M * 3 * H + 128);  // fused path: q|k [M, 2H] then V^T [H, M] + the slack its last tile reads
  e.q = c.take<bf16_t>((size_t)M * H);
  e.k = c.take<bf16_t>((size_t)M * H);
  e.vt = c.take<bf16_t>((size_t)B * H * Sp);
  e.ao = c.take<bf16_t>((size_t)M * H);
  e.hf = c.take<bf16_t>((size_t)M * w->ffn);
  e.bytes = c.total();
  return e;
}
}  // namespace

extern "C" long ta_encoder_workspace_bytes(const ta_encoder_weights* w, int B, int T) {
  return (long)enc_carve(w, B, T, nullptr).bytes;
}

extern "C" int ta_encoder_forward(const ta_encoder_weights* w, const float* feats, int B, int T, const float* frame_keep,
                                  void* out_bf16, float* out_f32, void* ws, long ws_bytes, hipStream_t st) {
  if (B <= 0 || T <= 0) return TA_OK;
  const int H = w->hidden, F = w->ffn, NM = w->n_mels, nh = w->heads;
  if (H % 128 || F % 128 || (3 * NM) % 64 || H / nh != 64 || (!out_bf16 && !out_f32)) return TA_ERR_ARG;
  const int S = (T - 1) / 2 + 1, Sp = pad64(S);
  if (S > w->max_pos) return TA_ERR_ARG;
  const int M = B * S;
  EncWs e = enc_carve(w, B, T, ws);
  if ((long)e.bytes > ws_bytes) return TA_ERR_ARG;
  // conv front end as two row-mapped GEMMs over zero-padded time-major buffers
  RC(ta_feats_to_time_major(feats, e.x0, B, NM, T, st));
  RC(ta_zero_pad_rows(e.x1, B, T, H, st));
  RC(ta_gemm_bf16_nt(e.x0, w->conv1_w, e.x1, B * T, H, 3 * NM, NM, T, (long)(T + 2) * NM, H, T, (long)(T + 2) * H, H,
                     w->conv1_b, nullptr, 1, 1, 1, nullptr, st));
  // Residual stream: bf16, the dtype the reference's encoder runs in (model_dtype bfloat16: every residual add and
  // LayerNorm input is bf16 there).  It halves the bytes of the two residual GEMM epilogues and of the LayerNorm reads
  // per layer -- HBM time that nothing overlaps.  ta_encoder_weights.res_f32 = 1 keeps an fp32 stream instead.
  const bool res_f32 = w->res_f32 != 0;
  const int rb = res_f32 ? 0 : 1;
  auto ln = [&](const float* gw, const float* gb, void* yb, float* yf, const float* rowscale) -> int {
    return rb ? ta_layernorm_bf16(e.xr, gw, gb, yb, yf, rowscale, M, H, w->ln_eps, st)
              : ta_layernorm_f32(e.xr, gw, gb, yb, yf, rowscale, M, H, w->ln_eps, st);
  };
  auto res_gemm = [&](const void* A, const void* Wm, int K, const float* bias) -> int {      // xr += A Wm^T + bias
    if (rb) { ta_gemm_opts o = opts_none(); o.residual_bf16 = e.xr; return gemm_opt(A, Wm, e.xr, M, H, K, bias, nullptr, 0, 1, o, st); }
    return gemm(A, Wm, e.xr, M, H, K, bias, e.xr, 0, 0, st);
  };
  RC(ta_gemm_bf16_nt(e.x1, w->conv2_w, e.xr, M, H, 3 * H, 2L * H, S, (long)(T + 2) * H, H, 0, 0, 0, w->conv2_b, nullptr,
                     1, rb, 1, nullptr, st));
  const float scale = 0.125f;   // head_dim ** -0.5, head_dim = 64
  // Default: q | k | v from ONE GEMM + ta_attention_enc_fwd (needs the derived weight images, head_dim 64).  TA355_ENC_QKV_FUSED=0
  // keeps the three-kernel path (q|k|v GEMM + ta_enc_qkv_post + ta_attention_fwd); read per call: the tests compare both.
  // (Rounds 1-2 had a third form -- rope in the q|k epilogue + V^T = Wv xn^T as a second GEMM; rounds 1-3 a folded-LayerNorm form:
  // both removed in round 5)
  const char* fz = getenv("TA355_ENC_QKV_FUSED");
  const bool fa = !(fz && *fz == '0') && w->rope_il && (H / nh) == 64;
  for (int l = 0; l < w->n_layers; ++l) {
    const ta_enc_layer& L = w->layers[l];
    RC(ln(L.ln1_w, L.ln1_b, e.xn, nullptr, nullptr));
    if (fa && L.wqkv_fa && L.bqkv_fa) {
      // Round 3: q | k | v out of ONE GEMM as a token-major [M, 3H] buffer (rope on the q | k columns, q pre-scaled by
      // head_dim^-0.5 log2 e through the weight image), read in place by the DMA-staged base-2 attention kernel
      // (csrc/attention_enc.hip): no V^T GEMM, no transposed image, any M.
      ta_gemm_opts o = opts_none(); o.rope_tab = w->rope_il; o.rope_rows = S; o.rope_cols = 2 * H;
      RC(gemm_opt(e.xn, L.wqkv_fa, e.qkv, M, 3 * H, H, L.bqkv_fa, nullptr, 2, 1, o, st));
      RC(ta_attention_enc_fwd(e.qkv, e.ao, B, nh, S, st));
      RC(res_gemm(e.ao, L.wo, H, L.bo));
    } else {
      RC(gemm(e.xn, L.wqkv, e.qkv, M, 3 * H, H, L.bqkv, nullptr, 0, 1, st));
      RC(ta_enc_qkv_post(e.qkv, w->rope_cos, w->rope_sin, e.q, e.k, e.vt, B, nh, S, Sp, st));
      RC(ta_attention_fwd(e.q, e.k, e.vt, e.ao, nullptr, nullptr, B, nh, nh, S, Sp, 64, 0, scale, st));
      RC(res_gemm(e.ao, L.wo, H, L.bo));
    }
    RC(ln(L.ln2_w, L.ln2_b, e.xn, nullptr, nullptr));
    RC(gemm(e.xn, L.w1, e.hf, M, F, H, L.b1, nullptr, 1, 1, st));
    RC(res_gemm(e.hf, L.w2, F, L.b2));
  }
  RC(ln(w->norm_w, w->norm_b, out_bf16, out_f32, frame_keep));
  return TA_OK;
}

// ============================================================================ MLP projector
namespace {
struct MlpTape { float *h1, *r1, *h2, *r2; bf16_t* a1; size_t bytes; };
MlpTape mlp_tape(const ta_mlp_weights* w, int B, int S, void* base) {
  const int N = (S - w->k) / w->k + 1;
  const long Mp = (long)B * N;
  Carver c(base);
  MlpTape t;
  t.h1 = c.take<float>((size_t)Mp * w->hidden);
  t.r1 = c.take<float>((size_t)Mp);
  t.a1 = c.take<bf16_t>((size_t)Mp * w->hidden);
  t.h2 = c.take<float>((size_t)Mp * w->llm_dim);
  t.r2 = c.take<float>((size_t)Mp);
  t.bytes = c.total();
  return t;
}
struct MlpBwdWs { float *dh2, *da1, *skws; bf16_t *dh2b, *dh2T, *a1T, *dh1b, *dh1T, *xsT; size_t bytes; };
MlpBwdWs mlp_bwd_ws(const ta_mlp_weights* w, int B, int S, void* base) {
  const int N = (S - w->k) / w->k + 1, Hd = w->hidden, D = w->llm_dim, In = w->k * w->enc_dim;
  const long Mp = (long)B * N; const int Kp = pad64((int)Mp);
  Carver c(base);
  MlpBwdWs s;
  s.dh2 = c.take<float>((size_t)Mp * D);
  s.dh2b = c.take<bf16_t>((size_t)Mp * D);
  s.dh2T = c.take<bf16_t>((size_t)D * Kp);
  s.a1T = c.take<bf16_t>((size_t)Hd * Kp);
  s.da1 = c.take<float>((size_t)Mp * Hd);
  s.dh1b = c.take<bf16_t>((size_t)Mp * Hd);
  s.dh1T = c.take<bf16_t>((size_t)Hd * Kp);
  s.xsT = c.take<bf16_t>((size_t)In * Kp);
  const int s1 = pick_splits(Hd, In, Kp), s2 = pick_splits(D, Hd, Kp);
  const size_t sk1 = (size_t)ta_gemm_splitk_ws_bytes(Hd, In, s1), sk2 = (size_t)ta_gemm_splitk_ws_bytes(D, Hd, s2);
  s.skws = c.take<float>((sk1 > sk2 ? sk1 : sk2) / 4 + 4);
  s.bytes = c.total();
  return s;
}
}  // namespace

extern "C" long ta_mlp_tape_bytes(const ta_mlp_weights* w, int B, int S) { return (long)mlp_tape(w, B, S, nullptr).bytes; }
extern "C" long ta_mlp_bwd_workspace_bytes(const ta_mlp_weights* w, int B, int S) {
  return (long)mlp_bwd_ws(w, B, S, nullptr).bytes;
}

extern "C" int ta_mlp_projector_forward(const ta_mlp_weights* w, const void* x, int B, int S, float* y, void* tape,
                                        hipStream_t st) {
  const int k = w->k, E = w->enc_dim, Hd = w->hidden, D = w->llm_dim, In = k * E;
  const int N = (S - k) / k + 1;
  if (B <= 0 || N <= 0) return TA_OK;
  if (In % 64 || Hd % 64 || D % 4 || Hd % 4) return TA_ERR_ARG;
  const int Mp = B * N;
  MlpTape t = mlp_tape(w, B, S, tape);
  // frame stacking is a row map of the encoder output: row (b,n) starts at b*S*E + n*k*E   (projectors.py:79-87)
  RC(ta_gemm_bf16_nt(x, w->w1, t.h1, Mp, Hd, In, In, N, (long)S * E, Hd, 0, 0, 0, nullptr, nullptr, 0, 0, 1, nullptr, st));
  RC(ta_rmsnorm_fwd(t.h1, w->g1, t.a1, nullptr, t.r1, Mp, Hd, w->eps, 1, st));
  RC(gemm(t.a1, w->w2, t.h2, Mp, D, Hd, nullptr, nullptr, 0, 0, st));
  RC(ta_rmsnorm_fwd(t.h2, w->g2, nullptr, y, t.r2, Mp, D, w->eps, 0, st));
  return TA_OK;
}

extern "C" int ta_mlp_projector_backward(const ta_mlp_weights* w, const void* x, int B, int S, const float* dy,
                                         const void* tape, float* dW1, float* dg1, float* dW2, float* dg2, void* ws,
                                         long ws_bytes, hipStream_t st) {
  const int k = w->k, E = w->enc_dim, Hd = w->hidden, D = w->llm_dim, In = k * E;
  const int N = (S - k) / k + 1;
  if (B <= 0 || N <= 0) return TA_OK;
  const int Mp = B * N, Kp = pad64(Mp);
  MlpTape t = mlp_tape(w, B, S, (void*)tape);
  MlpBwdWs s = mlp_bwd_ws(w, B, S, ws);
  if ((long)s.bytes > ws_bytes) return TA_ERR_ARG;
  if (hipMemsetAsync(dg1, 0, (size_t)Hd * 4, st) != hipSuccess) return TA_ERR_LAUNCH;
  if (hipMemsetAsync(dg2, 0, (size_t)D * 4, st) != hipSuccess) return TA_ERR_LAUNCH;
  // norm_2 backward -> dH2 ; dW2 = dH2^T A1 ; dA1 = dH2 W2
  RC(ta_rmsnorm_bwd(dy, t.h2, t.r2, w->g2, nullptr, nullptr, s.dh2b, dg2, Mp, D, 0, st));
  RC(ta_transpose_to_bf16(s.dh2b, 0, D, 0, 0, s.dh2T, Kp, Mp, D, st));
  RC(ta_transpose_to_bf16(t.a1, 0, Hd, 0, 0, s.a1T, Kp, Mp, Hd, st));
  const int s2 = pick_splits(D, Hd, Kp);
  RC(ta_gemm_bf16_nt(s.dh2T, s.a1T, dW2, D, Hd, Kp, Kp, 0, 0, Hd, 0, 0, 0, nullptr, nullptr, 0, 0, s2, s.skws, st));
  RC(gemm(s.dh2b, w->w2_t, s.da1, Mp, Hd, D, nullptr, nullptr, 0, 0, st));
  // GELU' + norm backward -> dH1 ; dW1 = dH1^T Xs
  RC(ta_rmsnorm_bwd(s.da1, t.h1, t.r1, w->g1, nullptr, nullptr, s.dh1b, dg1, Mp, Hd, 1, st));
  RC(ta_transpose_to_bf16(s.dh1b, 0, Hd, 0, 0, s.dh1T, Kp, Mp, Hd, st));
  RC(ta_transpose_to_bf16(x, 0, In, (long)S * E, N, s.xsT, Kp, Mp, In, st));
  const int s1 = pick_splits(Hd, In, Kp);
  RC(ta_gemm_bf16_nt(s.dh1T, s.xsT, dW1, Hd, In, Kp, Kp, 0, 0, In, 0, 0, 0, nullptr, nullptr, 0, 0, s1, s.skws, st));
  return TA_OK;
}

// ============================================================================ Qwen3 LM
namespace {
// Residual stream of the LM (x_in / x1 of every layer, kept in the tape): bf16, the dtype the reference's LM runs in
// (model_dtype bfloat16).  Halves the bytes of the o / down GEMM epilogues, of the RMSNorm reads (forward and
// backward) and of the tape.  ta_lm_weights.res_f32 = 1 keeps fp32 instead; the gradient stream d(x) follows it unless dx_f32 = 1.
inline bool lm_res_bf16(const ta_lm_weights* w) { return w->res_f32 == 0; }
struct LmLayerTape {
  float *x_in, *r_in, *rq, *rk, *lse, *x1, *r_post;
  bf16_t *qkv0, *q, *k, *v, *qt, *kt, *vt, *ao, *gu;
  // LoRA only: the adapted linears' inputs, the rank-space activations xa = x (s Acat)^T, and the bf16 images
  bf16_t *xn_s, *xn2_s, *act_s, *xa_qkv, *xa_o, *xa_gu, *xa_d;
  LoraImg i_qkv, i_o, i_gu, i_d;
};
struct LmTape {
  LmLayerTape* L;    // host array (static storage below)
  float *x_final, *r_f;
  bf16_t *hn, *dlogits;
  size_t bytes;
};
constexpr int MAX_LM_LAYERS = 64;
struct LmDims { int D, F, nq, nkv, hd, NQKV, Lp; long M; };
LmDims lm_dims(const ta_lm_weights* w, int B, int L) {
  LmDims d;
  d.D = w->hidden; d.F = w->ffn; d.nq = w->heads; d.nkv = w->kv_heads; d.hd = w->head_dim;
  d.NQKV = (d.nq + 2 * d.nkv) * d.hd; d.Lp = pad64(L); d.M = (long)B * L;
  return d;
}
// short causal sequences: QK-norm + RoPE + head split ride in the attention kernel's staging (ta_attention_fwd_qkv)
inline bool lm_attn_fused(const LmDims& d, int L) { return d.hd == 128 && L <= 192 && (d.nq / d.nkv) * ((L + 31) / 32) <= 12; }
LmTape lm_tape(const ta_lm_weights* w, int B, int L, int n_lab, void* base, LmLayerTape* store) {
  const LmDims d = lm_dims(w, B, L);
  Carver c(base);
  LmTape t; t.L = store;
  for (int l = 0; l < w->n_layers; ++l) {
    LmLayerTape& p = store[l];
    p.x_in = c.take<float>((size_t)d.M * d.D);
    p.r_in = c.take<float>((size_t)d.M);
    p.qkv0 = c.take<bf16_t>((size_t)d.M * d.NQKV);
    p.rq = c.take<float>((size_t)d.M * d.nq);
    p.rk = c.take<float>((size_t)d.M * d.nkv);
    p.q = c.take<bf16_t>((size_t)d.M * d.nq * d.hd);
    p.k = c.take<bf16_t>((size_t)d.M * d.nkv * d.hd);
    p.v = c.take<bf16_t>((size_t)d.M * d.nkv * d.hd);
    p.qt = nullptr; p.kt = nullptr;                 // no Q^T / K^T images: the attention backward reads them transposed out of its row tiles
    p.vt = c.take<bf16_t>((size_t)B * d.nkv * d.hd * d.Lp);
    p.ao = c.take<bf16_t>((size_t)d.M * d.nq * d.hd);
    p.lse = c.take<float>((size_t)B * d.nq * L);
    p.x1 = c.take<float>((size_t)d.M * d.D);
    p.r_post = c.take<float>((size_t)d.M);
    p.gu = c.take<bf16_t>((size_t)d.M * 2 * d.F);
    if (w->lora_rank > 0 || w->train_base) {         // inputs of the linears: adapter / weight gradients need them
      p.xn_s = c.take<bf16_t>((size_t)d.M * d.D); p.xn2_s = c.take<bf16_t>((size_t)d.M * d.D);
      p.act_s = c.take<bf16_t>((size_t)d.M * d.F);
    }
    if (w->lora_rank > 0) {
      p.xa_qkv = c.take<bf16_t>((size_t)d.M * 64); p.xa_o = c.take<bf16_t>((size_t)d.M * 64);
      p.xa_gu = c.take<bf16_t>((size_t)d.M * 64); p.xa_d = c.take<bf16_t>((size_t)d.M * 64);
      auto img = [&](LoraImg& g, int in, int N) {
        g.a = c.take<bf16_t>((size_t)64 * in); g.at = c.take<bf16_t>((size_t)64 * in);
        g.b = c.take<bf16_t>((size_t)64 * N); g.bt = c.take<bf16_t>((size_t)64 * N);
      };
      img(p.i_qkv, d.D, d.NQKV); img(p.i_o, d.nq * d.hd, d.D); img(p.i_gu, d.D, 2 * d.F); img(p.i_d, d.F, d.D);
    }
  }
  t.x_final = c.take<float>((size_t)d.M * d.D);
  t.r_f = c.take<float>((size_t)d.M);
  t.hn = c.take<bf16_t>((size_t)d.M * d.D);
  t.dlogits = c.take<bf16_t>((size_t)(n_lab > 0 ? n_lab : 1) * w->vocab_pad);
  t.bytes = c.total();
  return t;
}
struct LmWs {
  bf16_t *xn, *act, *hl, *dxb, *dact, *dgu, *dao, *dot, *dq, *dk, *dv, *dqkv, *dyB;
  float* lora_part;            // round 4: per-chunk partial adapter gradients of every layer (two-level reduction, no atomics)
  long lp_layer, lp_off[8], lp_size[8]; int lp_chunks[8];   // kinds in ta_lm_lora_grads order: la_qkv, lb_qkv, la_o, lb_o, la_gu, lb_gu, la_d, lb_d
  float *logits, *dhl, *dhn, *dxa, *dxb32, *dxn, *delta, *skws;
  // trainable LM: transposed bf16 images of one dW product's operands ([rows, Kp], Kp = tokens rounded up to 64) and
  // the split-K slabs of the dW GEMMs
  bf16_t *tA, *tB;
  float* wsk;
  size_t bytes;
};
// split-K of a weight-gradient GEMM  dW[N_out, K_in] = dY^T X  (contraction over the tokens)
inline int wgrad_splits(int n_out, int k_in, int kp) { return pick_splits(n_out, k_in, kp); }
LmWs lm_ws(const ta_lm_weights* w, int B, int L, int n_lab, void* base) {
  const LmDims d = lm_dims(w, B, L);
  const int nl = n_lab > 0 ? n_lab : 1;
  Carver c(base);
  LmWs s;
  s.xn = c.take<bf16_t>((size_t)d.M * d.D);
  s.act = c.take<bf16_t>((size_t)d.M * d.F);
  s.hl = c.take<bf16_t>((size_t)nl * d.D);
  s.logits = c.take<float>((size_t)nl * w->vocab_pad);
  s.dhl = c.take<float>((size_t)nl * d.D);
  s.dhn = c.take<float>((size_t)d.M * d.D);
  s.dxa = c.take<float>((size_t)d.M * d.D);
  s.dxb32 = c.take<float>((size_t)d.M * d.D);
  s.dxn = c.take<float>((size_t)d.M * d.D);
  s.dxb = c.take<bf16_t>((size_t)d.M * d.D);
  s.dact = c.take<bf16_t>((size_t)d.M * d.F);
  s.dgu = c.take<bf16_t>((size_t)d.M * 2 * d.F);
  s.dao = c.take<bf16_t>((size_t)d.M * d.nq * d.hd);
  s.dot = nullptr;                                  // likewise no dO^T image
  s.delta = c.take<float>((size_t)B * d.nq * L);
  s.dq = c.take<bf16_t>((size_t)d.M * d.nq * d.hd);
  s.dk = c.take<bf16_t>((size_t)d.M * d.nkv * d.hd);
  s.dv = c.take<bf16_t>((size_t)d.M * d.nkv * d.hd);
  s.dqkv = c.take<bf16_t>((size_t)d.M * d.NQKV);
  s.dyB = c.take<bf16_t>((size_t)d.M * 64);
  s.lora_part = nullptr; s.lp_layer = 0;
  if (w->lora_rank > 0) {
    const int r = w->lora_rank, bq = d.nq * d.hd;
    const int members[4] = {3, 1, 2, 1}, Ng[4] = {d.NQKV, d.D, 2 * d.F, d.D}, ing[4] = {d.D, bq, d.D, d.F};
    long off = 0;
    for (int g = 0; g < 4; ++g) {
      const int chunks = ta_cdiv((int)d.M, ta_i_lora_tn2_rows((int)d.M, Ng[g], ing[g]));
      s.lp_size[2 * g] = (long)members[g] * r * ing[g]; s.lp_size[2 * g + 1] = (long)Ng[g] * r;      // la_g, lb_g
      for (int h = 0; h < 2; ++h) { s.lp_chunks[2 * g + h] = chunks; s.lp_off[2 * g + h] = off; off += s.lp_size[2 * g + h] * chunks; }
    }
    s.lp_layer = off;
    s.lora_part = c.take<float>((size_t)off * w->n_layers);
  }
  const int sp = pick_splits(nl, d.D, w->vocab_pad);
  s.skws = c.take<float>((size_t)ta_gemm_splitk_ws_bytes(nl, d.D, sp) / 4 + 4);
  s.tA = s.tB = nullptr; s.wsk = nullptr;
  if (w->train_base) {
    const int Kp = pad64((int)d.M), Kl = pad64(nl), bq = d.nq * d.hd;
    const size_t a_rows = (size_t)std::max(std::max(2 * d.F, d.NQKV), d.D), b_rows = (size_t)std::max(std::max(d.F, bq), d.D);
    s.tA = c.take<bf16_t>(std::max(a_rows * Kp, (size_t)w->vocab_pad * Kl));     // also dlogits^T [vocab_pad, Kl]
    s.tB = c.take<bf16_t>(std::max(b_rows * Kp, (size_t)d.D * Kl));
    const int shp[4][2] = {{d.NQKV, d.D}, {d.D, bq}, {2 * d.F, d.D}, {d.D, d.F}};
    size_t mx = 0;
    for (auto& q : shp) {
      mx = std::max(mx, (size_t)ta_gemm_splitk_ws_bytes(q[0], q[1], wgrad_splits(q[0], q[1], Kp)));
      mx = std::max(mx, (size_t)ta_gemm_bf16_tn_ws_bytes((int)d.M, q[0], q[1]));
    }
    s.wsk = c.take<float>(mx / 4 + 4);
  }
  s.bytes = c.total();
  return s;
}
}  // namespace

extern "C" long ta_lm_tape_bytes(const ta_lm_weights* w, int B, int L, int n_label_rows) {
  LmLayerTape store[MAX_LM_LAYERS];
  if (w->n_layers > MAX_LM_LAYERS) return -1;
  return (long)lm_tape(w, B, L, n_label_rows, nullptr, store).bytes;
}
extern "C" long ta_lm_workspace_bytes(const ta_lm_weights* w, int B, int L, int n_label_rows) {
  return (long)lm_ws(w, B, L, n_label_rows, nullptr).bytes;
}

// The decoder layers of Qwen3 (TF:models/qwen3/modeling_qwen3.py:283-324) over B*L token rows.  Training keeps one
// tape entry per layer; inference (`alias`) reuses entry 0 for every layer.  `ext`: adapter images held outside the
// tape (generation keeps them across decode steps); kcache/vcache: see ta_lm_prefill.
static int lm_layers_forward(const ta_lm_weights* w, const LmDims& d, int B, int L, const int* kmask, const int* pos,
                             LmLayerTape* store, bool alias, float* x_final, LmWs& s, const ta_i_lora_layer_imgs* ext,
                             bf16_t* kcache, bf16_t* vcache, int Lmax, hipStream_t st) {
  const int M = (int)d.M;
  const float scale = 1.0f / sqrtf((float)d.hd);
  const bool lora = w->lora_rank > 0;
  const int r = w->lora_rank;
  if (lora && (r < 1 || 3 * r > 64)) return TA_ERR_ARG;   // one 64-wide K tile holds the 3 members of the q|k|v group
  const int lgm = lora ? (w->lora_groups ? w->lora_groups : 15) : 0;      // groups that carry an adapter (1 qkv, 2 o, 4 gate|up, 8 down)
  // y = x W^T + xa Bext^T with xa = x (s Acat)^T: one skinny GEMM for xa, then the frozen GEMM runs one extra K-tile
  ta_gemm_opts kx = opts_none();                     // K extension of the NEXT frozen GEMM (consumed and cleared by it)
  auto lora_fwd = [&](const bf16_t* x, int in, const LoraImg& g, bf16_t* xa, int members) -> int {
    RC(ta_i_lora_skinny_nt(x, in, g.a, xa, M, members * r, st));
    kx = opts_kext(xa, g.b);
    return TA_OK;
  };
  auto take_ext = [&]() { const ta_gemm_opts o = kx; kx = opts_none(); return o; };
  if (lora) {
    // bf16 images of every layer's adapters (kept in the tape / the caller's image buffer for backward and decoding).
    // The masters of consecutive layers are usually one tensor [L, ...] and the images are carved with a constant
    // per-layer stride: then each of the 8 image kinds is ONE launch over all layers instead of one per layer.
    const int NB = 1 << 30, bq = d.nq * d.hd, bk = bq + d.nkv * d.hd, NL = w->n_layers;
    auto imgs_of = [&](int l) {
      LmLayerTape q = store[(alias || ext) ? 0 : l];
      if (ext) { q.i_qkv = ext[l].g[0]; q.i_o = ext[l].g[1]; q.i_gu = ext[l].g[2]; q.i_d = ext[l].g[3]; }
      return q;
    };
    const LmLayerTape q0 = imgs_of(0), q1 = imgs_of(NL > 1 ? 1 : 0);
    const ta_lm_layer &L0 = w->layers[0], &L1 = w->layers[NL > 1 ? 1 : 0];
    bool strided = NL > 1 && (ext || !alias);
    const long is_ = strided ? (long)(q1.i_qkv.a - q0.i_qkv.a) : 0;        // image stride (bf16 elements), same for every kind
    struct Kind { const float* m0; const float* m1; bf16_t* o0; bf16_t* o1; bf16_t* t0; bf16_t* t1; };
    const Kind kinds[8] = {{L0.la_qkv, L1.la_qkv, q0.i_qkv.a, q1.i_qkv.a, q0.i_qkv.at, q1.i_qkv.at}, {L0.lb_qkv, L1.lb_qkv, q0.i_qkv.b, q1.i_qkv.b, q0.i_qkv.bt, q1.i_qkv.bt},
                           {L0.la_o, L1.la_o, q0.i_o.a, q1.i_o.a, q0.i_o.at, q1.i_o.at}, {L0.lb_o, L1.lb_o, q0.i_o.b, q1.i_o.b, q0.i_o.bt, q1.i_o.bt},
                           {L0.la_gu, L1.la_gu, q0.i_gu.a, q1.i_gu.a, q0.i_gu.at, q1.i_gu.at}, {L0.lb_gu, L1.lb_gu, q0.i_gu.b, q1.i_gu.b, q0.i_gu.bt, q1.i_gu.bt},
                           {L0.la_d, L1.la_d, q0.i_d.a, q1.i_d.a, q0.i_d.at, q1.i_d.at}, {L0.lb_d, L1.lb_d, q0.i_d.b, q1.i_d.b, q0.i_d.bt, q1.i_d.bt}};
    long ms[8];
    for (int k = 0; k < 8 && strided; ++k) {
      ms[k] = (long)(kinds[k].m1 - kinds[k].m0);
      if ((long)(kinds[k].o1 - kinds[k].o0) != is_ || (long)(kinds[k].t1 - kinds[k].t0) != is_) strided = false;
    }
    for (int l = 2; l < NL && strided; ++l) {
      const ta_lm_layer& Ll = w->layers[l];
      const LmLayerTape ql = imgs_of(l);
      const float* mm[8] = {Ll.la_qkv, Ll.lb_qkv, Ll.la_o, Ll.lb_o, Ll.la_gu, Ll.lb_gu, Ll.la_d, Ll.lb_d};
      const bf16_t* oo[8] = {ql.i_qkv.a, ql.i_qkv.b, ql.i_o.a, ql.i_o.b, ql.i_gu.a, ql.i_gu.b, ql.i_d.a, ql.i_d.b};
      for (int k = 0; k < 8; ++k)
        if (mm[k] != kinds[k].m0 + l * ms[k] || oo[k] != kinds[k].o0 + l * is_) strided = false;
    }
    const int reps = strided ? 1 : NL, per = strided ? NL : 1;
    for (int l = 0; l < reps; ++l) {
      const ta_lm_layer& Ll = w->layers[l];
      const LmLayerTape ql = imgs_of(l);
      if (lgm & 1) RC(ta_i_lora_pack_a(Ll.la_qkv, w->lora_scale, ql.i_qkv.a, ql.i_qkv.at, 3 * r, d.D, per, strided ? ms[0] : 0, is_, st));
      if (lgm & 1) RC(ta_i_lora_pack_b(Ll.lb_qkv, ql.i_qkv.b, ql.i_qkv.bt, d.NQKV, r, bq, bk, per, strided ? ms[1] : 0, is_, st));
      if (lgm & 2) RC(ta_i_lora_pack_a(Ll.la_o, w->lora_scale, ql.i_o.a, ql.i_o.at, r, bq, per, strided ? ms[2] : 0, is_, st));
      if (lgm & 2) RC(ta_i_lora_pack_b(Ll.lb_o, ql.i_o.b, ql.i_o.bt, d.D, r, NB, NB, per, strided ? ms[3] : 0, is_, st));
      if (lgm & 4) RC(ta_i_lora_pack_a(Ll.la_gu, w->lora_scale, ql.i_gu.a, ql.i_gu.at, 2 * r, d.D, per, strided ? ms[4] : 0, is_, st));
      if (lgm & 4) RC(ta_i_lora_pack_b(Ll.lb_gu, ql.i_gu.b, ql.i_gu.bt, 2 * d.F, r, d.F, NB, per, strided ? ms[5] : 0, is_, st));
      if (lgm & 8) RC(ta_i_lora_pack_a(Ll.la_d, w->lora_scale, ql.i_d.a, ql.i_d.at, r, d.F, per, strided ? ms[6] : 0, is_, st));
      if (lgm & 8) RC(ta_i_lora_pack_b(Ll.lb_d, ql.i_d.b, ql.i_d.bt, d.D, r, NB, NB, per, strided ? ms[7] : 0, is_, st));
    }
  }
  for (int l = 0; l < w->n_layers; ++l) {
    const ta_lm_layer& Lw = w->layers[l];
    LmLayerTape p = store[alias ? 0 : l];
    if (ext) { p.i_qkv = ext[l].g[0]; p.i_o = ext[l].g[1]; p.i_gu = ext[l].g[2]; p.i_d = ext[l].g[3]; }
    float* x_next = (l + 1 < w->n_layers) ? store[alias ? 0 : l + 1].x_in : x_final;
    const bool keep = lora || (w->train_base && !alias);
    bf16_t* xn = keep ? p.xn_s : s.xn;
    const bool rb = lm_res_bf16(w);
    auto norm = [&](const float* x, const float* gw, bf16_t* y, float* r) -> int {
      return rb ? ta_rmsnorm_fwd_bf16(x, gw, y, nullptr, r, M, d.D, w->eps, st) : ta_rmsnorm_fwd(x, gw, y, nullptr, r, M, d.D, w->eps, 0, st);
    };
    auto res_gemm = [&](const void* A, const void* Wm, float* out, int K, const float* res) -> int {   // out = res + A Wm^T
      ta_gemm_opts o = take_ext();
      if (rb) { o.residual_bf16 = res; return gemm_opt(A, Wm, out, M, d.D, K, nullptr, nullptr, 0, 1, o, st); }
      return gemm_opt(A, Wm, out, M, d.D, K, nullptr, res, 0, 0, o, st);
    };
    RC(norm(p.x_in, Lw.ln_in_w, xn, p.r_in));
    if (lgm & 1) RC(lora_fwd(xn, d.D, p.i_qkv, p.xa_qkv, 3));
    RC(gemm_opt(xn, Lw.wqkv, p.qkv0, M, d.NQKV, d.D, nullptr, nullptr, 0, 1, take_ext(), st));
    // short causal sequences: QK-norm + RoPE + head split ride in the attention kernel's staging; longer ones take two kernels
    const bool fused_fwd = lm_attn_fused(d, L);
    if (fused_fwd)
      // the head-major copy of V exists for the KV cache (prefill) and for the un-fused backward (trainable q_norm / k_norm); the fused
      // backward reads V in place from qkv0, which the tape keeps anyway
      RC(ta_attention_fwd_qkv(p.qkv0, Lw.qn_w, Lw.kn_w, w->rope_cos, w->rope_sin, pos, p.q, p.k, (kcache || w->train_base) ? p.v : nullptr, p.rq, p.rk, p.ao, p.lse, kmask,
                              B, d.nq, d.nkv, L, scale, w->eps, st));
    else
      RC(ta_lm_qkv_post_fwd(p.qkv0, Lw.qn_w, Lw.kn_w, w->rope_cos, w->rope_sin, pos, p.q, p.k, p.v, p.qt, p.kt, p.vt, p.rq,
                            p.rk, B, d.nq, d.nkv, L, d.Lp, w->eps, st));
    if (kcache) {   // greedy decoding: keys / values of the prompt go to the cache [layer, B, Hkv, Lmax, hd]
      const size_t row = (size_t)L * d.hd * 2, pitch = (size_t)Lmax * d.hd * 2, rows = (size_t)B * d.nkv;
      if (hipMemcpy2DAsync(kcache + (size_t)l * rows * Lmax * d.hd, pitch, p.k, row, row, rows, hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemcpy2DAsync(vcache + (size_t)l * rows * Lmax * d.hd, pitch, p.v, row, row, rows, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return TA_ERR_LAUNCH;
    }
    if (!fused_fwd) RC(ta_attention_fwd(p.q, p.k, p.vt, p.ao, p.lse, kmask, B, d.nq, d.nkv, L, d.Lp, d.hd, 1, scale, st));
    if (lgm & 2) RC(lora_fwd(p.ao, d.nq * d.hd, p.i_o, p.xa_o, 1));
    RC(res_gemm(p.ao, Lw.wo, p.x1, d.nq * d.hd, p.x_in));
    bf16_t* xn2 = keep ? p.xn2_s : s.xn;
    bf16_t* act = keep ? p.act_s : s.act;
    RC(norm(p.x1, Lw.ln_post_w, xn2, p.r_post));
    if (lgm & 4) RC(lora_fwd(xn2, d.D, p.i_gu, p.xa_gu, 2));
    RC(gemm_opt(xn2, Lw.wgu, p.gu, M, 2 * d.F, d.D, nullptr, nullptr, 0, 1, take_ext(), st));
    RC(ta_swiglu_fwd(p.gu, act, M, d.F, st));
    if (lgm & 8) RC(lora_fwd(act, d.F, p.i_d, p.xa_d, 1));
    RC(res_gemm(act, Lw.wd, x_next, d.F, p.x1));
  }
  return TA_OK;
}

extern "C" int ta_lm_forward_loss(const ta_lm_weights* w, const long* ids, const int* src_row, const float* audio,
                                  const int* kmask, const int* pos, int B, int L, const int* label_rows,
                                  const long* label_targets, int n_lab, float loss_scale, float* loss, float* nll_rows,
                                  void* logits_out, void* tape, void* ws, long ws_bytes, hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  if (w->n_layers > MAX_LM_LAYERS || w->head_dim != 128 || w->hidden % 128 || w->ffn % 64 || w->vocab_pad % 128 ||
      w->vocab > w->vocab_pad || L > w->max_pos)
    return TA_ERR_ARG;
  const LmDims d = lm_dims(w, B, L);
  const int M = (int)d.M;
  LmLayerTape store[MAX_LM_LAYERS];
  LmTape t = lm_tape(w, B, L, n_lab, tape, store);
  LmWs s = lm_ws(w, B, L, n_lab, ws);
  if ((long)s.bytes > ws_bytes) return TA_ERR_ARG;
  float* x = store[0].x_in;
  // inputs_embeds = embed_tokens(ids) with the <audio> rows replaced by projector rows (asr_modeling.py:498,511-515)
  if (lm_res_bf16(w)) RC(ta_embed_scatter(ids, src_row, w->embed_f32, audio, nullptr, x, M, d.D, w->vocab, st));
  else RC(ta_embed_scatter(ids, src_row, w->embed_f32, audio, x, nullptr, M, d.D, w->vocab, st));
  RC(lm_layers_forward(w, d, B, L, kmask, pos, store, false, t.x_final, s, nullptr, nullptr, nullptr, 0, st));
  if (lm_res_bf16(w)) RC(ta_rmsnorm_fwd_bf16(t.x_final, w->norm_w, t.hn, nullptr, t.r_f, M, d.D, w->eps, st));
  else RC(ta_rmsnorm_fwd(t.x_final, w->norm_w, t.hn, nullptr, t.r_f, M, d.D, w->eps, 0, st));
  if (logits_out)   // the reference's outputs.logits (bf16 under autocast), all positions
    RC(gemm(t.hn, w->embed_bf16, logits_out, M, w->vocab_pad, d.D, nullptr, nullptr, 0, 1, st));
  if (n_lab > 0) {
    // loss (and dlogits for backward) only over the positions that carry a label: identical value, the
    // ignored rows contribute exactly zero to both loss and gradient.
    RC(ta_gather_rows_bf16(t.hn, label_rows, s.hl, n_lab, d.D, st));
    // The labelled logits are bf16 -- what the reference's bf16 lm_head produces before `logits.float()`
    // (TF:loss/loss_utils.py:55) -- which halves the bytes of the head epilogue and of the CE pass (0.25 ms per step).
    RC(gemm(s.hl, w->embed_bf16, s.logits, n_lab, w->vocab_pad, d.D, nullptr, nullptr, 0, 1, st));
    RC(ta_cross_entropy(s.logits, 1, w->vocab_pad, nullptr, label_targets, n_lab, w->vocab, loss_scale, nll_rows, loss,
                        t.dlogits, w->vocab_pad, st));
  }
  return TA_OK;
}

// ---------------------------------------------------------------------------- greedy decoding: prompt pass
// (tiny_audio/asr_modeling.py:562-646 -> HF greedy search with a KV cache.)
namespace {
struct PrefillWs { ta_lm_weights w1; LmLayerTape layer[1]; LmTape t; LmWs s; size_t tape_bytes, bytes; };
PrefillWs prefill_ws(const ta_lm_weights* w, int B, int L, void* base) {
  PrefillWs p;
  p.w1 = *w; p.w1.n_layers = 1;                                   // one layer's worth of activations, reused by every layer
  p.t = lm_tape(&p.w1, B, L, 1, base, p.layer);
  p.tape_bytes = p.t.bytes;
  p.s = lm_ws(w, B, L, B, base ? (char*)base + p.tape_bytes : nullptr);   // hl / logits sized for B "label" rows
  p.bytes = p.tape_bytes + p.s.bytes;
  return p;
}
}  // namespace
extern "C" long ta_lm_prefill_workspace_bytes(const ta_lm_weights* w, int B, int L) {
  return (long)prefill_ws(w, B, L, nullptr).bytes;
}
extern "C" long ta_lm_lora_image_bytes(const ta_lm_weights* w) {
  return w->lora_rank > 0 ? (long)lora_imgs_carve(w, nullptr, nullptr) : 0;
}
extern "C" int ta_lm_prefill(const ta_lm_weights* w, const long* ids, const int* src_row, const float* audio,
                             const int* kmask, const int* pos, int B, int L, void* kcache, void* vcache, int Lmax,
                             const int* last_rows, float* logits, void* lora_img, void* ws, long ws_bytes,
                             hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  if (w->n_layers > MAX_LM_LAYERS || w->head_dim != 128 || w->hidden % 128 || w->ffn % 64 || w->vocab_pad % 128 ||
      w->vocab > w->vocab_pad || L > w->max_pos || Lmax < L || !kcache || !vcache || !last_rows || !logits)
    return TA_ERR_ARG;
  if (w->lora_rank > 0 && !lora_img) return TA_ERR_ARG;
  const LmDims d = lm_dims(w, B, L);
  const int M = (int)d.M;
  PrefillWs p = prefill_ws(w, B, L, ws);
  if ((long)p.bytes > ws_bytes) return TA_ERR_ARG;
  ta_i_lora_layer_imgs imgs[MAX_LM_LAYERS];
  if (w->lora_rank > 0) lora_imgs_carve(w, lora_img, imgs);
  if (lm_res_bf16(w)) RC(ta_embed_scatter(ids, src_row, w->embed_f32, audio, nullptr, p.layer[0].x_in, M, d.D, w->vocab, st));
  else RC(ta_embed_scatter(ids, src_row, w->embed_f32, audio, p.layer[0].x_in, nullptr, M, d.D, w->vocab, st));
  RC(lm_layers_forward(w, d, B, L, kmask, pos, p.layer, true, p.t.x_final, p.s, w->lora_rank > 0 ? imgs : nullptr,
                       (bf16_t*)kcache, (bf16_t*)vcache, Lmax, st));
  if (lm_res_bf16(w)) RC(ta_rmsnorm_fwd_bf16(p.t.x_final, w->norm_w, p.t.hn, nullptr, p.t.r_f, M, d.D, w->eps, st));
  else RC(ta_rmsnorm_fwd(p.t.x_final, w->norm_w, p.t.hn, nullptr, p.t.r_f, M, d.D, w->eps, 0, st));
  RC(ta_gather_rows_bf16(p.t.hn, last_rows, p.s.hl, B, d.D, st));
  RC(gemm(p.s.hl, w->embed_bf16, logits, B, w->vocab_pad, d.D, nullptr, nullptr, 0, 0, st));
  return TA_OK;
}

extern "C" int ta_lm_backward(const ta_lm_weights* w, const int* src_row, const int* kmask, const int* pos, int B, int L,
                              const int* label_rows, int n_lab, float* d_audio, long n_audio_rows, float* d_embeds,
                              const ta_lm_lora_grads* lora_grads, const ta_lm_wgrads* wg, const long* ids, const void* tape,
                              void* ws, long ws_bytes, hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  if (w->n_layers > MAX_LM_LAYERS) return TA_ERR_ARG;
  const bool lora = w->lora_rank > 0;
  if (lora && !lora_grads) return TA_ERR_ARG;
  if (wg && (!w->train_base || lora || !wg->layers || (wg->dembed && !ids))) return TA_ERR_ARG;
  const LmDims d = lm_dims(w, B, L);
  const int M = (int)d.M;
  LmLayerTape store[MAX_LM_LAYERS];
  LmTape t = lm_tape(w, B, L, n_lab, (void*)tape, store);
  LmWs s = lm_ws(w, B, L, n_lab, ws);
  if ((long)s.bytes > ws_bytes) return TA_ERR_ARG;
  const float scale = 1.0f / sqrtf((float)d.hd);
  if (d_audio && hipMemsetAsync(d_audio, 0, (size_t)n_audio_rows * d.D * 4, st) != hipSuccess) return TA_ERR_LAUNCH;
  const int r = w->lora_rank;
  const int lgm = lora ? (w->lora_groups ? w->lora_groups : 15) : 0;
  const int bq = d.nq * d.hd, bk = bq + d.nkv * d.hd;            // member row boundaries inside the fused qkv group
  auto zero_lora = [&](const ta_lm_lora_grads& g) -> bool {
    auto z = [&](float* q, size_t n) { return hipMemsetAsync(q, 0, n * 4, st) == hipSuccess; };
    return z(g.dla_qkv, (size_t)3 * r * d.D) && z(g.dlb_qkv, (size_t)d.NQKV * r) && z(g.dla_o, (size_t)r * bq) &&
           z(g.dlb_o, (size_t)d.D * r) && z(g.dla_gu, (size_t)2 * r * d.D) && z(g.dlb_gu, (size_t)2 * d.F * r) &&
           z(g.dla_d, (size_t)r * d.F) && z(g.dlb_d, (size_t)d.D * r);
  };
  bool lora_parts = false;
  if (lora) {
    // The per-layer gradients are usually slices of 8 stacked tensors [n_layers, ...]: then 8 memsets clear everything
    // (224 tiny fills cost 0.6 ms per step at 28 layers); otherwise layer by layer.
    const int NL = w->n_layers;
    const size_t sz[8] = {(size_t)3 * r * d.D, (size_t)d.NQKV * r, (size_t)r * bq, (size_t)d.D * r, (size_t)2 * r * d.D,
                          (size_t)2 * d.F * r, (size_t)r * d.F, (size_t)d.D * r};
    auto fld = [&](const ta_lm_lora_grads& g, int k) -> float* {
      float* const f[8] = {g.dla_qkv, g.dlb_qkv, g.dla_o, g.dlb_o, g.dla_gu, g.dlb_gu, g.dla_d, g.dlb_d};
      return f[k];
    };
    bool stacked = true;
    for (int k = 0; k < 8 && stacked; ++k)
      for (int l = 1; l < NL; ++l)
        if (fld(lora_grads[l], k) != fld(lora_grads[0], k) + (size_t)l * sz[k]) { stacked = false; break; }
    // round 4: with stacked gradient tensors the adapter-gradient kernels store per-chunk partial sums and ONE kernel at the end adds
    // them in a fixed order -- deterministic, and no memset of the gradients; otherwise (per-layer tensors, r % 4 != 0): float atomics
    lora_parts = stacked && s.lora_part && (r % 4 == 0);
    for (int k = 0; k < 8 && lora_parts; ++k) if (sz[k] != (size_t)s.lp_size[k]) lora_parts = false;
    if (stacked) {
      if (!lora_parts || lgm != 15)
        for (int k = 0; k < 8; ++k)
          if (hipMemsetAsync(fld(lora_grads[0], k), 0, sz[k] * NL * 4, st) != hipSuccess) return TA_ERR_LAUNCH;
    } else {
      for (int l = 0; l < NL; ++l) if (!zero_lora(lora_grads[l])) return TA_ERR_LAUNCH;
    }
  }
  if (n_lab <= 0) {
    if (d_embeds && hipMemsetAsync(d_embeds, 0, (size_t)M * d.D * 4, st) != hipSuccess) return TA_ERR_LAUNCH;
    return TA_OK;
  }
  // Adapter backward for one group, given dy [M, N] (bf16), the group's input x [M, in] and xa [M, 64]:
  //   dyB = dy Bext            (rank space, [M, 64])       dBext = dy^T xa   (block-masked)
  //   dx  = dy W + dyB (sAcat) (K extension of the dX GEMM) dAcat = s (dyB)^T x
  // `arm` only prepares the K extension; the caller then issues the frozen dX GEMM.
  ta_gemm_opts kx = opts_none();                     // K extension of the NEXT frozen dX GEMM
  auto lora_bwd = [&](const bf16_t* dy, int N, const bf16_t* x, int in, const bf16_t* xa, const LoraImg& g, float* dla,
                      float* dlb, int members, int b0, int b1, int layer, int grp) -> int {
    RC(ta_i_lora_skinny_nt(dy, N, g.bt, s.dyB, M, members * r, st));
    // dB = dy^T xa and dA = s (dy B)^T x in ONE launch (round 3); per-chunk partial sums when the gradients are stacked (round 4)
    float* pb = lora_parts ? s.lora_part + (long)layer * s.lp_layer + s.lp_off[2 * grp + 1] : nullptr;
    float* pa = lora_parts ? s.lora_part + (long)layer * s.lp_layer + s.lp_off[2 * grp] : nullptr;
    RC(ta_i_lora_skinny_tn2(dy, N, xa, members * r, dlb, r, 1, 1.0f, r, b0, b1, x, in, s.dyB, members * r, dla, 1, in, w->lora_scale,
                            0, 0, 0, M, pb, s.lp_size[2 * grp + 1], pa, s.lp_size[2 * grp], st));
    kx = opts_kext(s.dyB, g.at);
    return TA_OK;
  };
  auto take_ext = [&]() { const ta_gemm_opts o = kx; kx = opts_none(); return o; };
  // d hidden (labelled rows) = dlogits x E   (split-K over the vocabulary), scattered back to all positions
  const int sp = pick_splits(n_lab, d.D, w->vocab_pad);
  RC(ta_gemm_bf16_nt(t.dlogits, w->embed_t_bf16, s.dhl, n_lab, d.D, w->vocab_pad, w->vocab_pad, 0, 0, d.D, 0, 0, 0, nullptr,
                     nullptr, 0, 0, sp, s.skws, st));
  if (hipMemsetAsync(s.dhn, 0, (size_t)M * d.D * 4, st) != hipSuccess) return TA_ERR_LAUNCH;
  RC(ta_scatter_rows_f32(s.dhl, label_rows, s.dhn, n_lab, d.D, st));
  // ---- weight gradients (full decoder fine-tuning): dW[N_out, K_in] += dY^T X as an NT GEMM over transposed bf16 images
  const int Kp = pad64(M);
  // the TN kernel (csrc/gemm_tn.hip) where it measured faster at M = 6144 (q|k|v 85 vs 93 us, o 58 vs 69), two transposes + the NT GEMM
  // elsewhere (gate|up 151 vs 127, down 88 vs 78)
  auto wgrad = [&](const bf16_t* dy, int n_out, const bf16_t* x, int k_in, float* dW) -> int {
    if (!dW) return TA_OK;
    if (n_out <= 4096 && k_in <= 2048)
      return ta_gemm_bf16_tn(dy, x, dW, M, n_out, k_in, 1, s.wsk, ta_gemm_bf16_tn_ws_bytes(M, n_out, k_in), st);
    RC(ta_transpose_to_bf16(dy, 0, n_out, 0, 0, s.tA, Kp, M, n_out, st));
    RC(ta_transpose_to_bf16(x, 0, k_in, 0, 0, s.tB, Kp, M, k_in, st));
    return ta_gemm_bf16_nt(s.tA, s.tB, dW, n_out, k_in, Kp, Kp, 0, 0, k_in, 0, 0, 0, nullptr, dW, 0, 0,
                           wgrad_splits(n_out, k_in, Kp), s.wsk, st);
  };
  if (wg) {
    // tied lm_head: dE[v, :] += sum over the labelled rows of dlogits[r, v] * hn[r, :]
    if (wg->dembed) {
      const int Kl = pad64(n_lab);
      RC(ta_gather_rows_bf16(t.hn, label_rows, s.hl, n_lab, d.D, st));       // (the forward's copy lived in ITS workspace)
      RC(ta_transpose_to_bf16(t.dlogits, 0, w->vocab_pad, 0, 0, s.tA, Kl, n_lab, w->vocab_pad, st));
      RC(ta_transpose_to_bf16(s.hl, 0, d.D, 0, 0, s.tB, Kl, n_lab, d.D, st));
      RC(ta_gemm_bf16_nt(s.tA, s.tB, wg->dembed, w->vocab, d.D, Kl, Kl, 0, 0, d.D, 0, 0, 0, nullptr, wg->dembed, 0, 0, 1,
                         nullptr, st));
    }
    if (wg->dnorm) RC(ta_rmsnorm_dw(s.dhn, 0, t.x_final, lm_res_bf16(w), t.r_f, wg->dnorm, M, d.D, st));
  }
  float* dx = s.dxa;      // gradient w.r.t. the residual stream (f32) + its bf16 image for the GEMMs
  float* dx_alt = s.dxb32;
  // dyb: the gradient of an RMSNorm OUTPUT is bf16 in both stream modes.  It is read exactly once, so fp32 there only doubles the
  // producing dX GEMM's epilogue bytes and this read (the accumulating d(x) stream keeps its own storage mode).  That is also the
  // recipe's arithmetic: under bf16 autocast the Linear behind the norm takes a bf16 copy of its input, so autograd's gradient for it
  // is a bf16 tensor that is only cast up on its way into the norm's backward (round 6; rounds 4-5 wrote fp32 here in the
  // fp32-stream mode: 64.4 us per dX GEMM against 62.9, 20.6 us per RMSNorm backward against ~18).
  const int gb = 1;
  // round 4: with the bf16 forward stream the d(x) stream is bf16 as well (ta_lm_weights.dx_f32 = 1: fp32 as in rounds 1-3) -- the
  // reference's bf16 model back-propagates bf16 gradients of its bf16 activations; s.dxb then IS the stream (updated in place by
  // every RMSNorm backward) and the two f32 images are written only by the last call, for the f32 consumers below the stack:
  // 50 MB instead of 88 MB per RMSNorm backward
  const bool dx_bf16 = lm_res_bf16(w) && w->dx_f32 == 0;
  auto norm_bwd = [&](const float* dy, int dyb, const float* x, const float* r, const float* gw, const float* dres, float* dxf,
                      bool last = false) -> int {
    if (dx_bf16) return ta_rmsnorm_bwd_bf16s(dy, dyb, x, r, gw, dres ? s.dxb : nullptr, last ? dxf : nullptr, s.dxb, M, d.D, st);
    if (lm_res_bf16(w)) return ta_rmsnorm_bwd_bf16(dy, dyb, x, r, gw, dres, dxf, s.dxb, M, d.D, st);
    return dyb ? ta_rmsnorm_bwd_dyb(dy, x, r, gw, dres, dxf, s.dxb, M, d.D, st)
               : ta_rmsnorm_bwd(dy, x, r, gw, dres, dxf, s.dxb, nullptr, M, d.D, 0, st);
  };
  RC(norm_bwd(s.dhn, 0, t.x_final, t.r_f, w->norm_w, nullptr, dx, w->n_layers == 0));
  for (int l = w->n_layers - 1; l >= 0; --l) {
    const ta_lm_layer& Lw = w->layers[l];
    const LmLayerTape& p = store[l];
    // ---- MLP: x2 = x1 + down(silu(gate) * up)
    if (lgm & 8) RC(lora_bwd(s.dxb, d.D, p.act_s, d.F, p.xa_d, p.i_d, lora_grads[l].dla_d, lora_grads[l].dlb_d, 1, 1 << 30, 1 << 30, l, 3));
    const ta_lm_layer_wgrads* g = wg ? &wg->layers[l] : nullptr;
    if (g) RC(wgrad(s.dxb, d.D, p.act_s, d.F, g->dwd));
    // (Fusing the SwiGLU backward into this GEMM's epilogue measured 0.4 ms per step SLOWER -- at one workgroup per CU nothing overlaps
    // an epilogue, so bytes moved there cost more than the separate streaming kernel at 6 TB/s; the form left the library in round 5.)
    RC(gemm_opt(s.dxb, Lw.wd_t, s.dact, M, d.F, d.D, nullptr, nullptr, 0, 1, take_ext(), st));
    RC(ta_swiglu_bwd(s.dact, p.gu, s.dgu, M, d.F, st));
    if (lgm & 4) RC(lora_bwd(s.dgu, 2 * d.F, p.xn2_s, d.D, p.xa_gu, p.i_gu, lora_grads[l].dla_gu, lora_grads[l].dlb_gu, 2, d.F, 1 << 30, l, 2));
    if (g) RC(wgrad(s.dgu, 2 * d.F, p.xn2_s, d.D, g->dwgu));
    RC(gemm_opt(s.dgu, Lw.wgu_t, s.dxn, M, d.D, 2 * d.F, nullptr, nullptr, 0, gb, take_ext(), st));
    if (g && g->dln_post) RC(ta_rmsnorm_dw(s.dxn, gb, p.x1, lm_res_bf16(w), p.r_post, g->dln_post, M, d.D, st));
    RC(norm_bwd(s.dxn, gb, p.x1, p.r_post, Lw.ln_post_w, dx, dx_alt));
    // ---- attention: x1 = x + o_proj(attn)
    if (lgm & 2) RC(lora_bwd(s.dxb, d.D, p.ao, bq, p.xa_o, p.i_o, lora_grads[l].dla_o, lora_grads[l].dlb_o, 1, 1 << 30, 1 << 30, l, 1));
    if (g) RC(wgrad(s.dxb, d.D, p.ao, bq, g->dwo));
    RC(gemm_opt(s.dxb, Lw.wo_t, s.dao, M, d.nq * d.hd, d.D, nullptr, nullptr, 0, 1, take_ext(), st));
    // frozen q_norm / k_norm: the q|k|v post-processing backward rides in the attention backward's epilogue; trainable norms (full
    // fine-tuning) take head-major dQ / dK / dV + ta_lm_qkv_post_bwd, which also produces d(q_norm) / d(k_norm).
    // (Round 4 built two more forms and removed them in round 5 after measuring them slower: one workgroup per (clip, kv head) with
    // K / V resident -- 92.8 us warm / 117.5 cold per layer against 91.8 / 106.0 here --, and Delta = rowsum(dO o O) inside this
    // kernel -- 87.3 us against 67.2 + 9.5; profiles/r04_g/h_*, r04_zj_*.)
    RC(ta_attn_bwd_prep(s.dao, p.ao, s.delta, s.dot, B, d.nq, L, d.Lp, st));
    if (!(g && (g->dqn || g->dkn))) {
      RC(ta_attention_bwd_qkv(p.q, p.k, (w->train_base || !lm_attn_fused(d, L)) ? p.v : nullptr, s.dao, (long)d.nq * d.hd, p.lse, s.delta, kmask, p.qkv0, p.rq, p.rk, Lw.qn_w, Lw.kn_w,
                              w->rope_cos, w->rope_sin, pos, s.dqkv, B, d.nq, d.nkv, L, d.Lp, d.hd, 1, scale, st));
    } else {
      RC(ta_attention_bwd(p.q, p.qt, p.k, p.kt, p.v, s.dao, (long)d.nq * d.hd, s.dot, p.lse, s.delta, kmask, s.dq, s.dk, s.dv,
                          B, d.nq, d.nkv, L, d.Lp, d.hd, 1, scale, st));
      RC(ta_lm_qkv_post_bwd(s.dq, s.dk, s.dv, p.qkv0, p.rq, p.rk, Lw.qn_w, Lw.kn_w, w->rope_cos, w->rope_sin, pos, s.dqkv,
                            g ? g->dqn : nullptr, g ? g->dkn : nullptr, B, d.nq, d.nkv, L, st));
    }
    if (g) RC(wgrad(s.dqkv, d.NQKV, p.xn_s, d.D, g->dwqkv));
    if (lgm & 1) RC(lora_bwd(s.dqkv, d.NQKV, p.xn_s, d.D, p.xa_qkv, p.i_qkv, lora_grads[l].dla_qkv, lora_grads[l].dlb_qkv, 3, bq, bk, l, 0));
    RC(gemm_opt(s.dqkv, Lw.wqkv_t, s.dxn, M, d.D, d.NQKV, nullptr, nullptr, 0, gb, take_ext(), st));
    if (g && g->dln_in) RC(ta_rmsnorm_dw(s.dxn, gb, p.x_in, lm_res_bf16(w), p.r_in, g->dln_in, M, d.D, st));
    RC(norm_bwd(s.dxn, gb, p.x_in, p.r_in, Lw.ln_in_w, dx_alt, dx, l == 0));
  }
  if (d_embeds && hipMemcpyAsync(d_embeds, dx, (size_t)M * d.D * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return TA_ERR_LAUNCH;
  if (d_audio && src_row) RC(ta_audio_grad_gather(src_row, dx, d_audio, M, d.D, st));
  if (wg && wg->dembed) RC(ta_embed_grad_scatter(ids, src_row, dx, wg->dembed, M, d.D, w->vocab, st));
  if (lora_parts) {                                  // second level of the adapter-gradient reduction: every layer, every kind, one launch
    LoraReduceDesc rd;
    float* const f0[8] = {lora_grads[0].dla_qkv, lora_grads[0].dlb_qkv, lora_grads[0].dla_o, lora_grads[0].dlb_o,
                          lora_grads[0].dla_gu, lora_grads[0].dlb_gu, lora_grads[0].dla_d, lora_grads[0].dlb_d};
    for (int k = 0; k < 8; ++k) {
      const bool on = (lgm >> (k >> 1)) & 1;
      rd.out[k] = f0[k]; rd.out_ls[k] = s.lp_size[k]; rd.size[k] = on ? s.lp_size[k] : 0; rd.part_off[k] = s.lp_off[k];
      rd.chunks[k] = s.lp_chunks[k];
    }
    rd.part = s.lora_part; rd.part_ls = s.lp_layer;
    RC(ta_i_lora_reduce_parts(rd, w->n_layers, st));
  }
  return TA_OK;
}
