/* ta355.h -- C ABI of libta355.so: the MI355X-native (gfx950) projector-training hot path of
 * alexkroman/tiny-audio.  Plain pointers + sizes + a hipStream_t; no torch types; every function
 * returns an int status (0 = ok, 1 = bad argument, 2 = launch failure) and never allocates device
 * memory: callers pass workspaces sized by the matching *_workspace_bytes() query.  All pointers are
 * DEVICE pointers unless a comment says "host".  `long` is 64-bit (LP64).  bf16 buffers are passed as
 * `void*` (raw bfloat16 bits).
 *
 * The reference has no FFI of its own (it is 100 % Python on top of torch/transformers); the seams this
 * library sits behind are the reference's nn.Module boundaries.  Each entry point
 * names the reference interface it replaces; INTEGRATION.md shows the ctypes stubs a maintainer adds.
 */
#ifndef TA355_H
#define TA355_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define TA_OK 0
#define TA_ERR_ARG 1
#define TA_ERR_LAUNCH 2

int ta_version(void); /* ABI version, currently 4 (round 6: the residual-stream storage mode left process-wide state -- ta_set_stream_modes /
                          ta_get_stream_modes are gone, ta_encoder_weights.res_f32 and ta_lm_weights.res_f32 / dx_f32 carry it per handle; ta_rmsnorm_bwd_dyb is new;
                          ta_attention_fwd_qkv / ta_attention_bwd_qkv accept V = NULL.
                          Version 3 was round 5's: the opt-in paths that measured slower left the library -- ta_enc_layer lost its
                          wqk_il / *_ln image fields, ta_gemm_opts its swiglu_* / lnf_* fields, ta_layernorm_stats / ta_attention_bwd_gqa /
                          ta_attention_bwd_qkv_o went.  Version 2 was round 3's: ta_gemm_opts.rope_cols, ta_enc_layer.wqkv_fa / bqkv_fa,
                          ta_attention_enc_fwd, ta_logmel_f32's scratch contract) */

/* ---- storage dtype of the residual streams (the numerics contract of DESIGN.md section 6): a field of the weights handle
 * each call receives (ta_encoder_weights.res_f32; ta_lm_weights.res_f32 / dx_f32), so two models of different model_dtype -- or a
 * decoding thread beside a training step (tiny_audio/asr_modeling.py:733-734) -- share the library without sharing a mode.
 * The reference runs its frozen models either as bf16 MODULES (ASRConfig default model_dtype="bfloat16",
 * tiny_audio/asr_config.py:41: every residual add / norm input is bf16) or -- the training recipe of BASELINE configs[1] -- as
 * fp32 modules under bf16 autocast (configs/config.yaml:14-18 + configs/training/production.yaml:49; loaders
 * tiny_audio/asr_modeling.py:203-254: residual stream, norm in/out and embeddings stay fp32, only Linear / attention run in bf16).
 * 0 = bf16 storage (the first regime), 1 = fp32 storage (the second).
 * MFMA operand (bf16) and accumulator (fp32) types are the same in both; norms / softmax / CE arithmetic is fp32 in both.
 * A forward's tape must be read back by a backward whose handle carries the same res_f32 / dx_f32.  Workspace / tape sizes do not
 * depend on the mode (the fp32 size is always reserved). */

/* ============================================================================================
 * Composite ops (what a binding would call)
 * ============================================================================================ */

/* ---- log-mel features: replaces WhisperFeatureExtractor.__call__ as used at
 *      scripts/train.py:327-333 and tiny_audio/asr_processing.py:74-80
 *      (TF:models/whisper/feature_extraction_whisper.py:135-168,330-339).
 * wav [B, Ls] f32 zero-padded to the longest clip, lens [B] true sample counts.
 * dft [400, 402] / window [400] / melfb [201, n_mels] are host-built constant tables uploaded once (of dft the kernel reads row
 *      n = 1, the twiddles W400^j of its 16 x 25 mixed-radix FFT).
 * feats [B, n_mels, T] f32, mask [B, T] int32, T = Ls / 160.
 * scratch: float[ta_logmel_scratch_floats(B, Ls, n_mels)] -- every persistent workgroup records the maximum of its tiles per clip
 *      there (plain stores; no initial contents required, no atomics) and the second launch reduces them into the (max - 8) floor.
 *      (ABI 2: int[2 * B] behind an init launch, one device-scope atomicMax per workgroup.)
 * mel_ranges: int[2 * n_mels] {first, end} frequency bin of every mel filter, from ta_logmel_mel_ranges -- a function of the
 *      filter bank alone, computed once when the tables are uploaded. */
long ta_logmel_scratch_floats(int B, int Ls, int n_mels);
int ta_logmel_mel_ranges(const float* melfb, int n_mels, int* mel_ranges, hipStream_t st);
int ta_logmel_f32(const float* wav, const long* lens, int B, int Ls, const float* dft, const float* window,
                  const float* melfb, int n_mels, float* feats, int* mask, float* scratch, const int* mel_ranges, hipStream_t st);

/* ---- frozen GLM-ASR encoder: replaces model.audio_tower(input_features=...).last_hidden_state
 *      (tiny_audio/asr_modeling.py:448-450; TF:models/glmasr/modeling_glmasr.py:313-327). */
typedef struct {
  const float *ln1_w, *ln1_b, *ln2_w, *ln2_b; /* [H] */
  const void* wqkv;   /* bf16 [3H, H] rows = q | k | v */
  const float* bqkv;  /* [3H] (k part zero: k_proj has no bias) */
  const void* wo;     /* bf16 [H, H] */
  const float* bo;
  const void* w1;     /* bf16 [F, H] */
  const float* b1;
  const void* w2;     /* bf16 [H, F] */
  const float* b2;
  /* Optional (both NULL = the three-kernel path: q|k|v GEMM + ta_enc_qkv_post + ta_attention_fwd): the image of the
   * ta_attention_enc_fwd path.
   *   wqkv_fa bf16 [3H, H]: the q rows and k rows with every head's rows reordered for ta_gemm_opts.rope_tab (row 2i = head dim i,
   *           row 2i+1 = head dim i+16 for i < 16, rows 32..63 unchanged), the q rows MULTIPLIED by head_dim^-0.5 * log2(e) (the
   *           softmax then runs in base 2 with no per-score multiply), then the v rows as they are;
   *   bqkv_fa [3H] likewise (q part scaled, k part zero, v part = v_proj.bias).
   * q | k | v then come out of ONE GEMM (rope on the first 2H columns: ta_gemm_opts.rope_cols) as a token-major [M, 3H]
   * buffer that the attention kernel reads in place -- no V^T image, no M % 8 restriction. */
  const void* wqkv_fa;
  const float* bqkv_fa;
} ta_enc_layer;

typedef struct {
  int hidden, ffn, n_layers, heads, n_mels, max_pos;
  float ln_eps;
  const void* conv1_w;  /* bf16 [H, 3*n_mels], column = tap*n_mels + cin */
  const float* conv1_b;
  const void* conv2_w;  /* bf16 [H, 3*H], column = tap*H + cin */
  const float* conv2_b;
  const float *norm_w, *norm_b;
  const float *rope_cos, *rope_sin; /* [max_pos, 16] (partial rotary: 32 of 64 dims) */
  const ta_enc_layer* layers;       /* host array [n_layers] */
  const float* rope_il;             /* [max_pos, 16, 2] (cos, sin) interleaved, or NULL (see ta_enc_layer.wqkv_fa) */
  int res_f32;                      /* storage of the residual stream: 0 = bf16, 1 = fp32 (see the top of this file) */
} ta_encoder_weights;

long ta_encoder_workspace_bytes(const ta_encoder_weights* w, int B, int T);
/* feats [B, n_mels, T] f32 -> out_bf16 [B*S, H] (and/or out_f32), S = (T-1)/2+1.
 * frame_keep [B*S] f32 or NULL: the train-time whole-frame dropout mask of
 * ASRModel._maybe_drop_audio_tokens (tiny_audio/asr_modeling.py:458-479), fused into the final LayerNorm. */
int ta_encoder_forward(const ta_encoder_weights* w, const float* feats, int B, int T, const float* frame_keep,
                       void* out_bf16, float* out_f32, void* ws, long ws_bytes, hipStream_t st);

/* ---- MLP projector: replaces MLPAudioProjector.forward + its autograd backward
 *      (tiny_audio/projectors.py:57-71,79-87). x bf16 [B, S, E] -> y f32 [B*N, D], N = (S-k)/k+1. */
typedef struct {
  int enc_dim, k, hidden, llm_dim;
  float eps;
  const void* w1;   /* bf16 [Hd, k*E]   (cast of linear_1.weight) */
  const void* w2;   /* bf16 [D, Hd]     (cast of linear_2.weight) */
  const void* w2_t; /* bf16 [Hd, D]     (transposed cast, for dA = dH2 W2) */
  const float* g1;  /* norm.weight   [Hd] */
  const float* g2;  /* norm_2.weight [D]  */
} ta_mlp_weights;

long ta_mlp_tape_bytes(const ta_mlp_weights* w, int B, int S);
long ta_mlp_bwd_workspace_bytes(const ta_mlp_weights* w, int B, int S);
int ta_mlp_projector_forward(const ta_mlp_weights* w, const void* x_bf16, int B, int S, float* y, void* tape,
                             hipStream_t st);
/* dy f32 [B*N, D] -> dW1 [Hd,kE], dg1 [Hd], dW2 [D,Hd], dg2 [D] (f32, overwritten). */
int ta_mlp_projector_backward(const ta_mlp_weights* w, const void* x_bf16, int B, int S, const float* dy,
                              const void* tape, float* dW1, float* dg1, float* dW2, float* dg2, void* ws,
                              long ws_bytes, hipStream_t st);

/* ---- shared + sparse MoE projector: replaces MoEAudioProjector.forward/_forward_sparse + autograd backward
 *      (tiny_audio/projectors.py:185-351): frame-stack -> RMSNorm(kE) -> shared SimpleAdapter + top-2-of-E routed
 *      SimpleAdapters (fc1+bias -> erf-GELU -> fc2+bias), fp32 softmax router with optional multiplicative jitter,
 *      balance + z auxiliary loss.  Arrays w1.. hold E+1 device pointers (HOST arrays): routed experts 0..E-1, then
 *      the shared expert at index E. */
typedef struct {
  int enc_dim, k, hidden, llm_dim, num_experts;
  float eps, aux_coef, z_coef;
  const float* norm_w;       /* [k*E_enc] */
  const float* router_w;     /* f32 [E, k*E_enc] */
  const void* const* w1;     /* bf16 [Hd, kE] */
  const void* const* w1_t;   /* bf16 [kE, Hd] */
  const float* const* b1;    /* [Hd] */
  const void* const* w2;     /* bf16 [D, Hd] */
  const void* const* w2_t;   /* bf16 [Hd, D] */
  const float* const* b2;    /* [D] */
} ta_moe_weights;

/* bf16 images of all n = E + 1 adapters (routed experts, then the shared one) in two launches: f32 masters w1[i] [H, In], w2[i] [D, H],
 * b1[i] [H], b2[i] [D] (host arrays of device pointers) -> stacked W1a [n, H, In], W1ta [n, In, H], W2a [n, D, H], W2ta [n, H, D]
 * (bf16) and B1a [n, H], B2a [n, D] (f32): what ta_moe_weights points into (tiny_audio/projectors.py:257-283 keeps one nn.Linear
 * pair per expert). */
int ta_moe_pack_images(const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2, int n, int H, int In,
                       int D, void* W1a, void* W1ta, void* W2a, void* W2ta, float* B1a, float* B2a, hipStream_t st);
long ta_moe_tape_bytes(const ta_moe_weights* w, int B, int S);
long ta_moe_bwd_workspace_bytes(const ta_moe_weights* w, int B, int S);
/* noise: [T, E] jitter factors (train mode; the reference draws U(1-0.01, 1+0.01)) or NULL; aux: device scalar out. */
int ta_moe_projector_forward(const ta_moe_weights* w, const void* x_bf16, int B, int S, const float* noise, int training,
                             float* y, float* aux, void* tape, hipStream_t st);
/* gradients of sum(dy * y) + d_aux * aux; dW1/db1/dW2/db2 are HOST arrays of E+1 device pointers (overwritten). */
int ta_moe_projector_backward(const ta_moe_weights* w, const void* x_bf16, int B, int S, const float* dy, float d_aux,
                              const float* noise, int training, const void* tape, float* d_norm_w, float* d_router_w,
                              float* const* dW1, float* const* db1, float* const* dW2, float* const* db2, void* ws,
                              long ws_bytes, hipStream_t st);
/* The same with the upstream gradient of the auxiliary loss read from DEVICE memory (d_aux_dev: one f32): the autograd
 * engine hands it over as a device scalar, and reading it on the host would stall the launch queue once per step. */
int ta_moe_projector_backward_dev(const ta_moe_weights* w, const void* x_bf16, int B, int S, const float* dy,
                                  const float* d_aux_dev, const float* noise, int training, const void* tape,
                                  float* d_norm_w, float* d_router_w, float* const* dW1, float* const* db1,
                                  float* const* dW2, float* const* db2, void* ws, long ws_bytes, hipStream_t st);

/* round 4: the share of d(norm.weight) [k * enc_dim] and d(router.weight) [E, k * enc_dim] that comes from the auxiliary losses alone
 * (d_aux * d aux / d .; overwritten).  Same tape / workspace as the backward.  The trainer keeps it in a shadow of its flat gradient
 * buffer so that the auxiliary term keeps its full weight under the optimizer's division by the global label-token count without a
 * second collective (tiny_audio_amd/trainer.py). */
int ta_moe_router_aux_grads(const ta_moe_weights* w, const void* x_bf16, int B, int S, const float* d_aux_dev, const float* noise,
                            int training, const void* tape, float* d_norm_w_aux, float* d_router_w_aux, void* ws, long ws_bytes,
                            hipStream_t st);

/* ---- frozen Qwen3 LM + shifted CE: replaces model.language_model(inputs_embeds=, attention_mask=, labels=)
 *      together with the embed/masked_scatter glue of ASRModel.forward
 *      (tiny_audio/asr_modeling.py:497-526; TF:models/qwen3/modeling_qwen3.py:367-508; TF:loss/loss_utils.py:33-71)
 *      and its activation-gradient backward (encoder + LM frozen: dX only). */
typedef struct {
  const float* ln_in_w;             /* [D] */
  const void *wqkv, *wqkv_t;        /* bf16 [NQKV, D], [D, NQKV]; rows = q | k | v */
  const float *qn_w, *kn_w;         /* [head_dim] */
  const void *wo, *wo_t;            /* bf16 [D, Hq*hd], [Hq*hd, D] */
  const float* ln_post_w;           /* [D] */
  const void *wgu, *wgu_t;          /* bf16 [2F, D] rows = gate | up, [D, 2F] */
  const void *wd, *wd_t;            /* bf16 [D, F], [F, D] */
  /* LoRA adapters (stage 2, tiny_audio/asr_modeling.py:289-301; all NULL when disabled): fp32 MASTERS, the peft
   * tensors of the linears that share an input stacked on rows, group g in {qkv, o, gate|up, down}:
   *   la_g [members*r, in_g] = concat of lora_A_j [r, in];   lb_g [N_g, r] = concat of lora_B_j [out_j, r]. */
  const float *la_qkv, *lb_qkv, *la_o, *lb_o, *la_gu, *lb_gu, *la_d, *lb_d;
} ta_lm_layer;

/* gradients of the LoRA masters of one layer (same layouts; written, not accumulated) */
typedef struct {
  float *dla_qkv, *dlb_qkv, *dla_o, *dlb_o, *dla_gu, *dlb_gu, *dla_d, *dlb_d;
} ta_lm_lora_grads;

typedef struct {
  int vocab, vocab_pad, hidden, ffn, n_layers, heads, kv_heads, head_dim, max_pos;
  float eps;
  const float* embed_f32;   /* [vocab, D]   input embedding (fp32 master, as the reference looks it up) */
  const void* embed_bf16;   /* [vocab_pad, D] tied lm_head, rows >= vocab zero */
  const void* embed_t_bf16; /* [D, vocab_pad] */
  const float* norm_w;
  const float *rope_cos, *rope_sin; /* [max_pos, head_dim/2] */
  const ta_lm_layer* layers;        /* host array */
  int lora_rank;                    /* 0 = no adapters; else r with 3 r <= 64 (the reference default 8; 4 and 16 are tested) */
  float lora_scale;                 /* alpha / r */
  int train_base;                   /* 1 = full decoder fine-tuning (freeze_language_model=False, tiny_audio/asr_config.py:77,
                                       configs/experiments/embedded.yaml:23): the tape also keeps what ta_lm_wgrads needs */
  int lora_groups;                  /* lora_target_modules subsets (tiny_audio/asr_config.py:72-75): bit g set = group g of
                                       {1 q|k|v, 2 o, 4 gate|up, 8 down} carries an adapter; 0 = all four.  A group without
                                       a bit costs nothing (no rank-space GEMM, no K extension, gradients left at zero); inside
                                       a group, members that are not targeted keep A = B = 0 in the masters and so get exactly
                                       zero gradients (dA = s (dy B)^T x, dB = dy^T (x A^T)). */
  int res_f32;                      /* storage of the forward residual stream and of its rows in the tape: 0 = bf16, 1 = fp32 */
  int dx_f32;                       /* storage of the backward d(x) stream: 0 = follows res_f32, 1 = fp32 even over a bf16 forward
                                       stream (rounds 1-3) */
} ta_lm_weights;

/* Gradients of the LM's own weights (full decoder fine-tuning).  All f32, ACCUMULATED (+=) into the caller's buffers, which
 * the caller zeroes at the start of an optimizer step (gradient accumulation over micro-batches then needs nothing else).
 * dembed [vocab, D] receives both shares of the tied matrix: lm_head (dlogits^T h) and the input lookup (scatter-add of
 * d inputs_embeds at the text positions; <audio> rows were overwritten by the projector output and get none). */
typedef struct {
  float *dwqkv;  /* [NQKV, D] rows = q | k | v */
  float *dwo;    /* [D, heads*head_dim] */
  float *dwgu;   /* [2F, D] rows = gate | up */
  float *dwd;    /* [D, F] */
  float *dln_in, *dln_post;   /* [D] */
  float *dqn, *dkn;           /* [head_dim] */
} ta_lm_layer_wgrads;
typedef struct {
  const ta_lm_layer_wgrads* layers;   /* host array [n_layers] */
  float* dnorm;                       /* [D] */
  float* dembed;                      /* [vocab, D] */
} ta_lm_wgrads;

long ta_lm_tape_bytes(const ta_lm_weights* w, int B, int L, int n_label_rows);
long ta_lm_workspace_bytes(const ta_lm_weights* w, int B, int L, int n_label_rows);
/* ids [B,L] i64; src_row [B*L] from ta_audio_index (or NULL = text only); audio f32 [*, D] projector output;
 * kmask [B,L] int32 attention mask (NULL = all ones); pos [B*L] int32 (NULL = arange(L));
 * label_rows/label_targets: the n_label_rows positions whose shifted label != -100 (ta_label_rows);
 * loss_scale = 1 / num_items_in_batch.  loss: device scalar, ACCUMULATED into (zero it first).
 * logits_out: optional [B*L, vocab_pad] bf16 full logits (the reference's outputs.logits), else NULL. */
int ta_lm_forward_loss(const ta_lm_weights* w, const long* ids, const int* src_row, const float* audio,
                       const int* kmask, const int* pos, int B, int L, const int* label_rows,
                       const long* label_targets, int n_label_rows, float loss_scale, float* loss,
                       float* nll_rows, void* logits_out, void* tape, void* ws, long ws_bytes, hipStream_t st);
/* ---- primitives of the trainable transformer-style projectors (QFormer / MOSA, SURVEY.md section 8(f) rank 4;
 * tiny_audio/projectors.py:88-182,359-475 over TF:models/blip_2/modeling_blip_2.py Blip2QFormer*).  The linears are
 * ta_gemm_bf16_nt; these are the pieces around them. */
int ta_gelu_fwd(const void* h_bf16, void* a_bf16, long n, hipStream_t st);                     /* nn.GELU (erf) */
int ta_gelu_bwd(const void* da_bf16, const void* h_bf16, void* dh_bf16, long n, hipStream_t st);
int ta_colsum(const void* x, int is_f32, int R, int C, float* out, hipStream_t st);           /* out[c] = sum_r x[r,c]: bias grads */
int ta_relu_fwd(const void* h_bf16, void* a_bf16, long n, hipStream_t st);                     /* MOSA router (projectors.py:136-140) */
int ta_relu_bwd(const void* da_bf16, const void* h_bf16, void* dh_bf16, long n, hipStream_t st);
/* MOSA dense mixture (projectors.py:158-166): rw = softmax(logits [M,E]); out = sum_e rw[:,e] * o[e] with o f32 [E, M, D].
 * Backward: do_bf16 [E, M, D] = dout * rw[:,e]; dlogits [M,E] = softmax backward of drw[e] = <dout, o[e]>. */
int ta_mix_fwd(const float* logits, const float* o, float* rw, float* out, int M, int D, int E, hipStream_t st);
int ta_mix_bwd(const float* dout, const float* o, const float* rw, void* do_bf16, float* dlogits, int M, int D, int E,
               hipStream_t st);
/* y = LayerNorm(z * keep + res[row % res_rows]) (keep: dropout mask already divided by 1-p, or NULL; res optional);
 * saves xhat [M,H] and rstd [M] for ta_layernorm_bwd.  Blip2QFormerSelfOutput / Output: dense -> dropout -> +res -> LN. */
int ta_layernorm_res_fwd(const float* z, const float* keep, const float* res, long res_rows, const float* gamma,
                         const float* beta, float eps, float* xhat, float* rstd, float* y_f32, void* y_bf16, int M, int H,
                         hipStream_t st);
/* du (f32, gradient of the LN input = residual branch) and dz = du * keep (bf16, dense branch) are optional;
 * dgamma / dbeta are ACCUMULATED. */
int ta_layernorm_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, const float* keep, float* du,
                     void* dz_bf16, float* dgamma, float* dbeta, int M, int H, hipStream_t st);
/* Windowed multi-head attention (Blip2QFormerMultiHeadAttention, eager): EB windows, Q bf16 [EB*Lq, heads*hd],
 * K / V bf16 [EB*Lk, heads*hd]; P f32 [EB, heads, Lq, Lk] = softmax probabilities (saved); keep = attention-probs
 * dropout mask / (1-p) or NULL; O bf16 [EB*Lq, heads*hd]. */
int ta_attn_small_fwd(const void* Q, const void* K, const void* V, int EB, int heads, int hd, int Lq, int Lk, float scale,
                      const float* keep, float* P, void* O, hipStream_t st);
int ta_attn_small_bwd(const void* dO, const void* Q, const void* K, const void* V, const float* P, const float* keep,
                      float scale, void* dQ, void* dK, void* dV, int EB, int heads, int hd, int Lq, int Lk, hipStream_t st);

/* ---- greedy decoding: ASRModel.generate, tiny_audio/asr_modeling.py:562-646, which
 * drives HF GenerationMixin greedy search (num_beams 1, do_sample False: tiny_audio/asr_config.py:103-111) with a KV cache.
 * kcache / vcache: bf16 [n_layers, B, kv_heads, Lmax, head_dim], owned by the caller.  lora_img: scratch of
 * ta_lm_lora_image_bytes() bytes holding the bf16 adapter images between the prompt pass and the decode steps (NULL
 * without adapters).
 * ta_lm_prefill: the prompt pass over [B, L] rows (same inputs as ta_lm_forward_loss); fills cache slots [0, L) and
 * returns logits f32 [B, vocab_pad] of row last_rows[b] (= b*L + index of clip b's last prompt token). */
long ta_lm_prefill_workspace_bytes(const ta_lm_weights* w, int B, int L);
long ta_lm_lora_image_bytes(const ta_lm_weights* w);
int ta_lm_prefill(const ta_lm_weights* w, const long* ids, const int* src_row, const float* audio, const int* kmask,
                  const int* pos, int B, int L, void* kcache, void* vcache, int Lmax, const int* last_rows, float* logits,
                  void* lora_img, void* ws, long ws_bytes, hipStream_t st);
/* One decode step: ids [B] = the tokens emitted last, at RoPE position pos[b]; their K/V go to cache slot *slot_dev;
 * kmask [B, Lmax] marks the valid slots (including *slot_dev); logits f32 [B, vocab_pad] of the next token.  Every
 * per-step quantity is read from device memory, so the launch sequence is step-invariant (hipGraph-capturable). */
long ta_lm_decode_workspace_bytes(const ta_lm_weights* w, int B);
int ta_lm_decode_step(const ta_lm_weights* w, const long* ids, const int* pos, const int* kmask, const int* slot_dev, int B,
                      void* kcache, void* vcache, int Lmax, float* logits, const void* lora_img, void* ws, long ws_bytes,
                      hipStream_t st);
/* out[r] = argmax over the first n columns of row r (lowest index on ties) */
int ta_argmax_f32(const float* x, long ld, int n, int rows, long* out, hipStream_t st);
/* HF greedy bookkeeping for step t = *step_dev (TF:generation/utils.py _sample): finished clips emit pad_id, a clip
 * finishes on any eos id; writes out_seq[b, t], next_ids[b]; advances pos / *slot_dev / kmask for the next decode step
 * (not on t == 0, whose token comes from the prompt pass); *step_dev += 1; *n_unfinished = clips still running. */
/* The reference's non-default greedy settings (tiny_audio/asr_config.py:84-86,155-160; TF:generation/logits_process.py
 * RepetitionPenaltyLogitsProcessor + NoRepeatNGramLogitsProcessor) on the step's logits [B, ld] f32 (V valid columns) BEFORE
 * ta_argmax_f32.  The processors see prompt_ids [B, L] i64 followed by the first *step_dev tokens of out_seq [B, max_new]
 * (the reference passes input_ids to generate for exactly this: tiny_audio/asr_modeling.py:625-633).  repetition_penalty 1.0 and
 * no_repeat_ngram_size 0 make it a no-op; L + max_new <= 8192. */
int ta_logits_process(float* logits, long ld, int V, const long* prompt_ids, int L, const long* out_seq, int max_new,
                      const int* step_dev, int B, float repetition_penalty, int no_repeat_ngram_size, hipStream_t st);
/* Sampling (generation_config.do_sample / temperature / top_k / top_p: tiny_audio/asr_config.py:78-81, forwarded to HF at
 * tiny_audio/asr_modeling.py:631-637).  ta_logits_warp applies HF's warpers in HF's order, in place on logits f32 [B, ld] (columns
 * < V): TemperatureLogitsWarper (scores / T), TopKLogitsWarper (below the k-th largest -> -inf; 0 = off), TopPLogitsWarper (ascending
 * cumulative probability <= 1 - top_p -> -inf, the largest always kept; 1 = off).  ta_sample_f32 draws out[b] ~ softmax(logits[b, :V])
 * with a Philox4x32-10 uniform keyed by (seed, *step_dev, b): reproducible per seed, independent of launch geometry.  Call both
 * between the logits processors and ta_greedy_advance (ta_sample_f32 takes the place of ta_argmax_f32). */
int ta_logits_warp(float* logits, long ld, int V, int B, float temperature, int top_k, float top_p, hipStream_t st);
int ta_sample_f32(const float* logits, long ld, int V, int B, unsigned long long seed, const int* step_dev, long* out, hipStream_t st);
/* HF MinNewTokensLengthLogitsProcessor (generation_config.min_new_tokens: tiny_audio/asr_config.py:83, forwarded to
 * language_model.generate at tiny_audio/asr_modeling.py:631-637): while *step_dev (tokens generated so far, device memory) is below
 * min_new, logits[b, ids[e]] = -inf for every eos id.  Call between ta_logits_process and ta_argmax_f32. */
int ta_logits_suppress_until(float* logits, long ld, int V, const long* ids, int n_ids, int min_new, const int* step_dev, int B,
                             hipStream_t st);
int ta_greedy_advance(const long* amax, const long* eos_ids, int n_eos, long pad_id, int* finished, long* next_ids,
                      long* out_seq, int max_new, int* step_dev, int* slot_dev, int* pos, int* kmask, int Lmax, int B,
                      int* n_unfinished, hipStream_t st);

/* loss.backward() through the LM (replaces autograd over Qwen3ForCausalLM, tiny_audio/asr_modeling.py:517-533).
 * d_audio f32 [n_audio_rows, D] (zeroed, then rows referenced by src_row written; NULL = not wanted, e.g. frozen
 * projector); d_embeds optional [B*L, D]; lora_grads: host array [n_layers] (required iff w->lora_rank > 0, zeroed and
 * written); wgrads: NULL for a frozen LM, else the weight-gradient buffers of ta_lm_wgrads (w->train_base must be 1; they
 * are ACCUMULATED into) together with ids [B, L] i64, the token ids of the forward (input-lookup share of the embedding). */
int ta_lm_backward(const ta_lm_weights* w, const int* src_row, const int* kmask, const int* pos, int B, int L,
                   const int* label_rows, int n_label_rows, float* d_audio, long n_audio_rows, float* d_embeds,
                   const ta_lm_lora_grads* lora_grads, const ta_lm_wgrads* wgrads, const long* ids, const void* tape, void* ws,
                   long ws_bytes, hipStream_t st);

/* ============================================================================================
 * Primitive kernels (exported for the parity tests; also what the composites are built from)
 * ============================================================================================ */

/* GLM-ASR encoder self-attention straight from the q|k|v GEMM output (TF:models/glmasr/modeling_glmasr.py:187-217: non-causal,
 * no mask, head_dim 64).  qkv bf16 [B*S, 3*heads*64] token-major, thirds q | k | v, head h at columns h*64 of each third; the
 * q values must already carry head_dim^-0.5 * log2(e) (ta_enc_layer.wqkv_fa): out = softmax_base2(q k^T) v, bf16 [B*S, heads*64]. */
int ta_attention_enc_fwd(const void* qkv, void* out, int B, int heads, int S, hipStream_t st);

/* C = epilogue(A[M,K] x W[N,K]^T).  A/C rows are mapped  row -> (row / rpb) * bs + (row % rpb) * ld
 * (rpb <= 0 means "no batching").  act: 0 none, 1 erf-GELU.  residual: f32, C's row map.
 * splits > 1: split-K through splitk_ws (f32 [splits, M, N]); then no bias/act, plain C layout. */
int ta_gemm_bf16_nt(const void* A, const void* W, void* C, int M, int N, int K, long lda, int a_rpb, long a_bs,
                    long ldc, int c_rpb, long c_bs, long c_off, const float* bias, const float* residual, int act,
                    int out_bf16, int splits, float* splitk_ws, hipStream_t st);
/* Grouped / routed form (MoE experts): a_idx (gather list for A rows), seg {row base, row count} and krange
 * {first, end} 64-wide K tile are DEVICE int arrays read by the kernel, so routing counts never visit the host.
 * M is then only the upper bound used to size the grid. */
int ta_gemm_bf16_nt_ex(const void* A, const void* W, void* C, int M, int N, int K, long lda, int a_rpb, long a_bs,
                       long ldc, int c_rpb, long c_bs, long c_off, const float* bias, const float* residual, int act,
                       int out_bf16, int splits, float* splitk_ws, const int* a_idx, const int* seg,
                       const int* krange, hipStream_t st);
/* Optional extras of one GEMM call (all pointers NULL = plain GEMM).  Passed by value semantics: read during the call,
 * no state is kept (the composites run on any host thread / stream concurrently).
 *   a2/w2/k2/lda2   K extension (LoRA): C = epilogue(A W^T + A2 W2^T), A2 [M, k2] (row stride lda2), W2 [N, k2],
 *                   k2 % 64 == 0, in the same accumulator pass (no split-K, no krange)
 *   residual_bf16   bf16 residual with C's row map (may alias C); the call's f32 `residual` must then be NULL.  The
 *                   frozen models' residual adds in the model dtype (TF:models/glmasr/modeling_glmasr.py:249-270,
 *                   TF:models/qwen3/modeling_qwen3.py:283-324 on bf16 models)
 *   rope_tab/rows   act == 2: GLM-ASR partial rotary embedding (TF:models/glmasr/modeling_glmasr.py:153-168) applied to the
 *                   bf16 result after the bias.  Heads are 64 columns (N % 64 == 0); W's rows must be ordered so that the
 *                   first 32 columns of each head hold the 16 rotation pairs interleaved -- column 2i = head dim i,
 *                   column 2i+1 = head dim i+16 -- and columns 32..63 = head dims 32..63 (q and k permuted alike leave
 *                   q.k unchanged).  rope_tab f32 [rope_rows][16][2] = (cos, sin) of pair i at position p; row m of the
 *                   GEMM is position m % rope_rows. */
typedef struct {
  const void* a2; const void* w2; int k2; long lda2;
  const void* residual_bf16;
  const float* rope_tab; int rope_rows;
  int w_blocked;   /* W is given as [N/64][K/64][64][64] blocks (N % 64 == 0): each 64-row x 64-column K tile 8 KB contiguous */
  int rope_cols;   /* act == 2: the rotary embedding applies to columns [0, rope_cols) only (0 = all N columns); % 64 == 0 */
} ta_gemm_opts;
int ta_gemm_bf16_nt_opt(const void* A, const void* W, void* C, int M, int N, int K, long lda, int a_rpb, long a_bs,
                        long ldc, int c_rpb, long c_bs, long c_off, const float* bias, const float* residual, int act,
                        int out_bf16, int splits, float* splitk_ws, const int* a_idx, const int* seg,
                        const int* krange, const ta_gemm_opts* opts, hipStream_t st);
long ta_gemm_splitk_ws_bytes(int M, int N, int splits);
/* Grouped GEMMs: every expert of the MoE projector in ONE launch (tiny_audio/projectors.py:327-345 loops over the experts
 * in Python).  n_groups <= 8; exactly one of seg / krange is given (DEVICE int arrays, read by the kernel):
 *   rows form     seg = int[2 * n_groups] {row base, row count} over a common row space (the expert-sorted token slots).
 *                 Group e: C[rows of e, :] = act(A[rows of e (through a_idx, if given), :] (W + e * w_stride)^T + bias[e * N ...]).
 *                 M = upper bound of the TOTAL row count (grid sizing); the tile -> (expert, row tile) map is resolved on the
 *                 device, so routing counts never visit the host.
 *   K-slice form  krange = int[2 * n_groups] {first, end} 64-wide K tile.  Group e: (C + e * c_stride)[M, N] = A[:, slice e]
 *                 W[:, slice e]^T, f32, no bias / activation (per-expert weight gradients: the contraction runs over the slot
 *                 axis, which the counting-sort plan keeps contiguous and 64-aligned per expert). */
int ta_gemm_bf16_nt_grouped(const void* A, const void* W, void* C, int M, int N, int K, const float* bias, int act, int out_bf16,
                            const int* a_idx, const int* seg, const int* krange, int n_groups, long w_stride, long c_stride,
                            hipStream_t st);
/* "TN" product (contraction over the ROWS of both row-major operands): out f32 [Ny, Nx] (+)= Y[M, Ny]^T X[M, Nx] -- the
 * weight gradient dW = dY^T X of a linear layer (full decoder fine-tuning) without transposing dY and X first.
 * Ny, Nx multiples of 8; ws: ta_gemm_bf16_tn_ws_bytes(M, Ny, Nx) bytes of scratch (row-chunk partial sums). */
long ta_gemm_bf16_tn_ws_bytes(int M, int Ny, int Nx);
int ta_gemm_bf16_tn(const void* Y, const void* X, float* out, int M, int Ny, int Nx, int accumulate, void* ws, long ws_bytes,
                    hipStream_t st);
/* in-situ GEMM timing for bench.py's roofline leg: HIP events on the launch stream around every GEMM kernel.
 * collect(): host pointers; sums + clears the records (total kernel ms, total 2*M*N*K flops, launches). */
int ta_profile_gemm(int enable);   /* 0 off, 1 time every launch (collect below), 2 log every launch's shape (ta_profile_gemm_log) */
int ta_profile_gemm_collect(double* total_ms, double* total_flops, long* launches);
/* round 4: the launches since ta_profile_gemm(2) as rows of 24 longs {M, N, K, lda, a_rpb, a_bs, ldc, c_rpb, c_bs, c_off, act,
 * out_bf16, has_residual, residual_bf16, has_bias, splits, rope_cols, rope_rows, flags, K2, tile variant, groups, lnf mode,
 * persistent} (flags: 1 gather, 2 segments, 4 K range, 8 K extension, 16 blocked W, 32 grouped, 64 LayerNorm fold, 128 SwiGLU
 * backward, 256 / 512 identity A / C row map).  Host memory; returns the row count (out may be NULL).  The step-shape parity test
 * (tests/test_gpu_round4.py) runs one real B = 32 step under it and replays every distinct launch against an fp32 matmul. */
long ta_profile_gemm_log(long* out, long max_rows);
/* round 4: the GEMM launcher reads its environment knobs (TA355_GEMM_VARIANT, TA355_GEMM_DEBUG, TA355_V7_MASK, ... : DESIGN.md 7b)
 * ONCE, at its first launch; this re-reads them (tests and A/B scripts that switch a knob between launches). */
int ta_gemm_reload_knobs(void);

int ta_layernorm_f32(const float* x, const float* w, const float* b, void* y_bf16, float* y_f32,
                     const float* rowscale, int M, int H, float eps, hipStream_t st);
int ta_layernorm_bf16(const void* x_bf16, const float* w, const float* b, void* y_bf16, float* y_f32, const float* rowscale,
                      int M, int H, float eps, hipStream_t st);            /* same, the row is read as bf16 */
int ta_rmsnorm_fwd(const float* x, const float* w, void* y_bf16, float* y_f32, float* rstd, int M, int H,
                   float eps, int act_gelu, hipStream_t st);
int ta_rmsnorm_bwd(const float* dy, const float* x, const float* rstd, const float* w, const float* dres,
                   float* dx_f32, void* dx_bf16, float* dw_accum, int M, int H, int act_gelu, hipStream_t st);
/* the same with x read as bf16 (frozen weights: no dw, no GELU): the LM's residual stream in the model dtype */
int ta_rmsnorm_fwd_bf16(const void* x_bf16, const float* w, void* y_bf16, float* y_f32, float* rstd, int M, int H, float eps,
                        hipStream_t st);
int ta_rmsnorm_bwd_bf16(const void* dy, int dy_is_bf16, const void* x_bf16, const float* rstd, const float* w,
                        const float* dres, float* dx_f32, void* dx_bf16, int M, int H, hipStream_t st);
/* round 4: the same with the residual gradient ALSO bf16 -- the LM's d(x) stream in the reference's dtype (its bf16 model back-
 * propagates bf16 activations' gradients); dres_bf16 may alias dx_bf16 (updated in place), dx_f32 is optional (NULL except where an
 * f32 consumer follows: the embedding / audio-row gather at the bottom of the stack) */
int ta_rmsnorm_bwd_bf16s(const void* dy, int dy_is_bf16, const void* x_bf16, const float* rstd, const float* w,
                         const void* dres_bf16, float* dx_f32, void* dx_bf16, int M, int H, hipStream_t st);
/* the fp32-stream counterpart (round 6): x, dres, dx_f32 fp32, the incoming gradient bf16 (under the recipe's bf16 autocast the
 * gradient of a Linear's bf16 input IS a bf16 tensor), a bf16 image of dx beside it (either output may be NULL) */
int ta_rmsnorm_bwd_dyb(const void* dy_bf16, const float* x, const float* rstd, const float* w, const float* dres, float* dx_f32,
                       void* dx_bf16, int M, int H, hipStream_t st);

/* d loss / d weight of an RMSNorm (y = w * x * rstd): dw_accum[h] += sum_m dy[m,h] * x[m,h] * rstd[m]; dy and x are f32 or bf16 */
int ta_rmsnorm_dw(const void* dy, int dy_is_bf16, const void* x, int x_is_bf16, const float* rstd, float* dw_accum, int M,
                  int H, hipStream_t st);

int ta_attention_fwd(const void* Q, const void* K, const void* VT, void* O, float* LSE, const int* kmask, int B,
                     int Hq, int Hkv, int L, int Lp, int head_dim, int causal, float scale, hipStream_t st);
/* The same over caller-described operand layouts (element strides; NULL = the head-major defaults above):
 *   Q(b,h,row,d)  = Q  + b*q_bs + h*q_hs + row*q_rs + d        K likewise with k_*
 *   VT(b,h,d,col) = VT + b*v_bs + h*v_hs + d*v_rs + col
 * so the encoder reads q | k straight from the token-major GEMM output [B*L, 2*H*64] and V^T from the [H*64, B*L]
 * result of V^T = Wv xn^T (v_bs = L, v_rs = B*L: key columns beyond L belong to the next clip -- they are masked,
 * but must be readable and finite: keep 64 elements of slack behind the image).  Column offsets that are not
 * multiples of 8 elements are read with narrower loads. */
typedef struct { long q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs; } ta_attn_layout;
int ta_attention_fwd_ex(const void* Q, const void* K, const void* VT, void* O, float* LSE, const int* kmask, int B,
                        int Hq, int Hkv, int L, int Lp, int head_dim, int causal, float scale, const ta_attn_layout* lay,
                        hipStream_t st);
/* ta_lm_qkv_post_fwd + ta_attention_fwd in ONE launch for short causal sequences (head_dim 128, L <= 192, (Hq / Hkv) * ceil(L / 32)
 * <= 12; otherwise TA_ERR_ARG and nothing is launched): qkv0 is the pre-norm q | k | v GEMM output [B*L, (Hq + 2 Hkv) * 128]; writes
 * O [B*L, Hq*128], LSE and the backward's operands Q / K (normalised, rotated) and V head-major [B, heads, L, 128], rq / rk (1 / rms per
 * (token, head)).  V may be NULL (round 6): V is neither normalised nor rotated, so ta_attention_bwd_qkv can read it in place from
 * qkv0 and the head-major copy is only needed by callers that use it themselves (the KV cache of ta_lm_prefill, ta_attention_bwd).
 * tiny_audio path: TF:models/qwen3/modeling_qwen3.py:211-280 forward. */
int ta_attention_fwd_qkv(const void* qkv0, const float* qn_w, const float* kn_w, const float* cosT, const float* sinT,
                         const int* pos, void* Q, void* K, void* V, float* rq, float* rk, void* O, float* LSE,
                         const int* kmask, int B, int Hq, int Hkv, int L, float scale, float eps, hipStream_t st);
/* backward of the causal GQA attention (head_dim 128).  QT / KT / dOT are IGNORED since round 2 (pass NULL): the kernels read the
 * transposed fragments out of their row tiles with ds_read_b64_tr_b16, so no transposed image has to be produced or staged. */
int ta_attention_bwd(const void* Q, const void* QT, const void* K, const void* KT, const void* V, const void* dO,
                     long dO_stride, const void* dOT, const float* LSE, const float* Delta, const int* kmask,
                     void* dQ, void* dK, void* dV, int B, int Hq, int Hkv, int L, int Lp, int head_dim, int causal,
                     float scale, hipStream_t st);
/* the same with ta_lm_qkv_post_bwd fused into its epilogue (frozen q_norm / k_norm): d(qkv0) token-major [B*L, (Hq + 2 Hkv) * 128] is
 * written directly from the f32 accumulators; no head-major dQ / dK / dV (tiny_audio path: TF:models/qwen3/modeling_qwen3.py:211-280 backward).
 * V == NULL: the V rows are read in place from qkv0 (columns (Hq + Hkv) * 128 ...), see ta_attention_fwd_qkv. */
int ta_attention_bwd_qkv(const void* Q, const void* K, const void* V, const void* dO, long dO_stride, const float* LSE,
                         const float* Delta, const int* kmask, const void* qkv0, const float* rq, const float* rk,
                         const float* qn_w, const float* kn_w, const float* cosT, const float* sinT, const int* pos,
                         void* dqkv, int B, int Hq, int Hkv, int L, int Lp, int head_dim, int causal, float scale,
                         hipStream_t st);
int ta_enc_qkv_post(const void* qkv, const float* cosT, const float* sinT, void* Q, void* K, void* VT, int B, int H,
                    int S, int Sp, hipStream_t st);
/* QT / KT / VT: transposed images [B, heads, 128, Lp]; any of them may be NULL (not written).  The forward attention needs VT only. */
int ta_lm_qkv_post_fwd(const void* qkv0, const float* qn_w, const float* kn_w, const float* cosT, const float* sinT,
                       const int* pos, void* Q, void* K, void* V, void* QT, void* KT, void* VT, float* rq, float* rk,
                       int B, int Hq, int Hkv, int L, int Lp, float eps, hipStream_t st);
int ta_lm_qkv_post_bwd(const void* dQ, const void* dK, const void* dV, const void* qkv0, const float* rq,
                       const float* rk, const float* qn_w, const float* kn_w, const float* cosT, const float* sinT,
                       const int* pos, void* dqkv, float* dqn_accum, float* dkn_accum, int B, int Hq, int Hkv, int L, hipStream_t st);   /* dqn/dkn_accum [128] or NULL: += the q_norm / k_norm weight gradients */
/* Delta = rowsum(dO o O); dOT (optional, may be NULL): the transposed dO image, no longer read by ta_attention_bwd */
int ta_attn_bwd_prep(const void* dO, const void* O, float* Delta, void* dOT, int B, int Hq, int L, int Lp,
                     hipStream_t st);

int ta_swiglu_fwd(const void* gu, void* act, long M, int F, hipStream_t st);
int ta_swiglu_bwd(const void* dact, const void* gu, void* dgu, long M, int F, hipStream_t st);
int ta_cast_f32_bf16(const float* x, void* y, long n, hipStream_t st);
int ta_transpose_to_bf16(const void* in, int in_is_f32, long ld_in, long in_bs, int in_rpb, void* out, long ld_out,
                         int R, int C, hipStream_t st);
int ta_feats_to_time_major(const float* feats, void* out, int B, int C, int T, hipStream_t st);
int ta_zero_pad_rows(void* buf, int B, int T, int C, hipStream_t st);

/* <audio> placeholder bookkeeping: _gather_audio_embeds + masked_scatter
 * (tiny_audio/asr_modeling.py:27-44,511-515). */
int ta_audio_index(const long* ids, const long* counts, int* src_row, int B, int L, int N, long audio_id,
                   hipStream_t st);
int ta_embed_scatter(const long* ids, const int* src_row, const float* emb, const float* audio, float* x0,
                     void* x0_bf16, int n_rows, int D, long vocab, hipStream_t st);
int ta_audio_grad_gather(const int* src_row, const float* dx0, float* d_audio, int n_rows, int D, hipStream_t st);
/* input-lookup share of the embedding gradient: dembed[ids[m], :] += dx0[m, :] for every text row (src_row[m] == -1 or src_row NULL) */
int ta_embed_grad_scatter(const long* ids, const int* src_row, const float* dx0, float* dembed, int n_rows, int D, long vocab,
                          hipStream_t st);
int ta_gather_rows_bf16(const void* in, const int* idx, void* out, int n, int D, hipStream_t st);
int ta_scatter_rows_f32(const float* in, const int* idx, float* out, int n, int D, hipStream_t st);
int ta_bernoulli_keep(float* keep, long n, float keep_prob, unsigned long long seed, hipStream_t st);
/* Instrumentation (not on the hot path): `workgroups` x 256 threads stay resident for `micros` microseconds streaming buf [bytes]
 * (read + write back) -- the footprint of a ring all-reduce's channel workgroups next to the step's kernels.  scripts/allreduce_footprint.py
 * uses it to choose the N > 1 default (overlapped vs synchronous all-reduce) on a one-GPU box. */
int ta_debug_occupy(int workgroups, double micros, void* buf, long bytes, hipStream_t st);

int ta_cross_entropy(const void* logits, int logits_bf16, long ldl, const int* rows, const long* targets, int n, int V,
                     float scale, float* nll, float* loss_accum, void* dlogits_bf16, long ldd, hipStream_t st);
int ta_label_rows(const long* labels, int B, int L, int* rows, long* targets, int* n_out, hipStream_t st);

/* optimizer: clip_grad_norm_(max_norm) + AdamW on fp32 masters (configs/training/production.yaml:5-9) */
/* accum[0] += sum(g^2), DETERMINISTICALLY (two launches: per-block partials into scratch, then one block adds them in index order):
 * data-parallel ranks that hold the same all-reduced gradient must compute bit-identical clip coefficients, or their replicas
 * drift apart.  scratch: float[TA_SQNORM_SCRATCH_FLOATS], no initial contents required. */
#define TA_SQNORM_SCRATCH_FLOATS 1024
int ta_grad_sqnorm(const float* g, long n, float* accum, float* scratch, hipStream_t st);
int ta_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, const float* sqnorm, float max_norm, float grad_scale,
                  const float* denom, hipStream_t st);
/* the same update for a flat buffer of nseg parameter tensors in ONE launch (scripts/train.py:406-432: per-group learning rate and
 * weight decay): segment s ends at element seg_end[s] (device long[nseg], multiples of 4, seg_end[nseg-1] == n), has learning
 * rate seg_lr[s] * lr_mult and weight decay seg_wd[s] (device float[nseg]).  Bit-identical to ta_adamw_step per segment. */
int ta_adamw_step_multi(float* p, const float* g, float* m, float* v, long n, const long* seg_end, const float* seg_lr,
                        const float* seg_wd, int nseg, float lr_mult, float beta1, float beta2, float eps, int step,
                        const float* sqnorm, float max_norm, float grad_scale, const float* denom, hipStream_t st);

#ifdef __cplusplus
}
#endif
#endif /* TA355_H */
