"""Thin torch-tensor wrappers over the primitive kernels of libta355.so.

PyTorch is plumbing here (device memory, streams); all arithmetic happens in the HIP library.
Every wrapper launches on ``torch.cuda.current_stream()`` and raises ``Ta355Error`` on failure.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, lib

BF16 = torch.bfloat16
F32 = torch.float32


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    if _lib.DRY_RUN:          # test instrumentation only (see _lib.DRY_RUN)
        return None
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, dtype=None):
    assert (t.is_cuda or _lib.DRY_RUN) and t.is_contiguous(), "ta355 ops need contiguous CUDA(HIP) tensors"
    if dtype is not None:
        assert t.dtype == dtype, (t.dtype, dtype)
    return t


# The library reads its environment knobs once (csrc/gemm.hip GemmKnobs::load, the decode switch of generate.hip); tests and A/B
# scripts switch them between launches, so the wrapper calls ta_gemm_reload_knobs() whenever one has changed since the previous call.
# These three are all the library has (plus TA355_ENC_QKV_FUSED, read per call).
_KNOB_KEYS = ("TA355_GEMM_VARIANT", "TA355_GELU_LUT", "TA355_DECODE_FUSED")
_knob_state = None


def sync_gemm_knobs():
    global _knob_state
    import os
    cur = tuple(os.environ.get(k) for k in _KNOB_KEYS)
    if cur != _knob_state:
        if _knob_state is not None or any(v is not None for v in cur):
            lib().ta_gemm_reload_knobs()
        _knob_state = cur


def gemm_nt(A, W, M=None, N=None, K=None, *, out=None, out_dtype=BF16, bias=None, residual=None, act=0,
            a_map=None, c_map=None, splits=1, k_ext=None, residual_bf16=None, rope=None, w_blocked=False):
    """C[M,N] = epilogue(A[M,K] @ W[N,K]^T).  a_map=(ld, rpb, batch_stride); c_map=(ld, rpb, batch_stride, offset);
    k_ext=(A2 [M,K2], W2 [N,K2]) adds A2 @ W2^T inside the same launch (the LoRA rank-space tile);
    rope=(table f32 [rows,16,2], rows) with act=2: interleaved partial rotary embedding in the epilogue (ta355.h)."""
    _req(A, BF16); _req(W, BF16)
    sync_gemm_knobs()
    opts = None
    if residual_bf16 is not None or k_ext is not None or rope is not None or w_blocked:
        from ._lib import GemmOpts
        opts = GemmOpts()
        opts.w_blocked = int(bool(w_blocked))      # W given as [N/64][K/64][64][64] blocks (pass N and K explicitly)
        if rope is not None:
            _req(rope[0], F32)
            opts.rope_tab, opts.rope_rows = ptr(rope[0]), int(rope[1])
            opts.rope_cols = int(rope[2]) if len(rope) > 2 else 0       # rope on columns [0, rope_cols) only (0 = all)
        if residual_bf16 is not None:       # bf16 residual with C's row map (may alias `out`)
            _req(residual_bf16, BF16)
            opts.residual_bf16 = ptr(residual_bf16)
        if k_ext is not None:
            A2, W2 = k_ext
            _req(A2, BF16); _req(W2, BF16)
            opts.a2, opts.w2, opts.k2, opts.lda2 = ptr(A2), ptr(W2), W2.shape[1], A2.shape[1]
    N = N or W.shape[0]
    K = K or W.shape[1]
    M = M or A.numel() // K
    lda, a_rpb, a_bs = a_map or (K, 0, 0)
    ldc, c_rpb, c_bs, c_off = c_map or (N, 0, 0, 0)
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=out_dtype)
    ws = None
    if splits > 1:
        ws = torch.empty(lib().ta_gemm_splitk_ws_bytes(M, N, splits) // 4, device=A.device, dtype=F32)
    import ctypes as _C
    check(lib().ta_gemm_bf16_nt_opt(ptr(A), ptr(W), ptr(out), M, N, K, lda, a_rpb, a_bs, ldc, c_rpb, c_bs, c_off,
                                    ptr(bias), ptr(residual), act, 1 if out.dtype == BF16 else 0, splits, ptr(ws),
                                    None, None, None, None if opts is None else _C.addressof(opts), stream()), "ta_gemm_bf16_nt_opt")
    return out


def gemm_tn(Y, X, out=None, accumulate=False):
    """out f32 [Ny, Nx] (+)= Y[M, Ny]^T @ X[M, Nx]  (bf16 operands; the weight-gradient product dW = dY^T X)."""
    _req(Y, BF16); _req(X, BF16)
    M, Ny = Y.shape
    Nx = X.shape[1]
    if out is None:
        out = torch.zeros((Ny, Nx), device=Y.device, dtype=F32)
        accumulate = False
    ws = torch.empty(max(lib().ta_gemm_bf16_tn_ws_bytes(M, Ny, Nx), 16), device=Y.device, dtype=torch.uint8)
    check(lib().ta_gemm_bf16_tn(ptr(Y), ptr(X), ptr(out), M, Ny, Nx, int(bool(accumulate)), ptr(ws), ws.numel(), stream()),
          "ta_gemm_bf16_tn")
    return out


def gemm_nt_grouped(A, W, out, M, N, K, *, seg=None, krange=None, n_groups=1, bias=None, act=0, a_idx=None, w_stride=0, c_stride=0):
    """ta_gemm_bf16_nt_grouped: every group (MoE expert) in ONE launch.  rows form: ``seg`` int32 [2 * n] {row base, count},
    W [n, N, K] (w_stride = N * K), bias [n, N]; K-slice form: ``krange`` int32 [2 * n] 64-wide K-tile ranges, out f32
    [n, M, N] (c_stride = M * N)."""
    _req(A, BF16); _req(W, BF16)
    sync_gemm_knobs()
    check(lib().ta_gemm_bf16_nt_grouped(ptr(A), ptr(W), ptr(out), M, N, K, ptr(bias), act, int(out.dtype == BF16), ptr(a_idx),
                                        ptr(seg), ptr(krange), n_groups, w_stride, c_stride, stream()), "ta_gemm_bf16_nt_grouped")
    return out


def layernorm(x, w, b, eps=1e-5, rowscale=None, out_bf16=True, out_f32=False):
    if x.dtype == BF16:
        M, H = x.shape
        yb = torch.empty((M, H), device=x.device, dtype=BF16) if out_bf16 else None
        yf = torch.empty((M, H), device=x.device, dtype=F32) if out_f32 else None
        check(lib().ta_layernorm_bf16(ptr(x), ptr(w), ptr(b), ptr(yb), ptr(yf), ptr(rowscale), M, H, eps, stream()),
              "ta_layernorm_bf16")
        return yb, yf
    _req(x, F32)
    M, H = x.shape
    yb = torch.empty((M, H), device=x.device, dtype=BF16) if out_bf16 else None
    yf = torch.empty((M, H), device=x.device, dtype=F32) if out_f32 else None
    check(lib().ta_layernorm_f32(ptr(x), ptr(w), ptr(b), ptr(yb), ptr(yf), ptr(rowscale), M, H, eps, stream()),
          "ta_layernorm_f32")
    return yb, yf


def rmsnorm_fwd(x, w, eps=1e-6, act_gelu=False, out_bf16=True, out_f32=False):
    M, H = x.shape
    yb = torch.empty((M, H), device=x.device, dtype=BF16) if out_bf16 else None
    yf = torch.empty((M, H), device=x.device, dtype=F32) if out_f32 else None
    r = torch.empty(M, device=x.device, dtype=F32)
    if x.dtype == BF16:                  # residual stream kept in the model dtype
        assert not act_gelu
        check(lib().ta_rmsnorm_fwd_bf16(ptr(x), ptr(w), ptr(yb), ptr(yf), ptr(r), M, H, eps, stream()), "ta_rmsnorm_fwd_bf16")
        return yb, yf, r
    _req(x, F32)
    check(lib().ta_rmsnorm_fwd(ptr(x), ptr(w), ptr(yb), ptr(yf), ptr(r), M, H, eps, int(act_gelu), stream()),
          "ta_rmsnorm_fwd")
    return yb, yf, r


def rmsnorm_bwd(dy, x, rstd, w, dres=None, act_gelu=False, want_dw=False, out_bf16=True):
    if x.dtype != BF16:
        _req(dy, F32)
    M, H = x.shape
    dx = torch.empty((M, H), device=x.device, dtype=F32)
    dxb = torch.empty((M, H), device=x.device, dtype=BF16) if out_bf16 else None
    if x.dtype == BF16:
        assert not act_gelu and not want_dw
        check(lib().ta_rmsnorm_bwd_bf16(ptr(dy), int(dy.dtype == BF16), ptr(x), ptr(rstd), ptr(w), ptr(dres), ptr(dx), ptr(dxb), M, H, stream()),
              "ta_rmsnorm_bwd_bf16")
        return dx, dxb, None
    _req(x, F32)
    dw = torch.zeros(H, device=x.device, dtype=F32) if want_dw else None
    check(lib().ta_rmsnorm_bwd(ptr(dy), ptr(x), ptr(rstd), ptr(w), ptr(dres), ptr(dx), ptr(dxb), ptr(dw), M, H,
                               int(act_gelu), stream()), "ta_rmsnorm_bwd")
    return dx, dxb, dw


def pad64(n):
    return (n + 63) // 64 * 64


def attention_fwd(Q, K, VT, L, causal, scale, kmask=None, want_lse=True):
    """Q [B,Hq,L,hd], K [B,Hkv,L,hd], VT [B,Hkv,hd,Lp] bf16 -> O [B*L, Hq*hd] bf16, LSE [B,Hq,L]."""
    B, Hq, _, hd = Q.shape
    Hkv, Lp = K.shape[1], VT.shape[3]
    O = torch.empty((B * L, Hq * hd), device=Q.device, dtype=BF16)
    lse = torch.empty((B, Hq, L), device=Q.device, dtype=F32) if want_lse else None
    check(lib().ta_attention_fwd(ptr(Q), ptr(K), ptr(VT), ptr(O), ptr(lse), ptr(kmask), B, Hq, Hkv, L, Lp, hd,
                                 int(causal), scale, stream()), "ta_attention_fwd")
    return O, lse


def attention_fwd_strided(Q, K, VT, B, Hq, Hkv, L, hd, causal, scale, layout, kmask=None):
    """ta_attention_fwd_ex: operands described by element strides (q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs)."""
    import ctypes as _C
    from ._lib import AttnLayout
    lay = AttnLayout(*[int(v) for v in layout])
    O = torch.empty((B * L, Hq * hd), device=Q.device, dtype=BF16)
    check(lib().ta_attention_fwd_ex(ptr(Q), ptr(K), ptr(VT), ptr(O), None, ptr(kmask), B, Hq, Hkv, L, pad64(L), hd,
                                    int(causal), scale, _C.addressof(lay), stream()), "ta_attention_fwd_ex")
    return O


def attention_bwd(Q, QT, K, KT, V, dO, dOT, lse, delta, L, causal, scale, kmask=None):
    B, Hq, _, hd = Q.shape
    Hkv, Lp = K.shape[1], KT.shape[3]
    dQ = torch.empty_like(Q); dK = torch.empty_like(K); dV = torch.empty_like(V)
    check(lib().ta_attention_bwd(ptr(Q), ptr(QT), ptr(K), ptr(KT), ptr(V), ptr(dO), dO.shape[-1], ptr(dOT), ptr(lse),
                                 ptr(delta), ptr(kmask), ptr(dQ), ptr(dK), ptr(dV), B, Hq, Hkv, L, Lp, hd, int(causal),
                                 scale, stream()), "ta_attention_bwd")
    return dQ, dK, dV


def attention_enc_fwd(qkv, B, heads, S):
    """GLM-ASR encoder attention straight from the q|k|v GEMM output: qkv bf16 [B*S, 3*heads*64] (q pre-scaled by
    head_dim^-0.5 * log2 e) -> bf16 [B*S, heads*64] = softmax_base2(q k^T) v, non-causal, no mask."""
    _req(qkv, BF16)
    assert qkv.shape == (B * S, 3 * heads * 64)
    out = torch.empty((B * S, heads * 64), device=qkv.device, dtype=BF16)
    check(lib().ta_attention_enc_fwd(ptr(qkv), ptr(out), B, heads, S, stream()), "ta_attention_enc_fwd")
    return out


def attention_fwd_qkv(qkv0, qn_w, kn_w, cosT, sinT, B, Hq, Hkv, L, scale, eps=1e-6, kmask=None, pos=None):
    """ta_lm_qkv_post_fwd + ta_attention_fwd in one launch (short causal sequences): returns O, LSE, Q, K, V, rq, rk."""
    hd, dev = 128, qkv0.device
    mk = lambda h: torch.empty((B, h, L, hd), device=dev, dtype=BF16)
    Q, K, V = mk(Hq), mk(Hkv), mk(Hkv)
    rq = torch.empty((B * L, Hq), device=dev, dtype=F32); rk = torch.empty((B * L, Hkv), device=dev, dtype=F32)
    O = torch.empty((B * L, Hq * hd), device=dev, dtype=BF16)
    lse = torch.empty((B, Hq, L), device=dev, dtype=F32)
    check(lib().ta_attention_fwd_qkv(ptr(qkv0), ptr(qn_w), ptr(kn_w), ptr(cosT), ptr(sinT), ptr(pos), ptr(Q), ptr(K), ptr(V), ptr(rq),
                                     ptr(rk), ptr(O), ptr(lse), ptr(kmask), B, Hq, Hkv, L, scale, eps, stream()), "ta_attention_fwd_qkv")
    return O, lse, Q, K, V, rq, rk


def attention_bwd_qkv(Q, K, V, dO, lse, delta, qkv0, rq, rk, qn_w, kn_w, cosT, sinT, L, scale, kmask=None, pos=None):
    """Causal GQA attention backward with the q|k|v post-processing backward in its epilogue: returns d(qkv0) token-major."""
    B, Hq, _, hd = Q.shape
    Hkv, Lp = K.shape[1], pad64(L)
    dqkv = torch.empty_like(qkv0)
    check(lib().ta_attention_bwd_qkv(ptr(Q), ptr(K), ptr(V), ptr(dO), dO.shape[-1], ptr(lse), ptr(delta), ptr(kmask), ptr(qkv0),
                                     ptr(rq), ptr(rk), ptr(qn_w), ptr(kn_w), ptr(cosT), ptr(sinT), ptr(pos), ptr(dqkv), B, Hq, Hkv,
                                     L, Lp, hd, 1, scale, stream()), "ta_attention_bwd_qkv")
    return dqkv


def lm_qkv_post_fwd(qkv0, qn_w, kn_w, cosT, sinT, B, Hq, Hkv, L, eps=1e-6, pos=None):
    hd, Lp, dev = 128, pad64(L), qkv0.device
    mk = lambda h: torch.empty((B, h, L, hd), device=dev, dtype=BF16)
    mkT = lambda h: torch.empty((B, h, hd, Lp), device=dev, dtype=BF16)
    Q, K, V, QT, KT, VT = mk(Hq), mk(Hkv), mk(Hkv), mkT(Hq), mkT(Hkv), mkT(Hkv)
    rq = torch.empty((B * L, Hq), device=dev, dtype=F32); rk = torch.empty((B * L, Hkv), device=dev, dtype=F32)
    check(lib().ta_lm_qkv_post_fwd(ptr(qkv0), ptr(qn_w), ptr(kn_w), ptr(cosT), ptr(sinT), ptr(pos), ptr(Q), ptr(K),
                                   ptr(V), ptr(QT), ptr(KT), ptr(VT), ptr(rq), ptr(rk), B, Hq, Hkv, L, Lp, eps,
                                   stream()), "ta_lm_qkv_post_fwd")
    return Q, K, V, QT, KT, VT, rq, rk


def lm_qkv_post_bwd(dQ, dK, dV, qkv0, rq, rk, qn_w, kn_w, cosT, sinT, B, Hq, Hkv, L, pos=None, dqn=None, dkn=None):
    """dqn / dkn f32 [128]: += the q_norm / k_norm weight gradients (trainable LM)."""
    dqkv = torch.empty_like(qkv0)
    check(lib().ta_lm_qkv_post_bwd(ptr(dQ), ptr(dK), ptr(dV), ptr(qkv0), ptr(rq), ptr(rk), ptr(qn_w), ptr(kn_w),
                                   ptr(cosT), ptr(sinT), ptr(pos), ptr(dqkv), ptr(dqn), ptr(dkn), B, Hq, Hkv, L, stream()),
          "ta_lm_qkv_post_bwd")
    return dqkv


def rmsnorm_dw(dy, x, rstd, dw):
    """dw[h] += sum_m dy[m,h] * x[m,h] * rstd[m]   (dy, x: f32 or bf16 [M, H])"""
    M, H = x.shape
    check(lib().ta_rmsnorm_dw(ptr(dy), int(dy.dtype == BF16), ptr(x), int(x.dtype == BF16), ptr(rstd), ptr(dw), M, H, stream()),
          "ta_rmsnorm_dw")
    return dw


def embed_grad_scatter(ids, src_row, dx0, dembed):
    n, D = dx0.shape
    check(lib().ta_embed_grad_scatter(ptr(ids), ptr(src_row), ptr(dx0), ptr(dembed), n, D, dembed.shape[0], stream()),
          "ta_embed_grad_scatter")
    return dembed


def enc_qkv_post(qkv, cosT, sinT, B, H, S):
    Sp, dev = pad64(S), qkv.device
    Q = torch.empty((B, H, S, 64), device=dev, dtype=BF16); K = torch.empty_like(Q)
    VT = torch.empty((B, H, 64, Sp), device=dev, dtype=BF16)
    check(lib().ta_enc_qkv_post(ptr(qkv), ptr(cosT), ptr(sinT), ptr(Q), ptr(K), ptr(VT), B, H, S, Sp, stream()),
          "ta_enc_qkv_post")
    return Q, K, VT


def attn_bwd_prep(dO, O, B, Hq, L):
    Lp, dev = pad64(L), dO.device
    delta = torch.empty((B, Hq, L), device=dev, dtype=F32)
    dOT = torch.empty((B, Hq, 128, Lp), device=dev, dtype=BF16)
    check(lib().ta_attn_bwd_prep(ptr(dO), ptr(O), ptr(delta), ptr(dOT), B, Hq, L, Lp, stream()), "ta_attn_bwd_prep")
    return delta, dOT


def swiglu_fwd(gu, F):
    M = gu.shape[0]
    act = torch.empty((M, F), device=gu.device, dtype=BF16)
    check(lib().ta_swiglu_fwd(ptr(gu), ptr(act), M, F, stream()), "ta_swiglu_fwd")
    return act


def swiglu_bwd(dact, gu, F):
    dgu = torch.empty_like(gu)
    check(lib().ta_swiglu_bwd(ptr(dact), ptr(gu), ptr(dgu), gu.shape[0], F, stream()), "ta_swiglu_bwd")
    return dgu


def cast_bf16(x, out=None):
    _req(x, F32)
    y = torch.empty(x.shape, device=x.device, dtype=BF16) if out is None else out
    check(lib().ta_cast_f32_bf16(ptr(x), ptr(y), x.numel(), stream()), "ta_cast_f32_bf16")
    return y


def transpose_to_bf16(x, ld_out=None, in_map=None, rows=None, cols=None, out=None):
    """x [R, C] (f32 or bf16) -> bf16 [C, ld_out] (zero padded columns).  With ``in_map`` = (ld_in, batch_stride,
    rows_per_batch) the logical [rows, cols] matrix is an affine view of ``x`` (e.g. the im2col view of a padded
    time-major buffer: overlapping rows)."""
    R, Cc = (rows, cols) if rows is not None else x.shape
    ld_out = ld_out or R
    ld_in, in_bs, in_rpb = in_map or (Cc, 0, 0)
    if out is None:
        out = torch.empty((Cc, ld_out), device=x.device, dtype=BF16)
    check(lib().ta_transpose_to_bf16(ptr(x), int(x.dtype == F32), ld_in, in_bs, in_rpb, ptr(out), ld_out, R, Cc,
                                     stream()), "ta_transpose_to_bf16")
    return out


def audio_index(ids, counts, N, audio_id):
    B, L = ids.shape
    src = torch.empty(B * L, device=ids.device, dtype=torch.int32)
    check(lib().ta_audio_index(ptr(ids), ptr(counts), ptr(src), B, L, N, audio_id, stream()), "ta_audio_index")
    return src


def label_rows(labels):
    B, L = labels.shape
    rows = torch.empty(B * L, device=labels.device, dtype=torch.int32)
    tg = torch.empty(B * L, device=labels.device, dtype=torch.int64)
    n = torch.zeros(1, device=labels.device, dtype=torch.int32)
    check(lib().ta_label_rows(ptr(labels), B, L, ptr(rows), ptr(tg), ptr(n), stream()), "ta_label_rows")
    return rows, tg, n


def cross_entropy(logits, targets, V, scale, rows=None, want_dlogits=True, ldd=None):
    n = targets.shape[0]
    ldl = logits.shape[1]
    ldd = ldd or ldl
    nll = torch.empty(n, device=logits.device, dtype=F32)
    loss = torch.zeros(1, device=logits.device, dtype=F32)
    dl = torch.empty((n, ldd), device=logits.device, dtype=BF16) if want_dlogits else None
    check(lib().ta_cross_entropy(ptr(logits), int(logits.dtype == BF16), ldl, ptr(rows), ptr(targets), n, V, scale,
                                 ptr(nll), ptr(loss), ptr(dl), ldd, stream()), "ta_cross_entropy")
    return loss, nll, dl


def bernoulli_keep(n, keep_prob, seed, device):
    keep = torch.empty(n, device=device, dtype=F32)
    check(lib().ta_bernoulli_keep(ptr(keep), n, keep_prob, seed, stream()), "ta_bernoulli_keep")
    return keep


def grad_sqnorm(g, accum, scratch=None):
    """accum[0] += sum(g ** 2), bit-reproducibly (``scratch``: f32 [1024], allocated here when not given)."""
    if scratch is None:
        scratch = torch.empty(1024, device=g.device, dtype=F32)
    check(lib().ta_grad_sqnorm(ptr(g), g.numel(), ptr(accum), ptr(scratch), stream()), "ta_grad_sqnorm")


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, sqnorm=None, max_norm=0.0, grad_scale=1.0, denom=None):
    check(lib().ta_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, wd, step, ptr(sqnorm),
                              max_norm, grad_scale, ptr(denom), stream()), "ta_adamw_step")


def adamw_step_multi(p, g, m, v, seg_end, seg_lr, seg_wd, lr_mult, beta1, beta2, eps, step, sqnorm=None, max_norm=0.0, grad_scale=1.0,
                     denom=None):
    """One launch over a flat buffer of parameter segments (seg_end int64, seg_lr / seg_wd f32, all on the device)."""
    check(lib().ta_adamw_step_multi(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(seg_end), ptr(seg_lr), ptr(seg_wd), seg_end.numel(),
                                    lr_mult, beta1, beta2, eps, step, ptr(sqnorm), max_norm, grad_scale, ptr(denom), stream()),
          "ta_adamw_step_multi")


# ----------------------------------------------------------------------------- trainable-projector primitives (nn_prims.hip)
def gelu_fwd(h):
    _req(h, BF16)
    a = torch.empty_like(h)
    check(lib().ta_gelu_fwd(ptr(h), ptr(a), h.numel(), stream()), "ta_gelu_fwd")
    return a


def gelu_bwd(da, h):
    _req(da, BF16); _req(h, BF16)
    dh = torch.empty_like(h)
    check(lib().ta_gelu_bwd(ptr(da), ptr(h), ptr(dh), h.numel(), stream()), "ta_gelu_bwd")
    return dh


def colsum(x):
    """[R, C] (f32 or bf16) -> f32 [C]"""
    R, Cc = x.shape
    out = torch.empty(Cc, device=x.device, dtype=F32)
    check(lib().ta_colsum(ptr(x), int(x.dtype == F32), R, Cc, ptr(out), stream()), "ta_colsum")
    return out


def layernorm_res_fwd(z, gamma, beta, eps, res=None, keep=None, res_rows=0):
    """-> (y f32, y bf16, xhat, rstd)"""
    _req(z, F32)
    M, H = z.shape
    yf = torch.empty_like(z)
    yb = torch.empty((M, H), device=z.device, dtype=BF16)
    xhat, rstd = torch.empty_like(z), torch.empty(M, device=z.device, dtype=F32)
    check(lib().ta_layernorm_res_fwd(ptr(z), ptr(keep), ptr(res), res_rows, ptr(gamma), ptr(beta), eps, ptr(xhat), ptr(rstd),
                                     ptr(yf), ptr(yb), M, H, stream()), "ta_layernorm_res_fwd")
    return yf, yb, xhat, rstd


def layernorm_bwd(dy, xhat, rstd, gamma, dgamma, dbeta, keep=None, want_du=True, want_dz=True):
    """-> (du f32 or None, dz bf16 or None); dgamma / dbeta accumulate."""
    _req(dy, F32)
    M, H = dy.shape
    du = torch.empty_like(dy) if want_du else None
    dz = torch.empty((M, H), device=dy.device, dtype=BF16) if want_dz else None
    check(lib().ta_layernorm_bwd(ptr(dy), ptr(xhat), ptr(rstd), ptr(gamma), ptr(keep), ptr(du), ptr(dz), ptr(dgamma),
                                 ptr(dbeta), M, H, stream()), "ta_layernorm_bwd")
    return du, dz


def attn_small_fwd(Q, K, V, EB, heads, Lq, Lk, scale, keep=None):
    _req(Q, BF16); _req(K, BF16); _req(V, BF16)
    H = Q.shape[-1]
    P = torch.empty((EB, heads, Lq, Lk), device=Q.device, dtype=F32)
    O = torch.empty((EB * Lq, H), device=Q.device, dtype=BF16)
    check(lib().ta_attn_small_fwd(ptr(Q), ptr(K), ptr(V), EB, heads, H // heads, Lq, Lk, scale, ptr(keep), ptr(P), ptr(O),
                                  stream()), "ta_attn_small_fwd")
    return O, P


def attn_small_bwd(dO, Q, K, V, P, EB, heads, Lq, Lk, scale, keep=None):
    _req(dO, BF16)
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    H = Q.shape[-1]
    check(lib().ta_attn_small_bwd(ptr(dO), ptr(Q), ptr(K), ptr(V), ptr(P), ptr(keep), scale, ptr(dQ), ptr(dK), ptr(dV), EB,
                                  heads, H // heads, Lq, Lk, stream()), "ta_attn_small_bwd")
    return dQ, dK, dV


def relu_fwd(h):
    _req(h, BF16)
    a = torch.empty_like(h)
    check(lib().ta_relu_fwd(ptr(h), ptr(a), h.numel(), stream()), "ta_relu_fwd")
    return a


def relu_bwd(da, h):
    _req(da, BF16); _req(h, BF16)
    dh = torch.empty_like(h)
    check(lib().ta_relu_bwd(ptr(da), ptr(h), ptr(dh), h.numel(), stream()), "ta_relu_bwd")
    return dh


def mix_fwd(logits, o):
    """logits f32 [M, E], o f32 [E, M, D] -> (rw [M, E], out [M, D])"""
    _req(logits, F32); _req(o, F32)
    E, M, D = o.shape
    rw = torch.empty((M, E), device=o.device, dtype=F32)
    out = torch.empty((M, D), device=o.device, dtype=F32)
    check(lib().ta_mix_fwd(ptr(logits), ptr(o), ptr(rw), ptr(out), M, D, E, stream()), "ta_mix_fwd")
    return rw, out


def mix_bwd(dout, o, rw):
    """-> (do bf16 [E, M, D], dlogits f32 [M, E])"""
    _req(dout, F32)
    E, M, D = o.shape
    dob = torch.empty((E, M, D), device=o.device, dtype=BF16)
    dlg = torch.empty((M, E), device=o.device, dtype=F32)
    check(lib().ta_mix_bwd(ptr(dout), ptr(o), ptr(rw), ptr(dob), ptr(dlg), M, D, E, stream()), "ta_mix_bwd")
    return dob, dlg
