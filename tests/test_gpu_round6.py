"""-m gpu, round 6: the kernels and paths that are new this round, against a torch fp32 reference / the oracle.

* ``layernorm_f32x8_kernel``: the encoder's LayerNorm over an fp32 residual stream (the fp32-stream mode = the training recipe);
* the fp32-stream LM backward with bf16 gradients into the RMSNorm backward (``ta_i_rmsnorm_bwd_dyb``) against the oracle's fp32
  backward at reduced depth, tighter than the bf16-stream mode;
* the fp32-residual GEMM epilogues with batched residual loads (pairs of strips / one batch), every tile variant.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
BF16, F32 = torch.bfloat16, torch.float32


def rnd(*shape, seed=0, scale=1.0, dtype=F32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def relerr(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("M", [5, 517, 4100, 16000])          # one / two / four rows per half wave (ta_layernorm_f32's choice), ragged tails
@pytest.mark.parametrize("H", [256, 1280, 2048])
def test_layernorm_f32_stream_to_bf16(M, H):
    from tiny_audio_amd import ops
    x = rnd(M, H, seed=1, scale=3.0) + 0.5
    w, b = 1 + 0.1 * rnd(H, seed=2), 0.1 * rnd(H, seed=3)
    keep = (torch.arange(M, device=DEV) % 3 != 0).float()
    ref = torch.nn.functional.layer_norm(x, (H,), w, b, 1e-5)
    yb, yf = ops.layernorm(x, w, b, 1e-5, out_bf16=True, out_f32=False)           # the half-wave-per-row kernel
    assert yf is None and yb.dtype == BF16
    assert relerr(yb, ref) < 8e-3
    gb, gf = ops.layernorm(x, w, b, 1e-5, out_bf16=True, out_f32=True)            # the generic kernel: same arithmetic, another summation order
    assert float((yb != gb).float().mean()) < 2e-3 and relerr(yb, gb) < 8e-3
    yk, _ = ops.layernorm(x, w, b, 1e-5, rowscale=keep, out_bf16=True, out_f32=False)
    assert relerr(yk, ref * keep[:, None]) < 8e-3 and float(yk[::3].abs().max()) == 0.0


@pytest.mark.parametrize("M,H", [(7, 256), (301, 1024), (6144, 1024), (333, 2048)])
def test_rmsnorm_f32_stream_to_bf16(M, H):
    """rmsnorm_fwd_f32x8_kernel (f32 -> bf16 only) against torch and against the generic wave-per-row kernel."""
    from tiny_audio_amd import ops
    x, w = rnd(M, H, seed=1, scale=2.0), 1 + 0.1 * rnd(H, seed=2)
    r = torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-6)
    ref = w * (x * r)
    yb, yf, rstd = ops.rmsnorm_fwd(x, w, 1e-6, out_bf16=True, out_f32=False)
    assert yf is None and relerr(yb, ref) < 8e-3 and relerr(rstd, r.flatten()) < 1e-5
    gb, gf, grstd = ops.rmsnorm_fwd(x, w, 1e-6, out_bf16=True, out_f32=True)     # generic kernel
    assert float((yb != gb).float().mean()) < 2e-3 and relerr(rstd, grstd) < 1e-6


@pytest.mark.parametrize("M,H", [(5, 256), (301, 1024), (6144, 1024), (77, 2048), (40, 1280)])
@pytest.mark.parametrize("with_res", [True, False])
def test_rmsnorm_bwd_half_wave_forms(M, H, with_res):
    """Both forms the LM backward runs (the all-bf16 one on rmsnorm_bwd_bf16x8_kernel): bf16 everything with the d(x) stream updated IN PLACE
    (ta_rmsnorm_bwd_bf16s) and fp32 stream + bf16 incoming gradient (ta_rmsnorm_bwd_dyb), against torch autograd on the same inputs."""
    from tiny_audio_amd import _lib
    from tiny_audio_amd.ops import ptr, stream
    L_ = _lib.lib()
    x32 = rnd(M, H, seed=1, scale=2.0)
    w = 1 + 0.1 * rnd(H, seed=2)
    dy = rnd(M, H, seed=3).to(BF16)
    dres32 = rnd(M, H, seed=4)
    for f32 in (True, False):
        x = x32 if f32 else x32.to(BF16)
        dres = (dres32 if f32 else dres32.to(BF16)) if with_res else None
        xr = x.float().clone().requires_grad_(True)
        r = torch.rsqrt((xr * xr).mean(-1, keepdim=True) + 1e-6)
        (w * (xr * r)).backward(dy.float())
        ref = xr.grad + (dres.float() if with_res else 0.0)
        rstd = r.detach().flatten().contiguous()
        dxf = torch.empty(M, H, device=DEV, dtype=F32)
        if f32:
            dxb = torch.empty(M, H, device=DEV, dtype=BF16)
            _lib.check(L_.ta_rmsnorm_bwd_dyb(ptr(dy), ptr(x), ptr(rstd), ptr(w), ptr(dres), ptr(dxf), ptr(dxb), M, H, stream()))
        else:
            dxb = dres.clone() if with_res else torch.empty(M, H, device=DEV, dtype=BF16)      # in place: dres IS the output image
            _lib.check(L_.ta_rmsnorm_bwd_bf16s(ptr(dy), 1, ptr(x), ptr(rstd), ptr(w), ptr(dxb) if with_res else None, ptr(dxf), ptr(dxb), M, H, stream()))
        assert relerr(dxf, ref) < 2e-5, (f32, relerr(dxf, ref))
        assert relerr(dxb, ref) < 8e-3


@pytest.mark.parametrize("variant", [None, "0", "3", "4", "5", "10", "12"])
def test_gemm_f32_residual_in_place_every_variant(monkeypatch, variant):
    """x_f32 += A W^T + bias, in place (the fp32-stream residual GEMMs): the epilogues that batch the residual loads -- pairs of
    strips in the persistent 8-wave kernel, one batch in the one-wave-per-SIMD kernel -- and the strip-by-strip ones agree with fp32
    torch on a shape with partial tiles in both directions."""
    from tiny_audio_amd import ops
    if variant is not None:
        monkeypatch.setenv("TA355_GEMM_VARIANT", variant)
    for (M, N, K) in ((1000, 1280, 1280), (6144, 1024, 2048), (333, 200, 192)):
        A, W = rnd(M, K, seed=1, dtype=BF16), rnd(N, K, seed=2, scale=K ** -0.5, dtype=BF16)
        bias, x = rnd(N, seed=3), rnd(M, N, seed=4, scale=2.0)
        ref = x + A.float() @ W.float().T + bias
        out = x.clone()
        ops.gemm_nt(A, W, M, N, K, out=out, bias=bias, residual=out)
        assert relerr(out, ref) < 2e-5 * max(1.0, K / 256), (variant, M, N, K, relerr(out, ref))
        out2 = torch.empty_like(x)                                             # residual and output in different buffers (the LM's x1 = x + ...)
        ops.gemm_nt(A, W, M, N, K, out=out2, bias=bias, residual=x)
        assert torch.equal(out2, out)


def test_lm_backward_f32_streams_bf16_dy_vs_oracle():
    """The fp32-stream mode (model_dtype float32 = the recipe) at reduced depth against the fp32 oracle: loss and projector gradients.
    The gradient of every RMSNorm output travels as bf16 there since round 6 (what autograd hands back under bf16 autocast); the
    bound is the one the f32-stream tests of round 5 used."""
    from oracle import model as OM
    from oracle import weights as OW
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
    enc, lm = OW.enc_config(layers=2), OW.lm_config(vocab=2051, layers=3)
    AID, PAD, EOS = 2050, 2040, 2041
    wE, wL, wP = OW.init_encoder(enc, 0), OW.init_lm(lm, 1), OW.init_mlp_projector(1280, 1024, 1024)
    res = {}
    for dt in ("float32", "bfloat16"):
        cfg = ASRConfig(audio_config=enc, text_config=lm, projector_hidden_dim=1024, audio_token_id=AID, model_dtype=dt, audio_token_dropout=0.0)
        model = ASRModel(cfg, device=DEV, init="none")
        model.audio_tower.load_state_dict_hf(wE)
        model.language_model.load_state_dict_hf(wL)
        model.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in wP.items()})
        f = LogMelFeatureExtractor(128, DEV)([OW.synthetic_wave(0, 32000), OW.synthetic_wave(1, 24000)], sampling_rate=16000)
        feats = f["input_features"]
        mel = f["attention_mask"].sum(-1).cpu().numpy()
        counts = ((mel - 1) // 2 + 1 - 4) // 4 + 1
        ids, att, lab, counts = OW.synthetic_tokens(2, counts.tolist(), lm["vocab"], AID, PAD, EOS, n_text=10, n_suffix=4, ragged=True)
        model.train()
        out = model(input_ids=torch.from_numpy(ids), input_features=feats, attention_mask=torch.from_numpy(att), labels=torch.from_numpy(lab),
                    audio_token_counts=torch.from_numpy(counts), return_logits=False)
        out.loss.backward()
        torch.cuda.synchronize()
        if "ref" not in res:
            batch = dict(input_ids=ids, attention_mask=att, labels=lab, input_features=feats.cpu().numpy(), audio_token_counts=counts)
            W = dict(encoder=wE, lm=wL, projector=wP)
            ocfg = dict(enc=enc, lm=lm, projector_type="mlp", k=4, audio_token_id=AID)
            ref = OM.asr_forward(batch, W, ocfg, training=True)
            grads, _ = OM.asr_backward(ref, W, ocfg)
            res["ref"] = (float(ref["loss"]), grads)
        rl, grads = res["ref"]
        cos = {}
        for k, p in model.projector.named_parameters():
            a, b = p.grad.float().cpu().numpy().ravel().astype(np.float64), grads[k].ravel().astype(np.float64)
            cos[k] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        res[dt] = (abs(float(out.loss) - rl) / rl, min(cos.values()))
    assert res["float32"][0] < 2e-3 and res["float32"][1] > 0.9995, res
    assert res["bfloat16"][0] < 5e-3 and res["bfloat16"][1] > 0.999, res
