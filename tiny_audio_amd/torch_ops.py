"""``torch.ops.ta355.*``: the composites of libta355.so registered as PyTorch custom operators (torch.library).

north_star: "exposed as PyTorch-ROCm custom ops so ASRModel/ASRProcessor and scripts/train.py stay drop-in".  The
reference has no FFI; its seams are nn.Module boundaries, and each operator below sits on one:

    ta355::logmel              WhisperFeatureExtractor.__call__           scripts/train.py:327-333
    ta355::encoder_forward     model.audio_tower(input_features=...)      tiny_audio/asr_modeling.py:448-450
    ta355::mlp_projector       MLPAudioProjector.forward   (+ autograd)   tiny_audio/projectors.py:57-71
    ta355::moe_projector       MoEAudioProjector.forward   (+ autograd)   tiny_audio/projectors.py:257-347
    ta355::lm_forward_loss     embed + masked_scatter + Qwen3ForCausalLM(labels=...)  (+ autograd: d audio rows, LoRA
                               adapters or the LM's own weights)          tiny_audio/asr_modeling.py:497-526

Operators take tensors and plain scalars only.  The frozen models' packed bf16 weight images live in the Python module
that owns them (they are derived state, rebuilt when a master changes), so every operator carries an integer ``handle``
naming that module (``register_module``); trainable parameters are passed as tensors so that autograd sees them.  Saved
state for backward (the kernels' tapes) is returned as extra outputs, as torch.library requires.  Each forward has a
``*_backward`` operator of its own, registered with ``register_autograd``; fake (meta) implementations give the output
shapes so that the operators trace.

There is no CPU implementation: the implementations call the C ABI, which raises without a GPU (the dry-run marshalling
mode of ``_lib`` used by the CPU test-suite aside).
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from .ops import BF16, F32, ptr, stream

_MODULES: "weakref.WeakValueDictionary[int, object]" = weakref.WeakValueDictionary()
_NEXT = [1]


def register_module(mod) -> int:
    """-> integer handle of a weight-owning module (feature extractor, encoder, projector, LM); idempotent.
    A handle names ONE object: ``copy.deepcopy`` / pickling duplicate the ``_ta355_handle`` attribute with the rest of
    ``__dict__``, so a module that arrives with a handle somebody else holds (an EMA or reference copy of a model) is given a
    fresh one instead of rebinding the original's -- a later backward of the original would otherwise resolve its handle to
    the copy's packed weights."""
    h = mod.__dict__.get("_ta355_handle")
    owner = _MODULES.get(h) if h is not None else None
    if h is None or (owner is not None and owner is not mod):
        h = _NEXT[0]
        _NEXT[0] += 1
        object.__setattr__(mod, "_ta355_handle", h)
    _MODULES[h] = mod
    return h


def module_of(handle: int):
    mod = _MODULES.get(int(handle))
    if mod is None:
        raise _lib.Ta355Error(f"ta355 operator called with a stale module handle ({handle})")
    return mod


def _empty(dev):
    return torch.empty(0, device=dev)


def _is_bf16_image(x) -> bool:
    return x.dtype == BF16 and x.is_contiguous()


def _bf16_image(x):
    """-> (the bf16 tensor the kernels read, what the operator returns as its "bf16 image of x" output).  An operator's outputs
    must not alias its inputs: when x already IS a contiguous bf16 tensor (the frozen encoder's output) the output is an empty
    placeholder and the autograd formula saves the INPUT instead -- no copy of [B, S, E] per step (41 MB at B = 32)."""
    xd = x.detach()
    if _is_bf16_image(xd):
        return xd, torch.empty(0, device=x.device, dtype=BF16)
    xb = xd.to(BF16).contiguous()
    return xb, xb


# ============================================================================ log-mel
@torch.library.custom_op("ta355::logmel", mutates_args=())
def logmel(wav: Tensor, lens: Tensor, handle: int) -> Tuple[Tensor, Tensor]:
    """wav f32 [B, Ls] (zero padded), lens i64 [B] -> (features f32 [B, n_mels, Ls // 160], frame mask i32 [B, Ls // 160])."""
    return module_of(handle)._extract(wav, lens)


@logmel.register_fake
def _(wav, lens, handle):
    B, Ls = wav.shape
    n = module_of(handle).feature_size
    return wav.new_empty((B, n, Ls // 160)), wav.new_empty((B, Ls // 160), dtype=torch.int32)


# ============================================================================ frozen encoder
@torch.library.custom_op("ta355::encoder_forward", mutates_args=())
def encoder_forward(input_features: Tensor, frame_keep: Optional[Tensor], handle: int, return_f32: bool) -> Tensor:
    """f32 [B, n_mels, T] -> last_hidden_state [B, (T-1)//2+1, H] (bf16, or f32 on request); ``frame_keep`` f32 [B*S] is the
    train-time whole-frame dropout mask fused into the final LayerNorm (tiny_audio/asr_modeling.py:458-479)."""
    return module_of(handle)._forward_impl(input_features, frame_keep, return_f32)


@encoder_forward.register_fake
def _(input_features, frame_keep, handle, return_f32):
    enc = module_of(handle)
    B, _, T = input_features.shape
    return input_features.new_empty((B, enc.output_length(T), enc.config.hidden_size), dtype=F32 if return_f32 else BF16)


# ============================================================================ MLP projector
@torch.library.custom_op("ta355::mlp_projector", mutates_args=())
def mlp_projector(x: Tensor, w1: Tensor, g1: Tensor, w2: Tensor, g2: Tensor, handle: int) -> Tuple[Tensor, Tensor, Tensor]:
    """x [B, S, E] -> (y f32 [B, N, D], bf16 image of x, tape).  w1 / g1 / w2 / g2 are the fp32 masters (autograd inputs);
    the kernels read the module's packed bf16 images of them."""
    mod = module_of(handle)
    B, S, _ = x.shape
    xb, xb_out = _bf16_image(x)
    wts = mod._packed_weights()
    L_ = _lib.lib()
    N = mod.get_output_length(S)
    tape = torch.empty(L_.ta_mlp_tape_bytes(C.byref(wts), B, S), device=x.device, dtype=torch.uint8)
    y = torch.empty((B, N, mod.llm_dim), device=x.device, dtype=F32)
    _lib.check(L_.ta_mlp_projector_forward(C.byref(wts), ptr(xb), B, S, ptr(y), ptr(tape), stream()), "ta_mlp_projector_forward")
    return y, xb_out, tape


@mlp_projector.register_fake
def _(x, w1, g1, w2, g2, handle):
    mod = module_of(handle)
    B, S, _ = x.shape
    n_tape = _lib.lib().ta_mlp_tape_bytes(C.byref(mod._packed_weights_meta()), B, S)        # host-only size query
    return (x.new_empty((B, mod.get_output_length(S), mod.llm_dim), dtype=F32),
            x.new_empty((0,) if _is_bf16_image(x) else x.shape, dtype=BF16), x.new_empty((n_tape,), dtype=torch.uint8))


@torch.library.custom_op("ta355::mlp_projector_backward", mutates_args=())
def mlp_projector_backward(dy: Tensor, xb: Tensor, tape: Tensor, handle: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """-> (dW1, dg1, dW2, dg2) f32; no d x (the encoder is frozen)."""
    mod = module_of(handle)
    B, S, _ = xb.shape
    wts = mod._packed_weights()
    L_ = _lib.lib()
    dev = dy.device
    dy = dy.to(F32).contiguous()
    dW1 = torch.empty(mod.linear_1.weight.shape, device=dev, dtype=F32)
    dW2 = torch.empty(mod.linear_2.weight.shape, device=dev, dtype=F32)
    dg1 = torch.empty(mod.norm.weight.shape, device=dev, dtype=F32)
    dg2 = torch.empty(mod.norm_2.weight.shape, device=dev, dtype=F32)
    ws = torch.empty(L_.ta_mlp_bwd_workspace_bytes(C.byref(wts), B, S), device=dev, dtype=torch.uint8)
    _lib.check(L_.ta_mlp_projector_backward(C.byref(wts), ptr(xb), B, S, ptr(dy), ptr(tape), ptr(dW1), ptr(dg1), ptr(dW2),
                                            ptr(dg2), ptr(ws), ws.numel(), stream()), "ta_mlp_projector_backward")
    return dW1, dg1, dW2, dg2


@mlp_projector_backward.register_fake
def _(dy, xb, tape, handle):
    mod = module_of(handle)
    f = lambda p: dy.new_empty(p.shape, dtype=F32)
    return f(mod.linear_1.weight), f(mod.norm.weight), f(mod.linear_2.weight), f(mod.norm_2.weight)


def _mlp_setup(ctx, inputs, output):
    ctx.handle = inputs[5]
    ctx.module = module_of(ctx.handle)       # the graph keeps its module alive: the handle cannot go stale before the backward
    ctx.set_materialize_grads(False)         # or autograd zero-fills a gradient for the saved-state outputs (xb, tape) every step
    ctx.save_for_backward(output[1] if output[1].numel() else inputs[0], output[2])


def _mlp_bwd(ctx, dy, _dxb, _dtape):
    if dy is None:
        return (None,) * 6
    xb, tape = ctx.saved_tensors
    dW1, dg1, dW2, dg2 = torch.ops.ta355.mlp_projector_backward(dy, xb, tape, ctx.handle)
    return None, dW1, dg1, dW2, dg2, None


mlp_projector.register_autograd(_mlp_bwd, setup_context=_mlp_setup)


# ============================================================================ MoE projector
@torch.library.custom_op("ta355::moe_projector", mutates_args=())
def moe_projector(x: Tensor, noise: Optional[Tensor], params: Sequence[Tensor], handle: int,
                  training: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """x [B, S, E] -> (y f32 [B, N, D], aux f32 [], bf16 image of x, tape).  ``params``: norm.weight, router.weight, then
    fc1.weight, fc1.bias, fc2.weight, fc2.bias of routed experts 0..E-1 and of the shared expert (``_param_list``)."""
    mod = module_of(handle)
    B, S, _ = x.shape
    xb, xb_out = _bf16_image(x)
    wts = mod._packed_weights()
    L_ = _lib.lib()
    dev = x.device
    N = mod.get_output_length(S)
    tape = torch.empty(L_.ta_moe_tape_bytes(C.byref(wts), B, S), device=dev, dtype=torch.uint8)
    y = torch.empty((B, N, mod.llm_dim), device=dev, dtype=F32)
    aux = torch.zeros((), device=dev, dtype=F32)
    _lib.check(L_.ta_moe_projector_forward(C.byref(wts), ptr(xb), B, S, ptr(noise), int(training), ptr(y), ptr(aux), ptr(tape),
                                           stream()), "ta_moe_projector_forward")
    return y, aux, xb_out, tape


@moe_projector.register_fake
def _(x, noise, params, handle, training):
    mod = module_of(handle)
    B, S, _ = x.shape
    n_tape = _lib.lib().ta_moe_tape_bytes(C.byref(mod._packed_weights_meta()), B, S)
    return (x.new_empty((B, mod.get_output_length(S), mod.llm_dim), dtype=F32), x.new_empty((), dtype=F32),
            x.new_empty((0,) if _is_bf16_image(x) else x.shape, dtype=BF16), x.new_empty((n_tape,), dtype=torch.uint8))


@torch.library.custom_op("ta355::moe_projector_backward", mutates_args=())
def moe_projector_backward(dy: Tensor, d_aux: Tensor, xb: Tensor, noise: Optional[Tensor], tape: Tensor, handle: int,
                           training: bool) -> List[Tensor]:
    """Gradients of sum(dy * y) + d_aux * aux (d_aux: a DEVICE scalar) -> [d norm.weight, d router.weight, then one STACKED
    tensor per kind over the E routed experts + the shared expert: d fc1.weight [E+1, H, In], d fc1.bias [E+1, H],
    d fc2.weight [E+1, D, H], d fc2.bias [E+1, D]].  (Operator outputs may not alias one another, and the constant stride
    between the experts' buffers is what lets the library write them from one grouped launch.)"""
    mod = module_of(handle)
    B, S, _ = xb.shape
    wts = mod._packed_weights()
    L_ = _lib.lib()
    dev = dy.device
    dy = dy.to(F32).contiguous()
    E = mod.num_experts
    adapters = list(mod.experts) + [mod.shared_expert]
    f = lambda p: torch.empty(p.shape, device=dev, dtype=F32)
    g_norm, g_router = f(mod.norm.weight), f(mod.router.weight)
    # the adapters' gradients of one kind are slices of ONE tensor: a constant stride between the experts' buffers is what
    # lets the library write all per-expert weight gradients from one grouped launch
    stack = lambda ps: torch.empty((len(ps),) + tuple(ps[0].shape), device=dev, dtype=F32)
    GW1, Gb1 = stack([a.fc1.weight for a in adapters]), stack([a.fc1.bias for a in adapters])
    GW2, Gb2 = stack([a.fc2.weight for a in adapters]), stack([a.fc2.bias for a in adapters])
    gW1, gb1, gW2, gb2 = (list(t.unbind(0)) for t in (GW1, Gb1, GW2, Gb2))
    arr = lambda ts: (C.c_void_p * (E + 1))(*[t.data_ptr() for t in ts])
    ws = torch.empty(L_.ta_moe_bwd_workspace_bytes(C.byref(wts), B, S), device=dev, dtype=torch.uint8)
    da = d_aux.to(device=dev, dtype=F32).reshape(1).contiguous()
    _lib.check(L_.ta_moe_projector_backward_dev(C.byref(wts), ptr(xb), B, S, ptr(dy), ptr(da), ptr(noise), int(training),
                                                ptr(tape), ptr(g_norm), ptr(g_router), arr(gW1), arr(gb1), arr(gW2), arr(gb2),
                                                ptr(ws), ws.numel(), stream()), "ta_moe_projector_backward_dev")
    return [g_norm, g_router, GW1, Gb1, GW2, Gb2]


def moe_router_aux_grads(d_aux: Tensor, xb: Tensor, noise: Optional[Tensor], tape: Tensor, handle: int, training: bool):
    """d_aux * d(aux) / d(norm.weight), d(router.weight) alone (ta_moe_router_aux_grads): the part of those two gradients that must
    NOT be divided by the global token count."""
    mod = module_of(handle)
    B, S, _ = xb.shape
    wts = mod._packed_weights()
    L_ = _lib.lib()
    dev = xb.device
    sn = torch.empty(mod.norm.weight.shape, device=dev, dtype=F32)
    sr = torch.empty(mod.router.weight.shape, device=dev, dtype=F32)
    ws = torch.empty(L_.ta_moe_bwd_workspace_bytes(C.byref(wts), B, S), device=dev, dtype=torch.uint8)
    da = d_aux.to(device=dev, dtype=F32).reshape(1).contiguous()
    _lib.check(L_.ta_moe_router_aux_grads(C.byref(wts), ptr(xb), B, S, ptr(da), ptr(noise), int(training), ptr(tape), ptr(sn), ptr(sr),
                                          ptr(ws), ws.numel(), stream()), "ta_moe_router_aux_grads")
    return sn, sr


@moe_projector_backward.register_fake
def _(dy, d_aux, xb, noise, tape, handle, training):
    mod = module_of(handle)
    n = mod.num_experts + 1
    a = mod.shared_expert
    f = lambda *s: dy.new_empty(s, dtype=F32)
    return [f(*mod.norm.weight.shape), f(*mod.router.weight.shape), f(n, *a.fc1.weight.shape), f(n, *a.fc1.bias.shape),
            f(n, *a.fc2.weight.shape), f(n, *a.fc2.bias.shape)]


def _moe_setup(ctx, inputs, output):
    x, noise, params, handle, training = inputs
    ctx.handle, ctx.training, ctx.has_noise, ctx.n_params = handle, training, noise is not None, len(params)
    ctx.module = module_of(handle)
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(output[2] if output[2].numel() else x, output[3], *([noise] if noise is not None else []))


def _moe_bwd(ctx, dy, d_aux, _dxb, _dtape):
    if dy is None and d_aux is None:
        return None, None, [None] * ctx.n_params, None, None
    xb, tape, *rest = ctx.saved_tensors
    noise = rest[0] if ctx.has_noise else None
    if d_aux is None:
        d_aux = torch.zeros((), device=dy.device, dtype=F32)
    if dy is None:
        dy = torch.zeros((xb.shape[0], module_of(ctx.handle).get_output_length(xb.shape[1]), module_of(ctx.handle).llm_dim),
                         device=xb.device, dtype=F32)
    direct = getattr(ctx.module, "_grad_direct", None)
    if direct is not None and ctx.training:
        # ASRTrainer, one micro-batch per optimizer step: the library writes every gradient straight into its segment of the trainer's flat
        # buffer (the .grad views) -- no 22 temporaries, no 22 autograd accumulation launches.  The routed experts' segments sit at a
        # constant stride there as well (fc1.weight, fc1.bias, fc2.weight, fc2.bias per expert), so the grouped launches still apply.
        mod, L_ = ctx.module, _lib.lib()
        B, S, _ = xb.shape
        wts = mod._packed_weights()
        E = mod.num_experts
        g_norm, g_router = direct[0], direct[1]
        gW1, gb1, gW2, gb2 = ([direct[2 + 4 * i + j] for i in range(E + 1)] for j in range(4))
        arr = lambda ts: (C.c_void_p * (E + 1))(*[t.data_ptr() for t in ts])
        ws = torch.empty(L_.ta_moe_bwd_workspace_bytes(C.byref(wts), B, S), device=xb.device, dtype=torch.uint8)
        da = d_aux.to(device=xb.device, dtype=F32).reshape(1).contiguous()
        _lib.check(L_.ta_moe_projector_backward_dev(C.byref(wts), ptr(xb), B, S, ptr(dy.to(F32).contiguous()), ptr(da), ptr(noise), int(ctx.training),
                                                    ptr(tape), ptr(g_norm), ptr(g_router), arr(gW1), arr(gb1), arr(gW2), arr(gb2),
                                                    ptr(ws), ws.numel(), stream()), "ta_moe_projector_backward_dev")
        # more than one rank with one micro-batch per step: the gradients go straight to the flat buffer AND the auxiliary share of
        # the two router-path gradients goes to its shadow segment (this branch used to return before the shadow fill,
        # which left aux at weight 1 / N on N > 1 ranks)
        shadow = getattr(ctx.module, "_aux_shadow", None)
        if shadow is not None:
            sn, sr = moe_router_aux_grads(d_aux, xb, noise, tape, ctx.handle, ctx.training)
            shadow["norm.weight"].add_(sn)
            shadow["router.weight"].add_(sr)
        return None, None, [None] * ctx.n_params, None, None
    g_norm, g_router, GW1, Gb1, GW2, Gb2 = torch.ops.ta355.moe_projector_backward(dy, d_aux, xb, noise, tape, ctx.handle,
                                                                                   ctx.training)
    shadow = getattr(ctx.module, "_aux_shadow", None)         # (ASRTrainer: the auxiliary-loss share of the two router-path gradients)
    if shadow is not None and ctx.training:
        sn, sr = moe_router_aux_grads(d_aux, xb, noise, tape, ctx.handle, ctx.training)
        shadow["norm.weight"].add_(sn)
        shadow["router.weight"].add_(sr)
    grads = [g_norm, g_router]                                # the order of MoEAudioProjector._param_list()
    for i in range(GW1.shape[0]):
        grads += [GW1[i], Gb1[i], GW2[i], Gb2[i]]
    return None, None, grads, None, None


moe_projector.register_autograd(_moe_bwd, setup_context=_moe_setup)


# ============================================================================ LM + shifted cross-entropy
@torch.library.custom_op("ta355::lm_forward_loss", mutates_args=())
def lm_forward_loss(audio: Tensor, trainable: Sequence[Tensor], handle: int, input_ids: Tensor, src_row: Optional[Tensor],
                    kmask: Optional[Tensor], label_rows: Optional[Tensor], label_targets: Optional[Tensor], n_label_rows: int,
                    loss_scale: float, want_logits: bool, pos: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (loss f32 [], per-row nll f32 [max(n, 1)], logits bf16 [B*L, vocab_pad] or empty, tape, workspace).
    ``audio`` f32 [rows, D]: the projector's output rows that replace the <audio> positions (``src_row`` from
    ta_audio_index); ``trainable``: the LoRA masters or, with a trainable base LM, its fp32 masters (autograd inputs);
    ``pos`` int32 [B*L]: RoPE position of every token row (the reference's ``position_ids``; None = arange(L) per clip)."""
    lm = module_of(handle)
    a = audio.detach().to(F32).contiguous()
    loss, nll, logits, c = lm.forward_loss(input_ids, src_row, a, kmask, label_rows, label_targets, n_label_rows, loss_scale, want_logits,
                                           pos=pos)
    return loss.reshape(()), nll, (logits if logits is not None else _empty(a.device)), c["tape"], c["ws"]


@lm_forward_loss.register_fake
def _(audio, trainable, handle, input_ids, src_row, kmask, label_rows, label_targets, n_label_rows, loss_scale, want_logits, pos):
    lm = module_of(handle)
    B, L = input_ids.shape
    logits = audio.new_empty((B * L, lm.vocab_pad), dtype=BF16) if want_logits else audio.new_empty((0,))
    L_ = _lib.lib()
    n_tape = L_.ta_lm_tape_bytes(C.byref(lm._w), B, L, n_label_rows)
    n_ws = L_.ta_lm_workspace_bytes(C.byref(lm._w), B, L, n_label_rows)
    u8 = lambda n: audio.new_empty((n,), dtype=torch.uint8)
    return audio.new_empty((), dtype=F32), audio.new_empty((max(n_label_rows, 1),), dtype=F32), logits, u8(n_tape), u8(n_ws)


@torch.library.custom_op("ta355::lm_backward", mutates_args=())
def lm_backward(tape: Tensor, ws: Tensor, handle: int, input_ids: Tensor, src_row: Optional[Tensor], kmask: Optional[Tensor],
                label_rows: Optional[Tensor], n_label_rows: int, n_audio_rows: int, want_d_audio: bool,
                pos: Optional[Tensor]) -> List[Tensor]:
    """loss.backward() through the LM for d(loss) = 1 -> [d_audio f32 [n_audio_rows, D] (or empty), then one gradient per
    tensor of ``trainable`` (empty tensors when the LM accumulates straight into Parameter.grad: ASRTrainer's mode)]."""
    lm = module_of(handle)
    B, L = input_ids.shape
    ctx = dict(tape=tape, ws=ws, B=B, L=L, src_row=src_row, kmask=kmask, pos=pos, label_rows=label_rows,
               n_label_rows=n_label_rows, ids=input_ids)
    d_audio, _, lg = lm.backward_from_ctx(ctx, n_audio_rows, want_d_audio=want_d_audio)
    dev = tape.device
    out = [d_audio if d_audio is not None else _empty(dev)]
    for g in (lg or []):
        out.append(g if g is not None else _empty(dev))
    return out


@lm_backward.register_fake
def _(tape, ws, handle, input_ids, src_row, kmask, label_rows, n_label_rows, n_audio_rows, want_d_audio, pos):
    lm = module_of(handle)
    f = lambda *s: tape.new_empty(s, dtype=F32)
    ps = lm.lora_parameters() or lm.ft_parameters() or []
    return [f(n_audio_rows, lm.config.hidden_size) if want_d_audio else f(0)] + [f(*p.shape) for p in ps]


def _lm_setup(ctx, inputs, output):
    (audio, trainable, handle, input_ids, src_row, kmask, label_rows, _targets, n_label_rows, _scale, _want, pos) = inputs
    ctx.handle, ctx.n_label_rows, ctx.n_audio, ctx.n_train = handle, n_label_rows, audio.shape[0], len(trainable)
    ctx.module = module_of(handle)
    ctx.want_d_audio = audio.requires_grad
    # without this autograd materialises ZERO gradients for the unused outputs on every backward -- the tape alone is
    # 8.9 GB at B = 32 (a 1.35 ms fill per step, measured), the workspace 1.2 GB
    ctx.set_materialize_grads(False)
    ctx.present = [t is not None for t in (src_row, kmask, label_rows, pos)]
    ctx.save_for_backward(output[3], output[4], input_ids, *[t for t in (src_row, kmask, label_rows, pos) if t is not None])


def _lm_bwd(ctx, g_loss, _g_nll, _g_logits, _g_tape, _g_ws):
    if g_loss is None:                                        # the loss was not used
        return (None, [None] * ctx.n_train) + (None,) * 10
    tape, ws, ids, *rest = ctx.saved_tensors
    it = iter(rest)
    src_row, kmask, label_rows, pos = (next(it) if p else None for p in ctx.present)
    out = torch.ops.ta355.lm_backward(tape, ws, ctx.handle, ids, src_row, kmask, label_rows, ctx.n_label_rows, ctx.n_audio,
                                      ctx.want_d_audio, pos)
    d_audio = out[0] * g_loss if ctx.want_d_audio else None
    # gradients of the trainable LM tensors: scaled by d(loss) -- except in accumulate_into_grad mode, where the kernels
    # have already added them to Parameter.grad (empty placeholders come back) and d(loss) is 1 by ASRTrainer's contract
    tg = [(g * g_loss if g.numel() else None) for g in out[1:]]
    tg += [None] * (ctx.n_train - len(tg))
    return (d_audio, tg[: ctx.n_train]) + (None,) * 10


lm_forward_loss.register_autograd(_lm_bwd, setup_context=_lm_setup)

OPERATORS = ("logmel", "encoder_forward", "mlp_projector", "mlp_projector_backward", "moe_projector", "moe_projector_backward",
             "lm_forward_loss", "lm_backward")
