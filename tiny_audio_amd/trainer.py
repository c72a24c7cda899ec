"""Data-parallel training step for the projector (what HF ``Trainer`` does for scripts/train.py:630-643).

Reproduced semantics:
  * loss = sum of per-token CE over ALL ranks and accumulation micro-batches / global label-token count
    (TF:trainer.py:2040-2050,2141-2201).  Ranks back-propagate the SUM; the token count rides in the same flat
    buffer as the gradients, so one RCCL all-reduce per optimizer step delivers both (no second collective).
  * clip_grad_norm_(max_grad_norm) on the global norm, then AdamW on fp32 masters with decay only on weight
    matrices (scripts/train.py:427-432); cosine or polynomial(power) schedule with linear warm-up
    (configs/training/production.yaml:5-9, configs/experiments/transcription.yaml:22-25).

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL over xGMI.  The reduction helpers are pure
torch so the N>1 logic is covered by gloo world_size-2 tests on CPU; the optimizer kernels are HIP-only.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib, ops


@dataclass
class TrainingArguments:
    learning_rate: float = 1e-3
    weight_decay: float = 0.0
    max_grad_norm: float = 1.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    warmup_steps: int = 0
    max_steps: int = 1000
    lr_scheduler_type: str = "constant"          # "constant" | "cosine" | "polynomial"
    lr_scheduler_kwargs: dict = field(default_factory=dict)
    gradient_accumulation_steps: int = 1


def lr_multiplier(step: int, args: TrainingArguments) -> float:
    """transformers.optimization get_{constant,cosine,polynomial_decay}_schedule_with_warmup lambdas."""
    w, total = args.warmup_steps, args.max_steps
    if step < w:
        return step / max(1, w)
    if args.lr_scheduler_type == "constant":
        return 1.0
    if args.lr_scheduler_type == "cosine":
        progress = (step - w) / max(1, total - w)
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 2.0 * 0.5 * progress)))
    if args.lr_scheduler_type == "polynomial":
        lr_init, lr_end = args.learning_rate, args.lr_scheduler_kwargs.get("lr_end", 1e-7)
        power = args.lr_scheduler_kwargs.get("power", 1.0)
        if step > total:
            return lr_end / lr_init
        pct = 1.0 - (step - w) / max(1, total - w)
        return ((lr_init - lr_end) * pct ** power + lr_end) / lr_init
    raise ValueError(f"unknown lr_scheduler_type {args.lr_scheduler_type}")


_HF_NO_DECAY = [re.compile(p) for p in (r"bias", r"layernorm", r"rmsnorm", r"(?:^|\.)norm(?:$|\.)", r"_norm(?:$|\.)")]


def decay_flags(names, params, overrides: bool, layernorm_ids=frozenset()):
    """Which parameters take weight decay, as the reference's two optimizer paths decide it.

    No per-group override (``Trainer.create_optimizer``): every parameter decays except those inside an ``nn.LayerNorm``
    and those whose lower-cased name matches one of HF's patterns -- bias, layernorm, rmsnorm, ``.norm.``, ``_norm.``
    (TF:trainer.py get_decay_parameter_names).  Note ``projector.norm_2.weight`` matches none of them and decays.
    With ``decoder_learning_rate`` / ``decoder_weight_decay`` / ``projector_weight_decay`` set (scripts/train.py:397-405):
    ``get_parameter_names(model, ALL_LAYERNORM_LAYERS)`` minus names containing "bias" -- only nn.LayerNorm is exempt
    there, so RMSNorm scales (the projector's LlamaRMSNorm, Qwen3's norms) DO decay.

    ``layernorm_ids``: id() of parameters that live in an nn.LayerNorm module; parameters whose name carries Blip2's
    ``LayerNorm`` attribute are treated the same.  ``p._no_decay`` marks stacked LM masters that hold RMSNorm scales
    (their flat names here do not carry the reference's ``*_layernorm`` / ``*_norm`` names)."""
    out = []
    for n, p in zip(names, params):
        is_ln = id(p) in layernorm_ids or "LayerNorm" in n
        if overrides:
            out.append(not (is_ln or "bias" in n))
        else:
            out.append(not (is_ln or getattr(p, "_no_decay", False) or any(r.search(n.lower()) for r in _HF_NO_DECAY)))
    return out


class FlatTrainable:
    """All trainable parameters as views of ONE fp32 buffer; gradients likewise, plus two trailing slots:
    [ ... grads ..., (shadows ...,) label_token_count, loss_sum ].  (SURVEY.md section 2a, C1 + C2 folded together.)

    ``shadow_of``: names of parameters that also get a SHADOW gradient segment between the gradients and the two slots -- the share
    of their gradient that must keep full weight under the optimizer's division by the global token count (the MoE projector's
    auxiliary losses, round 4): it travels in the same all-reduce, and ``g += (count - 1) * shadow`` before the update undoes the
    division for that share."""

    EXTRA = 2

    def __init__(self, named_params, device=None, shadow_of=()):
        self.names, self.params = zip(*[(n, p) for n, p in named_params if p.requires_grad])
        device = device or self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for s in self.sizes:
            self.offsets.append(self.offsets[-1] + ((s + 3) // 4) * 4)      # keep every segment 16-byte aligned
        self.n = self.offsets[-1]
        self.shadow_names = [n for n in self.names if n in set(shadow_of)]
        self.shadow_offsets, so = {}, self.n
        for n in self.shadow_names:
            sz = self.sizes[self.names.index(n)]
            self.shadow_offsets[n] = (so, sz)
            so += ((sz + 3) // 4) * 4
        self.n_all = so                                   # gradients + shadows
        self.flat_p = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.flat_g = torch.zeros(self.n_all + self.EXTRA, device=device, dtype=torch.float32)
        self.flat_m = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.flat_v = torch.zeros(self.n, device=device, dtype=torch.float32)
        with torch.no_grad():
            for p, o, s in zip(self.params, self.offsets, self.sizes):
                self.flat_p[o:o + s].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + s].view_as(p)
                p.grad = self.flat_g[o:o + s].view_as(p)
        self.decay = decay_flags(self.names, self.params, overrides=False)

    @property
    def grads(self):
        return self.flat_g[: self.n]

    @property
    def count_slot(self):
        return self.flat_g[self.n_all: self.n_all + 1]

    @property
    def loss_slot(self):
        return self.flat_g[self.n_all + 1: self.n_all + 2]

    def shadow(self, name):
        o, sz = self.shadow_offsets[name]
        return self.flat_g[o:o + sz].view_as(self.params[self.names.index(name)])

    def grad_of(self, name):
        i = self.names.index(name)
        return self.flat_g[self.offsets[i]:self.offsets[i] + self.sizes[i]].view_as(self.params[i])

    def zero_grad(self):
        self.flat_g.zero_()
        for p, o, s in zip(self.params, self.offsets, self.sizes):       # autograd may have replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_g[o:o + s].data_ptr():
                p.grad = self.flat_g[o:o + s].view_as(p)


def _distributed(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


COLLECTIVES = {"allreduce_flat": 0}      # data-path collectives issued by this process (bench.py reports them per step)


def allreduce_flat(flat: torch.Tensor, group=None, async_op: bool = False):
    """SUM all-reduce of the flat [grads | count | loss] buffer (RCCL on GPUs, gloo in the CPU tests).
    ``async_op``: returns the collective's Work handle (None on a single rank) instead of waiting for it."""
    if _distributed(group):
        COLLECTIVES["allreduce_flat"] += 1
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return work if async_op else flat
    return None if async_op else flat


class _StreamTimer:
    """Sum of event-bracketed spans on the current stream (no host sync until ``total_ms`` is read)."""

    def __init__(self, enabled: bool):
        self.enabled, self.spans = enabled and torch.cuda.is_available(), []

    def __enter__(self):
        if self.enabled:
            self._a = torch.cuda.Event(enable_timing=True)
            self._a.record()
        return self

    def __exit__(self, *exc):
        if self.enabled:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.spans.append((self._a, b))
        return False

    def total_ms(self, reset: bool = True) -> float:
        if not self.spans:
            return 0.0
        self.spans[-1][1].synchronize()
        t = sum(a.elapsed_time(b) for a, b in self.spans)
        if reset:
            self.spans = []
        return t


class ASRTrainer:
    def __init__(self, model, args: Optional[TrainingArguments] = None, group=None, decoder_learning_rate: Optional[float] = None,
                 decoder_weight_decay: Optional[float] = None, projector_weight_decay: Optional[float] = None,
                 overlap_allreduce: bool = False, time_allreduce: bool = False, aux_shadow: Optional[bool] = None):
        """``decoder_*`` / ``projector_weight_decay``: the split parameter groups of scripts/train.py:384-437 -- parameters
        under ``language_model.`` (the LoRA adapters in stage 2) take the decoder LR / weight decay, everything else the
        base LR and the projector weight decay; each falls back to ``args.learning_rate`` / ``args.weight_decay``; norm
        scales and biases never decay.  The schedule multiplies both LRs alike (one LambdaLR over all groups).

        ``overlap_allreduce``: the flat-gradient all-reduce of step n is launched asynchronously (RCCL's own stream) and
        its optimizer update is applied inside step n+1 right after the FROZEN encoder's forward -- the only part of a step
        that does not read trainable weights -- so the collective runs under ~half a step of compute.  Same arithmetic,
        same order of updates; ``flush()`` applies a still-pending update (end of training, before saving / evaluating).
        ``time_allreduce``: bracket the time the compute stream spends on / waiting for the collective with events
        (``allreduce_exposed_ms()``).  ``aux_shadow``: see ``_aux_direct`` below."""
        self.model, self.args, self.group = model, args or TrainingArguments(), group
        self.decoder_learning_rate, self.decoder_weight_decay = decoder_learning_rate, decoder_weight_decay
        self.projector_weight_decay = projector_weight_decay
        # auxiliary losses (MoE balance + z loss) reach projector.norm.weight and projector.router.weight only: their auxiliary share
        # gets a shadow segment in the flat buffer (see training_step), and the projector fills it in its backward
        proj = getattr(model, "projector", None)
        aux_names = [f"projector.{n}" for n in getattr(proj, "aux_shadow_params", ())] if hasattr(proj, "get_aux_loss") else []
        # ... needed only when the global token count N is not known before the backward: more than one rank, or gradient accumulation.
        # One rank, one micro-batch per step: N is this batch's own (host-known) count, aux is back-propagated as aux * N and the
        # projector skips the shadow recomputation (one router / norm backward less per step: -0.12 ms)
        world = 1
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(group)
        # ``aux_shadow=True`` forces the shadow form on one rank too (tests: the N > 1 arithmetic on a 1-GPU box)
        self._aux_direct = world == 1 and int(self.args.gradient_accumulation_steps) == 1 and not aux_shadow
        self.flat = FlatTrainable(list(model.named_parameters()), shadow_of=() if self._aux_direct else aux_names)
        if hasattr(proj, "_aux_shadow"):
            proj._aux_shadow = None
        if self.flat.shadow_names:
            proj._aux_shadow = {n[len("projector."):]: self.flat.shadow(n) for n in self.flat.shadow_names}
        # one micro-batch per optimizer step: the MoE projector's backward writes its 22 gradients straight into their flat-buffer
        # segments (torch_ops._moe_bwd); with gradient accumulation autograd's += stays in charge
        if hasattr(proj, "_grad_direct"):
            proj._grad_direct = None
        if hasattr(proj, "_param_list") and hasattr(proj, "get_aux_loss") and int(self.args.gradient_accumulation_steps) == 1:
            pl = proj._param_list()
            if all(any(q is p_ for q in self.flat.params) for p_ in pl):
                proj._grad_direct = [self.flat.grad_of(self.flat.names[[id(q) for q in self.flat.params].index(id(p_))]) for p_ in pl]
        overrides = decoder_learning_rate is not None or decoder_weight_decay is not None or projector_weight_decay is not None
        ln_ids = frozenset(id(p) for m in model.modules() if isinstance(m, torch.nn.LayerNorm) for p in m.parameters(recurse=False))
        self.flat.decay = decay_flags(self.flat.names, self.flat.params, overrides, ln_ids)
        lm = getattr(model, "language_model", None)
        if lm is not None and getattr(lm, "train_base", False):
            lm.accumulate_into_grad = True      # d(loss) is 1 here: weight gradients go straight into the flat buffer
        self.sqnorm = torch.zeros(1, device=self.flat.flat_p.device, dtype=torch.float32)
        self._sq_scratch = torch.empty(1024, device=self.flat.flat_p.device, dtype=torch.float32)   # ta_grad_sqnorm's per-block partials
        self._adam_table = None
        self.global_step = 0
        self._micro = 0
        self.overlap_allreduce = bool(overlap_allreduce)
        self._pending = None                    # (work handle or None,) of the optimizer step that has not been applied yet
        self._need_zero = True
        self._timer = _StreamTimer(time_allreduce)
        self._last = torch.zeros(3, device=self.flat.flat_p.device, dtype=torch.float32)   # loss sum, token count, sqnorm
        self._aux_sum = None                    # sum of the auxiliary losses of the current optimizer step (device scalar)
        # DDP ranks must not draw identical dropout patterns (the reference's per-process torch RNG): fold the rank in
        if _distributed(group) and hasattr(model, "_drop_seed"):
            model._drop_seed += 1000003 * dist.get_rank(group)

    def _invalidate(self):
        proj = getattr(self.model, "projector", None)
        if proj is not None and hasattr(proj, "_pack_versions"):
            proj._pack_versions = None          # the HIP optimizer writes masters behind autograd's version counter
        lm = getattr(self.model, "language_model", None)
        if lm is not None and getattr(lm, "train_base", False):
            lm._ft_versions = None              # bf16 W / W^T images are rebuilt before the next forward

    def training_step(self, batch: dict, num_items_in_batch=None, return_logits: bool = False):
        """One micro-batch: forward + backward of SUM-CE; optimizer step every gradient_accumulation_steps.
        Returns the (not yet normalised) CE sum of this micro-batch.

        Auxiliary losses (MoE balance + z loss): HF Trainer back-propagates ``sum(nll) / num_items_in_batch + aux`` per
        micro-batch -- the auxiliary term is NOT token-normalised (tiny_audio/asr_modeling.py:528-531 adds it after the
        LM's loss).  Here every rank back-propagates the CE SUM + aux and the optimizer divides by the global token count N,
        which is known only after the all-reduce.  Round 4: the projector's backward ALSO writes the auxiliary share of the two
        gradients it reaches (norm.weight, router.weight) into a shadow segment of the flat buffer; shadow and gradients travel in
        the ONE collective of the step, and ``g += (N - 1) * shadow`` in front of the update leaves ``g_ce / N + g_aux``: full
        weight, no second collective, no ``num_items_in_batch`` needed (rounds 1-3 all-reduced the count ahead of the backward
        and back-propagated ``aux * N``).  ``num_items_in_batch`` is accepted and ignored."""
        after_encoder = self._apply_pending if self._pending is not None else None
        if self._pending is None and self._micro == 0 and self._need_zero:
            self._zero()
        out = self.model(**batch, num_items_in_batch=1.0, return_logits=return_logits,
                         **({"after_encoder": after_encoder} if after_encoder else {}))
        self.last_logits = out.logits           # None unless return_logits (the reference's outputs.logits [B, L, V])
        ce = getattr(out, "loss_ce", None)
        aux = out.aux_loss if (ce is not None and out.aux_loss is not None and out.aux_loss.numel() > 0) else None
        if aux is not None and self._aux_direct:
            (ce + aux.to(ce.device) * float(out.n_label_tokens)).backward()      # the optimizer divides by N: aux keeps its full weight
            with torch.no_grad():
                self._aux_sum = aux.detach().clone() if self._aux_sum is None else self._aux_sum + aux.detach()
        elif aux is not None:
            if not self.flat.shadow_names:
                raise _lib.Ta355Error("a projector with an auxiliary loss must name the parameters it reaches (aux_shadow_params)")
            (ce + aux.to(ce.device)).backward()
            with torch.no_grad():
                self._aux_sum = aux.detach().clone() if self._aux_sum is None else self._aux_sum + aux.detach()
        else:
            ce = out.loss if ce is None else ce
            ce.backward()
        with torch.no_grad():
            self.flat.count_slot.add_(float(out.n_label_tokens))
            self.flat.loss_slot.add_(ce.detach().reshape(1))
        self._micro += 1
        if self._micro == self.args.gradient_accumulation_steps:
            self.optimizer_step()
            self._micro = 0
        return ce.detach()

    def _zero(self):
        self.flat.zero_grad()
        self._aux_sum = None
        self._need_zero = False

    def optimizer_step(self):
        """All-reduce [grads | token count | loss sum] and apply clip + AdamW -- at once, or (overlap_allreduce) launch
        the collective now and apply the update inside the next training_step / flush()."""
        f = self.flat
        if self.overlap_allreduce:
            self._pending = (allreduce_flat(f.flat_g, self.group, async_op=True),)
            return
        with self._timer:
            allreduce_flat(f.flat_g, self.group)                  # grads, token count and loss sum in one collective
        self._apply_update()

    def _apply_pending(self):
        if self._pending is None:
            return
        (work,) = self._pending
        self._pending = None
        with self._timer:
            if work is not None:
                work.wait()                                       # the compute stream waits for the collective here
        self._apply_update()

    def flush(self):
        """Apply an optimizer step whose all-reduce is still in flight (overlap_allreduce)."""
        self._apply_pending()

    def _apply_update(self):
        a, f = self.args, self.flat
        self.global_step += 1
        if f.shadow_names:                      # the auxiliary share keeps its full weight under the division by the token count
            with torch.no_grad():
                nm1 = f.count_slot - 1.0
                for n in f.shadow_names:
                    f.grad_of(n).addcmul_(f.shadow(n), nm1.expand_as(f.shadow(n)))
        self.sqnorm.zero_()
        ops.grad_sqnorm(f.grads, self.sqnorm, self._sq_scratch)
        mult = lr_multiplier(self.global_step - 1, a)
        if self._adam_table is None:              # per-parameter (learning rate, weight decay) as a device table: ONE launch per step
            hp = [self.group_hparams(name, dec) for name, dec in zip(f.names, f.decay)]
            dev = f.flat_p.device
            self._adam_table = (torch.tensor(f.offsets[1:], dtype=torch.int64, device=dev),
                                torch.tensor([h[0] for h in hp], dtype=torch.float32, device=dev),
                                torch.tensor([h[1] for h in hp], dtype=torch.float32, device=dev))
        seg_end, seg_lr, seg_wd = self._adam_table
        ops.adamw_step_multi(f.flat_p, f.grads, f.flat_m, f.flat_v, seg_end, seg_lr, seg_wd, mult, a.adam_beta1, a.adam_beta2,
                             a.adam_epsilon, self.global_step, sqnorm=self.sqnorm, max_norm=a.max_grad_norm, grad_scale=1.0,
                             denom=f.count_slot)
        with torch.no_grad():
            self._last.copy_(torch.cat([f.loss_slot, f.count_slot, self.sqnorm]))
        self._last_aux = self._aux_sum
        self._invalidate()
        if self.overlap_allreduce:
            self._zero()                        # the next backward accumulates into a clean buffer
        else:
            self._need_zero = True

    def group_hparams(self, name: str, decays: bool):
        """(base lr, weight decay) of one parameter: scripts/train.py:406-432."""
        a = self.args
        if name.startswith("language_model."):
            lr = self.decoder_learning_rate if self.decoder_learning_rate is not None else a.learning_rate
            wd = self.decoder_weight_decay if self.decoder_weight_decay is not None else a.weight_decay
        else:
            lr = a.learning_rate
            wd = self.projector_weight_decay if self.projector_weight_decay is not None else a.weight_decay
        return lr, (wd if decays else 0.0)

    def last_loss(self) -> float:
        """Global mean CE of the last APPLIED optimizer step (one host sync; for logging).  Auxiliary losses are not
        included (``last_aux``)."""
        return float((self._last[0] / self._last[1].clamp(min=1)).item())

    def last_aux(self) -> float:
        """Sum of the auxiliary losses over the micro-batches of the last applied optimizer step (this rank)."""
        aux = getattr(self, "_last_aux", None)
        return 0.0 if aux is None else float(aux.item())

    def last_grad_norm(self) -> float:
        return float((self._last[2].sqrt() / self._last[1].clamp(min=1)).item())

    def allreduce_exposed_ms(self, reset: bool = True) -> float:
        """Milliseconds the compute stream spent inside (synchronous mode) or waiting for (overlap mode) the gradient
        all-reduce since the last call; needs ``time_allreduce=True``."""
        return self._timer.total_ms(reset)
