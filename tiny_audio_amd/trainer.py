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
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.distributed as dist

from . import ops


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


class FlatTrainable:
    """All trainable parameters as views of ONE fp32 buffer; gradients likewise, plus two trailing slots:
    [ ... grads ..., label_token_count, loss_sum ].  (SURVEY.md section 2a, C1 + C2 folded together.)"""

    EXTRA = 2

    def __init__(self, named_params, device=None):
        self.names, self.params = zip(*[(n, p) for n, p in named_params if p.requires_grad])
        device = device or self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for s in self.sizes:
            self.offsets.append(self.offsets[-1] + ((s + 3) // 4) * 4)      # keep every segment 16-byte aligned
        self.n = self.offsets[-1]
        self.flat_p = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.flat_g = torch.zeros(self.n + self.EXTRA, device=device, dtype=torch.float32)
        self.flat_m = torch.zeros(self.n, device=device, dtype=torch.float32)
        self.flat_v = torch.zeros(self.n, device=device, dtype=torch.float32)
        with torch.no_grad():
            for p, o, s in zip(self.params, self.offsets, self.sizes):
                self.flat_p[o:o + s].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + s].view_as(p)
                p.grad = self.flat_g[o:o + s].view_as(p)
        # decay on weight matrices only; norm scales and biases get 0 (scripts/train.py:397-432)
        self.decay = [not (n.endswith("bias") or "norm" in n.split(".")[-2:][0] or p.ndim < 2 or getattr(p, "_no_decay", False))
                      for n, p in zip(self.names, self.params)]

    @property
    def grads(self):
        return self.flat_g[: self.n]

    @property
    def count_slot(self):
        return self.flat_g[self.n: self.n + 1]

    @property
    def loss_slot(self):
        return self.flat_g[self.n + 1: self.n + 2]

    def zero_grad(self):
        self.flat_g.zero_()
        for p, o, s in zip(self.params, self.offsets, self.sizes):       # autograd may have replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_g[o:o + s].data_ptr():
                p.grad = self.flat_g[o:o + s].view_as(p)


def allreduce_flat(flat: torch.Tensor, group=None) -> torch.Tensor:
    """SUM all-reduce of the flat [grads | count | loss] buffer (RCCL on GPUs, gloo in the CPU tests)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class ASRTrainer:
    def __init__(self, model, args: Optional[TrainingArguments] = None, group=None, decoder_learning_rate: Optional[float] = None,
                 decoder_weight_decay: Optional[float] = None, projector_weight_decay: Optional[float] = None):
        """``decoder_*`` / ``projector_weight_decay``: the split parameter groups of scripts/train.py:384-437 -- parameters
        under ``language_model.`` (the LoRA adapters in stage 2) take the decoder LR / weight decay, everything else the
        base LR and the projector weight decay; each falls back to ``args.learning_rate`` / ``args.weight_decay``; norm
        scales and biases never decay.  The schedule multiplies both LRs alike (one LambdaLR over all groups)."""
        self.model, self.args, self.group = model, args or TrainingArguments(), group
        self.decoder_learning_rate, self.decoder_weight_decay = decoder_learning_rate, decoder_weight_decay
        self.projector_weight_decay = projector_weight_decay
        self.flat = FlatTrainable(list(model.named_parameters()))
        lm = getattr(model, "language_model", None)
        if lm is not None and getattr(lm, "train_base", False):
            lm.accumulate_into_grad = True      # d(loss) is 1 here: weight gradients go straight into the flat buffer
        self.sqnorm = torch.zeros(1, device=self.flat.flat_p.device, dtype=torch.float32)
        self.global_step = 0
        self._micro = 0

    def _invalidate(self):
        proj = getattr(self.model, "projector", None)
        if proj is not None and hasattr(proj, "_pack_versions"):
            proj._pack_versions = None          # the HIP optimizer writes masters behind autograd's version counter
        lm = getattr(self.model, "language_model", None)
        if lm is not None and getattr(lm, "train_base", False):
            lm._ft_versions = None              # bf16 W / W^T images are rebuilt before the next forward

    def training_step(self, batch: dict):
        """One micro-batch: forward + backward of SUM-CE; optimizer step every gradient_accumulation_steps.
        Returns the (not yet normalised) loss sum tensor of this micro-batch."""
        if self._micro == 0:
            self.flat.zero_grad()
        out = self.model(**batch, num_items_in_batch=1.0, return_logits=False)
        out.loss.backward()
        with torch.no_grad():
            self.flat.count_slot.add_(float(out.n_label_tokens))
            self.flat.loss_slot.add_(out.loss.detach().reshape(1))
        self._micro += 1
        if self._micro == self.args.gradient_accumulation_steps:
            self.optimizer_step()
            self._micro = 0
        return out.loss.detach()

    def optimizer_step(self):
        a, f = self.args, self.flat
        allreduce_flat(f.flat_g, self.group)                      # grads, token count and loss sum in one collective
        self.global_step += 1
        self.sqnorm.zero_()
        ops.grad_sqnorm(f.grads, self.sqnorm)
        mult = lr_multiplier(self.global_step - 1, a)
        for name, o, s, dec in zip(f.names, f.offsets, f.sizes, f.decay):
            lr_p, wd_p = self.group_hparams(name, dec)
            ops.adamw_step(f.flat_p[o:o + s], f.flat_g[o:o + s], f.flat_m[o:o + s], f.flat_v[o:o + s], lr_p * mult,
                           a.adam_beta1, a.adam_beta2, a.adam_epsilon, wd_p, self.global_step,
                           sqnorm=self.sqnorm, max_norm=a.max_grad_norm, grad_scale=1.0, denom=f.count_slot)
        self._invalidate()

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
        """Global mean loss of the last optimizer step (one host sync; for logging)."""
        f = self.flat
        return float((f.loss_slot / f.count_slot.clamp(min=1)).item())

    def last_grad_norm(self) -> float:
        f = self.flat
        return float((self.sqnorm.sqrt() / f.count_slot.clamp(min=1)).item())
