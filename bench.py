#!/usr/bin/env python3
"""bench.py -- training-step audio-seconds/second of the tiny-audio projector-training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  The driver launches the ranks itself (python -m torch.distributed.run ... bench.py
--gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment); a plain `python bench.py --gpus N`
re-launches itself the same way.  The line then carries `n_gpus` = N, `rccl_ranks` (counted by an actual all-reduce),
`allreduce` (bytes, exposed ms per step in the overlapped mode AND -- a second short leg of the same invocation -- in the
synchronous mode) and `replicas` (weight checksum / optimizer step count min == max over the ranks, per-rank ms_per_step,
RCCL version, transports named in RCCL's own NCCL_DEBUG log): the run validates itself.  An exception on ANY rank prints one
JSON line with an "error" field and the process exits non-zero.

One "step" = one full optimizer step of BASELINE.json configs[1]: B synthetic 10 s / 16 kHz clips per GPU ->
GPU log-mel -> frozen GLM-ASR encoder (32 layers) -> MLP projector (H=D=1024) -> frozen Qwen3-0.6B (28 layers,
L=192 tokens, 125 <audio> placeholders, 36 label tokens) -> shifted CE -> backward through LM (dX) and
projector (dW) -> flat gradient all-reduce (RCCL, N>1) -> global-norm clip + AdamW.  Inputs (waveforms, token
ids) are resident in HBM before the timed region; random-init weights at the true shapes (no checkpoints offline).

The single JSON line carries `roofline` (dominant kernel = the MFMA GEMM, timed in situ with HIP events on
its launch stream), `host_inputs` (the same step fed from pinned host memory: the PCIe-inclusive rate, never `value`),
`logits_full` (the same step with the reference's [B, L, V] logits materialised, timed in the same
run), `streams_other` (the same step with the residual streams stored in the OTHER dtype: `--streams`) and, at N=1,
`cpu_baseline` (the numpy oracle timed on the host cores on one clip: median of 5 passes after 2 warm-ups).

Numerics contract: MFMA operands bf16, accumulation fp32, norms / softmax / CE in fp32 -- the reference's
`bf16: true`.  `--streams` chooses where the residual streams are STORED: `f32` = what the training recipe keeps (fp32 modules
under bf16 autocast, configs/config.yaml:14-18 + configs/training/production.yaml:49), `bf16` = what ASRConfig's default
model_dtype="bfloat16" keeps.  `config.streams` names the mode of `value`; `numerics` carries the reference-measured drift of
both regimes (tests/golden/asr_full_recipe.npz) that the parity tests gate against.  Default since round 6: `f32`, the recipe's
regime (`regime` in the line says so in words); rounds 1-5 quoted the bf16-module regime.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: without this RCCL fails with `hipIpcGetMemHandle: invalid argument`.  Set BEFORE the HIP
# runtime comes up (it reads the environment when the first device call initialises it), i.e. before torch touches a GPU
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

# Round 6: the headline runs in the storage mode of the benchmarked RECIPE -- configs/config.yaml:14-18
# `model_dtype: float32` + production.yaml:49 `bf16: true` = fp32 modules under bf16 autocast, i.e. fp32 residual streams.  The
# bf16-module regime (ASRConfig's own default model_dtype, ~7 % faster) is timed in the same line as `streams_other`.
DEFAULT_STREAMS = "f32"
PEAK_BF16_DENSE_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (2:1-sparse marketing figure is 5 PF)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU per step")
    ap.add_argument("--seq-len", type=int, default=192)
    ap.add_argument("--logits", choices=["labelled", "full"], default="labelled",
                    help="'full' additionally materialises outputs.logits [B, L, V] (bf16) every step as the reference does; "
                         "'labelled' computes the loss head only on label positions (identical loss and gradients)")
    ap.add_argument("--streams", choices=["bf16", "f32"], default=DEFAULT_STREAMS,
                    help="storage dtype of the encoder / LM residual streams (ASRConfig.model_dtype): f32 = the training recipe's "
                         "fp32 modules under bf16 autocast, bf16 = bf16 modules (ASRConfig's default).  The other mode is timed as "
                         "`streams_other` in the same line")
    ap.add_argument("--dropout", type=float, default=0.10, help="audio_token_dropout (configs/config.yaml:32)")
    ap.add_argument("--projector", choices=["mlp", "moe", "qformer", "mosa"], default="mlp",
                    help="mlp = BASELINE configs[1]/[2]; moe = configs[3] (shared + 4 routed experts, top-2, jitter on)")
    ap.add_argument("--proj-hidden", type=int, default=1024, help="MLP projector hidden width (2048 = the 'embedded' recipe)")
    ap.add_argument("--lm", choices=["0.6b", "1.7b"], default="0.6b",
                    help="frozen LM shape: Qwen3-0.6B (north_star) or Qwen3-1.7B (hidden 2048, ffn 6144: transcription.yaml:14-16)")
    ap.add_argument("--lora", action="store_true",
                    help="BASELINE configs[4]: stage 2 -- frozen projector + rank-8 LoRA adapters on all 196 Qwen3 linears")
    ap.add_argument("--full-ft", action="store_true",
                    help="full decoder fine-tuning (configs/experiments/embedded.yaml): freeze_language_model=False, MLP projector "
                         "H=2048, decoder lr 1e-4; every LM weight trains (0.6 B fp32 masters, AdamW, bf16 W / W^T images rebuilt per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-logits-full", action="store_true", help="skip the two extra timed legs: materialised outputs.logits, and inputs fed from pinned host memory")
    ap.add_argument("--sync-allreduce", action="store_true", help="N > 1: all-reduce synchronously on the compute stream")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--step-times", action="store_true", help="diagnostic: per-step GPU times of the timed region (events) in the line")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1: torch.distributed backend.  'gloo' (CUDA tensors through the host) + --share-gpu run the whole N > 1 path -- "
                         "real kernels, flat all-reduce, deferred update, replica check -- on a ONE-GPU box; only the transport is not RCCL")
    ap.add_argument("--share-gpu", action="store_true", help="every rank uses cuda:0 (validation on a one-GPU box; the timings mean nothing)")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check WITHOUT a GPU: tiny model on the CPU, every kernel launch a marshalling-only stub, gloo instead "
                         "of RCCL.  Exercises the N > 1 control flow of this script (tests/test_host_logic.py); its numbers mean nothing")
    return ap.parse_args()


def algorithmic_gflop_per_clip(L, V, n_label, full_logits, H=1024, D=1024, F=3072, full_ft=False, projector="mlp", lora=False,
                               lora_rank=8, n_audio=125):
    """BASELINE.md section 3 / SURVEY.md 8(d) (2*MAC, dense).  lm_head is counted at the positions actually computed.
    MoE (configs[3]): every token runs the shared expert + 2 routed experts = 3x the adapter (5120 -> H -> D), forward and
    backward (dW1, dW2, d(act), and d(xn) for the RMSNorm / router gradients).  LoRA (configs[4]): 2*L*r*(in+out) per adapted
    linear forward and twice that backward, 7 linears x 28 layers; the projector is frozen there (no weight gradients).
    QFormer / MOSA projectors are counted as the MLP (their own flops are not modelled: understated, conservative)."""
    conv = 0.983 + 4.915
    enc = 32 * (2 * 500 * 4 * 1280 ** 2 + 4 * 500 ** 2 * 1280 + 4 * 500 * 1280 * 5120) / 1e9
    N = n_audio
    if projector == "moe":
        adapter = 2 * N * (5120 * H + H * D) / 1e9
        proj_f = 3 * adapter + 2 * N * 5120 * 4 / 1e9                                    # + the fp32 router
        proj_b = 3 * (2 * adapter + 2 * N * 5120 * H / 1e9) + 2 * 2 * N * 5120 * 4 / 1e9  # dW1, dW2, d(act), d(xn); router dW + dX
    else:
        proj_f = 2 * N * (5120 * H + H * D) / 1e9
        proj_b = 2 * N * (2 * 5120 * H + 2 * H * D) / 1e9 - 2 * N * 5120 * H / 1e9        # dW1, dW2, dA1 (no dX)
    lm_body = 28 * (2 * L * (D * 2048 + 2 * D * 1024 + 2048 * D + 3 * D * F) + 2 * L * L * 2048) / 1e9   # 16 q / 8 kv heads x 128
    head_rows = (L if full_logits else 0) + n_label
    head = 2 * head_rows * D * V / 1e9 + 2 * n_label * D * V / 1e9                        # fwd (+ labelled dH backward)
    lm_w = 0.0
    if full_ft:     # weight gradients: one more pass over every linear (no attention term) + the tied head's dE at the labelled rows
        lm_w = 28 * (2 * L * (D * 2048 + 2 * D * 1024 + 2048 * D + 3 * D * F)) / 1e9 + 2 * n_label * D * V / 1e9
    lora_gf = 0.0
    if lora:
        io = (D + 2048) + 2 * (D + 1024) + (2048 + D) + 2 * (D + F) + (F + D)             # q, k, v, o, gate, up, down: in + out
        lora_gf = 3 * 28 * 2 * L * lora_rank * io / 1e9                                   # forward + 2x backward (dX and dA / dB)
        proj_b = 0.0                                                                      # freeze_projector
    return conv + enc + proj_f + proj_b + 2 * lm_body + head + lm_w + lora_gf


def replica_report(flat_p, global_step, ms_per_step, group=None):
    """Are the data-parallel replicas still identical after the timed steps?  One MAX all-reduce of
    [s, -s, q, -q, step, -step, ms, -ms] (s / q = float64 sum / sum of squares of the flat trainable-parameter buffer) gives
    max and min of each over the ranks.  Pure torch (RCCL on GPUs, gloo in the CPU tests); a single process reports itself."""
    p = flat_p.detach().double()
    v = torch.stack([p.sum(), p.square().sum(), torch.tensor(float(global_step), dtype=torch.float64, device=p.device),
                     torch.tensor(float(ms_per_step), dtype=torch.float64, device=p.device)])
    t = torch.stack([v, -v], 1).reshape(-1).contiguous()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    t = t.cpu().tolist()
    mx, mn = t[0::2], [-x for x in t[1::2]]
    return {"replicas_identical": bool(mx[0] == mn[0] and mx[1] == mn[1] and mx[2] == mn[2]),
            "weight_checksum": {"sum_min": mn[0], "sum_max": mx[0], "sumsq_min": mn[1], "sumsq_max": mx[1]},
            "global_step": {"min": int(mn[2]), "max": int(mx[2])},
            "ms_per_step_by_rank": {"min": round(mn[3], 3), "max": round(mx[3], 3)}}


def parse_rccl_log(text):
    """Transports RCCL reports in its NCCL_DEBUG=INFO log: how many channel connections go `via P2P/...` (xGMI / PCIe peer
    access), `via SHM` (host memory) or `via NET` (sockets / IB), whether the topology dump names XGMI links, and the library
    version line.  Counts, not a verdict: the log format is RCCL's own."""
    import re
    out = {"via_p2p": len(re.findall(r"via P2P", text)), "via_shm": len(re.findall(r"via SHM", text)),
           "via_net": len(re.findall(r"via NET", text)), "xgmi_mentions": len(re.findall(r"(?i)xgmi", text)),
           "channels": None, "version_line": None}
    m = re.search(r"(\d+) coll channels", text)
    if m:
        out["channels"] = int(m.group(1))
    m = re.search(r"(?m)^.*(?:RCCL|NCCL) version[^\n]*$", text)
    if m:
        out["version_line"] = m.group(0).strip()[-120:]
    if out["via_p2p"] and not out["via_shm"] and not out["via_net"]:
        out["transport"] = "p2p (xGMI)" if out["xgmi_mentions"] else "p2p"
    elif out["via_p2p"] or out["via_shm"] or out["via_net"]:
        out["transport"] = "+".join(k[4:] for k in ("via_p2p", "via_shm", "via_net") if out[k])
    else:
        out["transport"] = "unknown (no channel lines in the log)"
    return out


def guarded(body, rank=0, world=1, out=sys.stdout):
    """Run ``body()``; an exception on this rank becomes ONE parseable JSON line with an "error" field (the launcher forwards
    every rank's stdout) and exit code 1 -- torch.distributed.run then tears the other ranks down."""
    try:
        return body()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001
        import traceback
        rec = {"metric": "training audio-sec/sec on 10s@16kHz clips", "value": None, "unit": "audio-s/s", "n_gpus": world,
               "error": f"{type(e).__name__}: {e}", "rank": rank, "traceback": traceback.format_exc()[-1500:]}
        print(json.dumps(rec), file=out, flush=True)
        raise SystemExit(1)


def relaunch_if_needed(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start one rank per GPU under
    torch.distributed.run (what the driver does itself) and hand its exit code back."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    n = a.gpus if (getattr(a, "dry_run", False) or getattr(a, "share_gpu", False)) else torch.cuda.device_count()
    if n < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {n} GPU(s) visible on this node")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    a = parse()
    relaunch_if_needed(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    guarded(lambda: run(a), rank, world)


def run(a):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dry = bool(a.dry_run)
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the ta355 hot path has no CPU fallback; --dry-run checks the plumbing only)")
    if world != max(a.gpus, 1) and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    if dry:
        dev = torch.device("cpu")
    else:
        if a.share_gpu:
            local = 0
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    rccl_ranks, rccl_log = 1, None
    if world > 1 and dry:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        probe = torch.ones(1)
        dist.all_reduce(probe)
        rccl_ranks = int(probe.item())
    elif world > 1 and a.dist_backend == "gloo":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)
        rccl_ranks = int(probe.item())
    elif world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the host driver only supports dmabuf IPC: without this RCCL fails with `hipIpcGetMemHandle: invalid argument`
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL's own account of what it connected (parsed into the line: replicas.rccl); a user's NCCL_DEBUG settings win
        if "NCCL_DEBUG" not in os.environ:
            rccl_log = f"/tmp/ta355_rccl_{os.getpid()}_r{rank}.log"
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,ENV", NCCL_DEBUG_FILE=rccl_log)
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=10))     # RCCL over xGMI
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)                                    # an actual collective: the number of ranks RCCL connected
        rccl_ranks = int(probe.item())
        if rccl_ranks != world:
            raise SystemExit(f"RCCL all-reduce saw {rccl_ranks} ranks, expected {world}")
    from tiny_audio_amd import _lib, ops
    from tiny_audio_amd.asr_config import ASRConfig
    from tiny_audio_amd.asr_modeling import ASRModel
    from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
    from tiny_audio_amd.synthetic import token_batch
    from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments

    text = dict(hidden_size=2048, intermediate_size=6144) if a.lm == "1.7b" else None
    if a.full_ft:
        a.proj_hidden = 2048 if a.proj_hidden == 1024 else a.proj_hidden
    audio, extra = None, {}
    n_samples = 160000
    if dry:                                                       # a toy model: only the control flow is exercised
        _lib.DRY_RUN = True
        audio = dict(hidden=256, ffn=512, layers=1, heads=4, n_mels=128, head_dim=64, rope_theta=10000.0, partial_rotary=0.5, ln_eps=1e-5)
        text = dict(vocab=1000, hidden=256, ffn=512, layers=2, heads=4, kv_heads=2, head_dim=128, rms_eps=1e-6, rope_theta=1e6)
        extra = dict(audio_token_id=999)
        a.proj_hidden, a.batch, a.seq_len, n_samples = 128, min(a.batch, 2), 64, 16000
        a.no_cpu_baseline = a.no_roofline = True
    mdt = {"bf16": "bfloat16", "f32": "float32"}
    cfg = ASRConfig(model_dtype=mdt[a.streams], audio_config=audio, text_config=text, freeze_language_model=not a.full_ft, projector_type=a.projector, projector_hidden_dim=a.proj_hidden, audio_token_dropout=a.dropout,
                    use_lora=a.lora, freeze_projector=a.lora, **extra)
    torch.manual_seed(0)                                          # identical frozen + projector weights on every rank
    model = ASRModel(cfg, device=dev, init="random", seed=0)
    model.train()
    fe = LogMelFeatureExtractor(128, dev)
    # N > 1: the flat-gradient all-reduce is launched asynchronously and its optimizer update applied after the NEXT step's
    # frozen-encoder forward (trainer.py), so the collective runs under ~half a step of compute
    overlap = world > 1 and not a.sync_allreduce
    trainer = ASRTrainer(model, TrainingArguments(learning_rate=1e-3, weight_decay=0.0, max_grad_norm=1.0,
                                                  warmup_steps=500, max_steps=50000, lr_scheduler_type="polynomial",
                                                  lr_scheduler_kwargs={"power": 0.5}),
                         decoder_learning_rate=1e-4 if a.full_ft else None, overlap_allreduce=overlap, time_allreduce=world > 1)
    B, L, V = a.batch, a.seq_len, cfg.text_config.vocab_size
    # synthetic inputs of SURVEY.md 8(d), resident in HBM: wav = 0.1 * N(0,1), 160000 samples per clip
    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    wav = 0.1 * torch.randn(B, n_samples, device=dev, generator=g)
    if rank == 0:                                                 # clip 0 = SURVEY 8(d)'s numpy clip: the one cpu_baseline's oracle runs
        from tiny_audio_amd.synthetic import synthetic_wave
        wav[0].copy_(torch.from_numpy(synthetic_wave(0, n_samples)))
    lens = torch.full((B,), n_samples, device=dev, dtype=torch.int64)
    n_audio = int(model.projector.get_output_length(n_samples // 320))   # 125 for the frame-stacking projectors, 102 for the QFormer
    ids, att, lab, counts, n_lab = token_batch(B, n_audio, V, cfg.audio_token_id, 990 if dry else cfg.pad_token_id, 991 if dry else cfg.eos_token_id, L=L,
                                               **(dict(n_suffix=4, n_text=10) if dry else {}))
    ids_d, att_d, lab_d = (torch.from_numpy(x).to(dev) for x in (ids, att, lab))
    counts_d = torch.from_numpy(counts).to(dev)

    host = {}

    def step(full_logits=False, from_host=False):
        if from_host:                                             # the collator's hand-over: raw waveforms + token tensors in pinned host memory
            wav_s = host["wav"].to(dev, non_blocking=True)
            ids_s, att_s, lab_s, cnt_s = (host[k].to(dev, non_blocking=True) for k in ("ids", "att", "lab", "cnt"))
            feats, _mask = fe.extract(wav_s, lens)
            rows, tg, _n = ops.label_rows(lab_s)
            trainer.training_step(dict(input_ids=ids_s, input_features=feats, attention_mask=att_s, labels=lab_s,
                                       audio_token_counts=cnt_s, label_meta=(rows, tg, n_lab)), return_logits=full_logits)
            return
        feats, _mask = fe.extract(wav, lens)                      # K1 on the GPU, inside the timed step
        # label positions: the device kernel runs inside the step; their COUNT is host knowledge of whoever built the
        # labels (the collator builds them on the CPU), which saves the one device->host sync of the reference's path
        rows, tg, _n = ops.label_rows(lab_d)
        batch = dict(input_ids=ids_d, input_features=feats, attention_mask=att_d, labels=lab_d,
                     audio_token_counts=counts_d, label_meta=(rows, tg, n_lab))
        trainer.training_step(batch, return_logits=full_logits)

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    def timed(n_steps, n_warm, **kw):
        for _ in range(max(n_warm - 1, 0)):
            step(**kw)
        trainer.flush()
        # Python's cyclic collector: a generation-2 pass over the ~10^5 objects of the two frozen models takes 15-55 ms, and the garbage
        # of model construction trips its threshold a few steps into the run -- one 56-98 ms step in an 8-step window (measured on the
        # MoE configuration, profiles/r04_zl_moe_step_times_gc.txt: 43.5-47.1 ms per step with it, 42.5 without).  Collect now and move
        # what exists to the permanent generation, as training scripts do after building the model: the timed steps then measure the
        # steady state (in a long run such a pass amortises to < 0.1 %).
        import gc
        gc.collect()
        gc.freeze()
        if n_warm >= 1:                                           # the last warm-up step runs AFTER the collection (the caching allocator
            step(**kw)                                            # re-settles there: 8 device allocations and +2.7 ms otherwise land in
            trainer.flush()                                       # the first timed step)
        trainer.allreduce_exposed_ms()                            # reset the event list
        barrier()
        evs = []
        t0 = time.perf_counter()
        for _ in range(n_steps):
            if a.step_times and dev.type == "cuda":
                e0 = torch.cuda.Event(enable_timing=True); e0.record(); evs.append(e0)
            step(**kw)
        trainer.flush()                                           # every optimizer update of the K steps is inside the timed region
        if evs:
            e1 = torch.cuda.Event(enable_timing=True); e1.record(); evs.append(e1)
        barrier()
        dt = time.perf_counter() - t0
        if evs:
            step_ms[:] = [round(evs[i].elapsed_time(evs[i + 1]), 3) for i in range(len(evs) - 1)]
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    full = a.logits == "full"
    step_ms = []
    dev_allocs0 = torch.cuda.memory_stats().get("num_device_alloc", 0) if (a.step_times and dev.type == "cuda") else 0
    from tiny_audio_amd import trainer as _trainer_mod
    # data-path collectives: the trainer issues one flat all-reduce per optimizer step whatever the configuration (MoE's auxiliary
    # shadows, LoRA's / the fine-tuned LM's gradients all live in the one buffer); counted over warm-up + timed steps, whose last
    # deferred collective `flush` completes inside the region
    coll0 = _trainer_mod.COLLECTIVES["allreduce_flat"]
    dt = timed(a.steps, a.warmup, full_logits=full)
    coll_per_step = (_trainer_mod.COLLECTIVES["allreduce_flat"] - coll0) / max(a.steps + a.warmup, 1)
    step_ms_main = list(step_ms)
    if a.step_times and dev.type == "cuda":
        step_ms_main.append({"device_allocs_in_run": torch.cuda.memory_stats().get("num_device_alloc", 0) - dev_allocs0})
    ar_ms = trainer.allreduce_exposed_ms() / max(a.steps, 1) if world > 1 else 0.0
    loss = trainer.last_loss()
    ms = dt / a.steps * 1e3
    value = world * B * 10.0 * a.steps / dt
    # N > 1: (i) are the replicas still identical (weights, step count), who was slow; (ii) the other all-reduce mode in the same
    # invocation (a short leg): what the deferred update buys
    replicas, ar_other = None, None
    if world > 1:
        kr = max(2, min(a.steps, 4))
        sync(); t_local = time.perf_counter()                     # this rank's own clock, no barrier: who is slow
        for _ in range(kr):
            step(full_logits=full)
        trainer.flush(); sync()
        replicas = replica_report(trainer.flat.flat_p, trainer.global_step, (time.perf_counter() - t_local) / kr * 1e3)
        trainer.flush()
        trainer.overlap_allreduce = not overlap
        k4 = max(2, min(a.steps, 8))
        dt4 = timed(k4, 1, full_logits=full)
        ar_other = {"mode": "synchronous" if overlap else "async / deferred update", "steps": k4, "ms_per_step": round(dt4 / k4 * 1e3, 3),
                    "ms_exposed_per_step": round(trainer.allreduce_exposed_ms() / k4, 4)}
        trainer.flush()
        trainer.overlap_allreduce = overlap

    # the same step with the reference's materialised outputs.logits [B, L, V] (bf16), on record beside the default line
    logits_full = None
    if not full and not a.no_logits_full:
        k2 = max(1, a.steps)
        dt2 = timed(k2, max(2, a.warmup), full_logits=True)       # warm-ups: the 1.9 GB logits buffer is first-touched outside the timed region
        logits_full = {"ms_per_step": round(dt2 / k2 * 1e3, 3), "value": round(world * B * 10.0 * k2 / dt2, 1), "steps": k2,
                       "warmup": max(2, a.warmup),
                       "note": "additionally writes outputs.logits [B, L, V] bf16 every step, as the reference's forward does"}
        trainer.last_logits = None

    # the same step with the residual streams stored in the OTHER dtype (config.model_dtype is read at every forward)
    streams_other = None
    if not a.no_logits_full and not dry:
        other = "f32" if a.streams == "bf16" else "bf16"
        cfg.model_dtype = mdt[other]
        k5 = max(1, a.steps)
        dt5 = timed(k5, max(2, a.warmup), full_logits=full)
        streams_other = {"streams": other, "ms_per_step": round(dt5 / k5 * 1e3, 3), "value": round(world * B * 10.0 * k5 / dt5, 1),
                         "steps": k5, "warmup": max(2, a.warmup), "final_loss": round(trainer.last_loss(), 4),
                         "note": "encoder / LM residual streams, the LM tape and d(x) stored in %s; same MFMA operand / accumulator types" % other}
        cfg.model_dtype = mdt[a.streams]
        step(full_logits=full); trainer.flush()                   # back in the headline's mode for the roofline / parity legs below

    # the same step fed from HOST buffers (20.5 MB of f32 waveforms + the token tensors per step over PCIe, pinned, same stream):
    # what the boundary costs when the dataloader hands over host memory.  Never `value`.
    host_inputs = None
    if not a.no_logits_full and not dry:
        host.update(wav=wav.cpu().pin_memory(), ids=ids_d.cpu().pin_memory(), att=att_d.cpu().pin_memory(),
                    lab=lab_d.cpu().pin_memory(), cnt=counts_d.cpu().pin_memory())
        k3 = max(1, a.steps)
        dt3 = timed(k3, max(2, a.warmup), full_logits=full, from_host=True)
        host_inputs = {"ms_per_step": round(dt3 / k3 * 1e3, 3), "value": round(world * B * 10.0 * k3 / dt3, 1), "steps": k3,
                       "warmup": max(2, a.warmup),
                       "note": "inputs copied from pinned host memory inside every step (PCIe-inclusive rate)"}
        host.clear()

    roofline = None
    if not a.no_roofline:
        # EVERY rank runs the two extra (untimed) steps -- they contain the gradient all-reduce -- but only rank 0
        # records: HIP events around each GEMM launch on the launch stream (csrc/gemm.hip: ta_profile_gemm).
        import ctypes as C
        lib = _lib.lib()
        if rank == 0:
            lib.ta_profile_gemm(1)
        for _ in range(2):
            step(full_logits=full)
        trainer.flush()
        barrier()
        if rank == 0:
            lib.ta_profile_gemm(0)
            tms, tfl, nl = C.c_double(), C.c_double(), C.c_long()
            lib.ta_profile_gemm_collect(C.byref(tms), C.byref(tfl), C.byref(nl))
            achieved = tfl.value / (tms.value * 1e-3) / 1e12 if tms.value > 0 else 0.0
            # HBM-side bytes per GEMM launch: PMC counters cannot be read from inside this process, so the value is the one
            # measured on THIS workload with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes over bench.py at the
            # default configuration, FETCH x 2 per the guide's gfx950 correction: scripts/gpu_pmc.sh + summarize_pmc.py,
            # committed as profiles/pmc_gemm_traffic_b32_mlp.json); null for any other configuration.
            traffic, traffic_src = None, None
            tj_name = "pmc_gemm_traffic_b32_mlp.json" if a.streams == "bf16" else "pmc_gemm_traffic_b32_mlp_f32.json"
            tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", tj_name)
            default_cfg = (B == 32 and a.projector == "mlp" and not a.lora and not a.full_ft and a.proj_hidden == 1024
                           and a.lm == "0.6b" and a.seq_len == 192)
            if default_cfg and os.path.exists(tj):
                with open(tj) as fh:
                    tr = json.load(fh)
                if abs(tr["launches"] / 2 - nl.value / 2) <= 2:            # same launch count per step as the measured run
                    traffic = round((tr["fetch_mb_per_launch"] + tr["write_mb_per_launch"]) * 1e6)
                    traffic_src = f"profiles/{tj_name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)"
            roofline = {"kernel": "gemm_nt_kernel (bf16 MFMA 16x16x32, all tile / epilogue variants)", "bound": "mfma",
                        "achieved": round(achieved, 1), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "traffic_measured_in_this_run": False,      # PMC counters need rocprofv3 around the process: read from the committed pass
                        "launches_per_step": nl.value // 2, "avg_launch_us": round(tms.value * 1e3 / max(nl.value, 1), 2),
                        "gemm_ms_per_step": round(tms.value / 2, 3),
                        "algorithmic_gflop_per_launch": round(tfl.value / max(nl.value, 1) / 1e9, 3),
                        "hbm_kernels": hbm_kernel_rates(B, L, cfg, fe, wav, lens, a.streams)}

    cpu, parity = None, None
    if not a.no_cpu_baseline and rank == 0 and world == 1 and a.projector == "mlp" and not a.lora and a.proj_hidden == 1024 and a.lm == "0.6b" and not a.full_ft:
        # Untimed: clip 0 alone through the GPU path with the frame dropout off (the timed steps drop 10 % of the frames with a
        # device RNG the oracle cannot replay), against the oracle's loss for the same clip and weights (cpu_baseline computes it).
        # The trainer has updated the projector since step 0: both sides read the CURRENT weights.
        with torch.no_grad():
            f0, _ = fe.extract(wav[:1].contiguous(), lens[:1].contiguous())
            o0 = model(input_ids=ids_d[:1], input_features=f0, attention_mask=att_d[:1], labels=lab_d[:1],
                       audio_token_counts=counts_d[:1], return_logits=False, frame_keep=torch.ones(500, device=dev))
            gpu_loss0 = float(o0.loss)
        cpu = cpu_baseline(model, cfg, L)
        ol = cpu.pop("_loss")
        parity = {"clip": "clip 0 of the batch = synthetic_wave(0), dropout off, current weights",
                  "loss_gpu": round(gpu_loss0, 5), "loss_oracle_fp32": round(ol, 5), "rel_diff": round(abs(gpu_loss0 - ol) / ol, 6),
                  "stated_tolerance": 1e-2}

    if rank == 0:
        D_, F_ = cfg.text_config.hidden_size, cfg.text_config.intermediate_size
        gf = algorithmic_gflop_per_clip(L, V, n_lab // B, full, H=a.proj_hidden, D=D_, F=F_, full_ft=a.full_ft,
                                            projector=a.projector, lora=a.lora, n_audio=n_audio)
        regime = ("fp32 modules under bf16 autocast = the +experiments=transcription recipe (fp32 residual streams / tape / d(x), bf16 MFMA "
                  "operands, fp32 accumulate)" if a.streams == "f32" else
                  "bf16 modules (ASRConfig's default model_dtype; NOT the fp32+autocast recipe: bf16 residual streams)")
        rec = {"metric": "training audio-sec/sec on 10s@16kHz clips", "value": round(value, 1), "unit": "audio-s/s", "regime": regime,
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", **({"dry_run": True} if dry else {}),
               **({"validation_only": "ranks share one GPU over gloo: the N > 1 control flow with real kernels, not a measurement"}
                  if (a.share_gpu or (world > 1 and a.dist_backend == "gloo" and not dry)) else {}),
               "data": "synthetic (0.1*N(0,1) waveforms, random-init weights at true shapes; the same batch every step)",
               "config": {"workload": ("embedded.yaml: full decoder fine-tuning, MLP projector (H=%d) + every LM weight" % a.proj_hidden if a.full_ft else
                                       "configs[4]: stage 2, frozen MLP projector + LoRA r=8 alpha=32 on q,k,v,o,gate,up,down" if a.lora
                                       else ("configs[1]: MLP projector (H=%d, D=%d)" % (a.proj_hidden, D_)) if a.projector == "mlp" else
                                       "QFormer projector (2 layers, 16 heads, windows of 15 -> 3 queries, 102 audio tokens)" if a.projector == "qformer" else
                                       "MOSA projector (2 stride-2 convs, 4 dense experts of width 4096)" if a.projector == "mosa" else
                                       "configs[3]: shared+sparse MoE projector (4 experts, top-2, H=D=1024)") +
                                      " bf16, GLM-ASR-Nano encoder 32L + Qwen3-%s 28L, "
                                      "10 s / 16 kHz clips, L=%d, %d label tokens/clip" % (a.lm.upper(), L, n_lab // B),
                          "clips_per_gpu": B, "global_batch": world * B, "seq_len": L, "parallelism": f"dp{world}",
                          "logits": a.logits, "streams": a.streams, "audio_token_dropout": a.dropout,
                          "algorithmic_gflop_per_clip": round(gf, 1),
                          "step_tflops": round(gf * world * B / (ms * 1e-3) / 1e3, 1)},
               "final_loss": round(loss, 4),
               **({"step_ms": step_ms_main} if a.step_times else {}),
               "rccl_ranks": rccl_ranks,
               "allreduce": None if world == 1 else {
                   "ms_exposed_per_step": round(ar_ms, 4), "elements": trainer.flat.flat_g.numel(),
                   "bytes": 4 * trainer.flat.flat_g.numel(), "other_mode": ar_other,
                   "collectives_per_step": coll_per_step,
                   "mode": "async on RCCL's stream, update applied after the next step's frozen-encoder forward" if overlap
                           else "synchronous on the compute stream",
                   "note": "events on the compute stream around the collective (sync) / around the wait for it (async)"},
               "replicas": replicas,
               "logits_full": logits_full, "streams_other": streams_other, "host_inputs": host_inputs, "roofline": roofline,
               "cpu_baseline": cpu, "parity": parity, "numerics": numerics_contract(a.streams)}
        if replicas is not None:
            replicas["rccl_version"] = ("gloo (dry run)" if dry else "gloo (--dist-backend gloo: not RCCL)" if a.dist_backend == "gloo"
                                        else ".".join(str(x) for x in torch.cuda.nccl.version()))
            if rccl_log and os.path.exists(rccl_log):
                with open(rccl_log, errors="replace") as fh:
                    replicas["rccl"] = parse_rccl_log(fh.read())
            replicas["env"] = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "NCCL_P2P_DISABLE", "NCCL_SHM_DISABLE")}
        print(json.dumps(rec), flush=True)
        if replicas is not None and not replicas["replicas_identical"]:
            raise RuntimeError(f"data-parallel replicas diverged: {replicas}")
    if world > 1:
        dist.destroy_process_group()


def numerics_contract(streams):
    """What the line's arithmetic is, and how far the REFERENCE's own bf16 regimes sit from its fp32 run at the benchmarked shape
    (tests/golden/asr_full_recipe.npz, generated by tests/golden/make_golden.py from /root/reference; the parity tests gate the
    HIP path's distance from the same fp32 run against these: tests/test_gpu_round5.py)."""
    out = {"mfma_operands": "bf16", "accumulate": "fp32", "norm_softmax_ce": "fp32", "trainable_masters": "fp32",
           "residual_stream_storage": streams,
           "mirrors": "fp32 modules + bf16 autocast (configs/config.yaml:14-18, production.yaml:49)" if streams == "f32"
                      else "bf16 modules (ASRConfig model_dtype default, tiny_audio/asr_config.py:41)"}
    fx = os.path.join(ROOT, "tests", "golden", "asr_full_recipe.npz")
    if os.path.exists(fx):
        g = np.load(fx)
        for mode in ("autocast", "bf16"):
            out[f"reference_{mode}_vs_reference_fp32"] = {
                "logits_maxabs": round(float(g[f"{mode}.logits_maxabs_vs_fp32"]), 4), "logits_rms": round(float(g[f"{mode}.logits_rms_vs_fp32"]), 4),
                "nll_maxabs": round(float(g[f"{mode}.nll_maxabs_vs_fp32"]), 4),
                "grad_cos_min": round(min(float(g[k]) for k in g.files if k.startswith(f"{mode}.gcos_vs_fp32.")), 5)}
    return out


def hbm_kernel_rates(B, L, cfg, fe, wav, lens, streams="f32"):
    """Achieved GB/s of the HBM-bound kernels of the step (north_star: feature / norm kernels against the HBM
    roofline), each timed over 20 launches with events on the launch stream; bytes are ALGORITHMIC."""
    import torch
    from tiny_audio_amd import ops
    dev = wav.device
    PEAK = 8000.0

    filler_a = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)

    def rate(fn, nbytes, reps=20):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # These kernels take 5-50 us and a Python-level launch 10-60 us: timed bare, the events measure the HOST's pace (round 3: RMSNorm
        # 0.29 of the roofline stand-alone against 0.47 inside the step).  ~3 ms of filler GEMMs go first, so that every timed launch
        # is already queued when the GPU reaches the first event: the events then bracket back-to-back kernel time only.
        for _ in range(4):
            torch.mm(filler_a, filler_a)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        return {"avg_us": round(us, 2), "achieved": round(nbytes / us / 1e3, 1), "peak": PEAK, "unit": "GB/s",
                "frac": round(nbytes / us / 1e3 / PEAK, 4), "algorithmic_mb": round(nbytes / 1e6, 2)}

    Me, H = B * 500, cfg.audio_config.hidden_size
    Ml, D, F = B * L, cfg.text_config.hidden_size, cfg.text_config.intermediate_size
    sdt, sb = (torch.float32, 4) if streams == "f32" else (torch.bfloat16, 2)      # the residual stream's storage in THIS run
    xe = torch.randn(Me, H, device=dev).to(sdt); we, be = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    xl = torch.randn(Ml, D, device=dev).to(sdt); wl = torch.ones(D, device=dev)
    gu = torch.randn(Ml, 2 * F, device=dev).to(torch.bfloat16)
    out = {f"layernorm_kernel (encoder, {streams} residual stream in -> bf16 out: the variant the step runs)": rate(lambda: ops.layernorm(xe, we, be), Me * H * (sb + 2)),
           f"rmsnorm_fwd_kernel (LM, {streams} residual stream in -> bf16 out: the variant the step runs)": rate(lambda: ops.rmsnorm_fwd(xl, wl), Ml * D * (sb + 2) + Ml * 4),
           "swiglu_fwd_kernel (LM, bf16 gate|up -> bf16)": rate(lambda: ops.swiglu_fwd(gu, F), Ml * F * 6),
           "logmel (f32 wav -> f32 [128, 1000]: persistent mixed-radix 16x25 FFT / mel kernel + finalize pass, both launches)": rate(lambda: fe.extract(wav, lens), B * (640000 + 512000))}
    return out


def cpu_baseline(model, cfg, L, reps=5, warmups=2):
    """The numpy oracle (a 'port': the reference's Python cannot travel) timed on the host cores on a bounded
    sample: ONE 10 s clip, one full-depth training step (forward + backward), fp32, same weights as the GPU model;
    ``warmups`` untimed passes, then the median of ``reps`` passes (median of >= 5 after 2 warm-ups)."""
    from oracle import features as OF
    from oracle import model as OM
    from oracle import weights as OW
    enc, lm = cfg.audio_config, cfg.text_config
    ecfg = OW.enc_config(enc.hidden_size, enc.intermediate_size, enc.num_hidden_layers, enc.num_attention_heads)
    lcfg = OW.lm_config(lm.vocab_size, lm.hidden_size, lm.intermediate_size, lm.num_hidden_layers, lm.num_attention_heads,
                        lm.num_key_value_heads, lm.head_dim, lm.rms_norm_eps, lm.rope_theta)
    W = dict(encoder=model.audio_tower.export_state_dict_hf(), lm=model.language_model.export_state_dict_hf(),
             projector={k: v.detach().float().cpu().numpy() for k, v in model.projector.state_dict().items()})
    ocfg = dict(enc=ecfg, lm=lcfg, projector_type="mlp", k=4, audio_token_id=cfg.audio_token_id)
    ids, att, lab, counts = OW.synthetic_tokens(1, 125, lm.vocab_size, cfg.audio_token_id, cfg.pad_token_id,
                                                cfg.eos_token_id, L=L)
    times, loss = [], 0.0
    for i in range(reps + warmups):
        t0 = time.perf_counter()
        wav, lens = OF.pad_batch([OW.synthetic_wave(0)])
        feats, _ = OF.log_mel(wav, lens)
        batch = dict(input_ids=ids, attention_mask=att, labels=lab, input_features=feats, audio_token_counts=counts)
        out = OM.asr_forward(batch, W, ocfg, training=True)
        OM.asr_backward(out, W, ocfg)
        if i >= warmups:
            times.append(time.perf_counter() - t0)
        loss = float(out["loss"])
    dt = sorted(times)[len(times) // 2]
    # BASELINE.md section 4 planned the baseline at B = 2 as well: ONE more pass over two clips (the numpy oracle is warm by now)
    ids2, att2, lab2, counts2 = OW.synthetic_tokens(2, 125, lm.vocab_size, cfg.audio_token_id, cfg.pad_token_id, cfg.eos_token_id, L=L)
    t0 = time.perf_counter()
    wav2, lens2 = OF.pad_batch([OW.synthetic_wave(0), OW.synthetic_wave(1)])
    feats2, _ = OF.log_mel(wav2, lens2)
    out2 = OM.asr_forward(dict(input_ids=ids2, attention_mask=att2, labels=lab2, input_features=feats2, audio_token_counts=counts2),
                          W, ocfg, training=True)
    OM.asr_backward(out2, W, ocfg)
    dt2 = time.perf_counter() - t0
    # `cores` = the BLAS threads numpy actually ran its GEMMs on (OpenBLAS caps its pool below the core count on large hosts), not
    # the host's core count; the element-wise numpy code around them is single-threaded
    cores, host_cores = os.cpu_count(), os.cpu_count()
    try:
        import threadpoolctl
        pools = [d.get("num_threads") for d in threadpoolctl.threadpool_info() if d.get("user_api") == "blas" and d.get("num_threads")]
        if pools:
            cores = max(pools)
    except Exception:  # noqa: BLE001
        pass
    return {"value": round(10.0 / dt, 3), "unit": "audio-s/s", "cores": cores, "host_cores": host_cores, "kind": "port", "_loss": loss,
            "b2": {"value": round(20.0 / dt2, 3), "unit": "audio-s/s", "sample": f"2 clips x 1 step, one pass, {dt2:.1f} s"},
            "sample": "1 clip x 1 full-depth training step (log-mel + fwd + bwd, fp32 numpy/OpenBLAS oracle): median of "
                      f"{reps} passes after {warmups} warm-ups, {dt:.1f} s each ({min(times):.1f}-{max(times):.1f}), loss {loss:.4f}"}


if __name__ == "__main__":
    main()
