#!/usr/bin/env python3
"""bench.py -- training-step audio-seconds/second of the tiny-audio projector-training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  The driver launches the ranks itself (python -m torch.distributed.run ... bench.py
--gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment); a plain `python bench.py --gpus N`
re-launches itself the same way.  The line then carries `n_gpus` = N, `rccl_ranks` (counted by an actual all-reduce)
and `allreduce.ms_exposed_per_step`.

One "step" = one full optimizer step of BASELINE.json configs[1]: B synthetic 10 s / 16 kHz clips per GPU ->
GPU log-mel -> frozen GLM-ASR encoder (32 layers) -> MLP projector (H=D=1024) -> frozen Qwen3-0.6B (28 layers,
L=192 tokens, 125 <audio> placeholders, 36 label tokens) -> shifted CE -> backward through LM (dX) and
projector (dW) -> flat gradient all-reduce (RCCL, N>1) -> global-norm clip + AdamW.  Inputs (waveforms, token
ids) are resident in HBM before the timed region; random-init weights at the true shapes (no checkpoints offline).

The single JSON line carries `roofline` (dominant kernel = the MFMA GEMM, timed in situ with HIP events on
its launch stream), `host_inputs` (the same step fed from pinned host memory: the PCIe-inclusive rate, never `value`),
`logits_full` (the same step with the reference's [B, L, V] logits materialised, timed in the same
run) and, at N=1, `cpu_baseline` (the numpy oracle timed on the host cores on one clip: median of 3 after a warm-up).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

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
    return ap.parse_args()


def algorithmic_gflop_per_clip(L, V, n_label, full_logits, H=1024, D=1024, F=3072, full_ft=False):
    """BASELINE.md section 3 (2*MAC, dense).  lm_head is counted at the positions actually computed."""
    conv = 0.983 + 4.915
    enc = 32 * (2 * 500 * 4 * 1280 ** 2 + 4 * 500 ** 2 * 1280 + 4 * 500 * 1280 * 5120) / 1e9
    proj_f = 2 * 125 * (5120 * H + H * D) / 1e9
    proj_b = 2 * 125 * (2 * 5120 * H + 2 * H * D) / 1e9 - 2 * 125 * 5120 * H / 1e9        # dW1, dW2, dA1 (no dX)
    lm_body = 28 * (2 * L * (D * 2048 + 2 * D * 1024 + 2048 * D + 3 * D * F) + 2 * L * L * 2048) / 1e9   # 16 q / 8 kv heads x 128
    head_rows = (L if full_logits else 0) + n_label
    head = 2 * head_rows * D * V / 1e9 + 2 * n_label * D * V / 1e9                        # fwd (+ labelled dH backward)
    lm_w = 0.0
    if full_ft:     # weight gradients: one more pass over every linear (no attention term) + the tied head's dE at the labelled rows
        lm_w = 28 * (2 * L * (D * 2048 + 2 * D * 1024 + 2048 * D + 3 * D * F)) / 1e9 + 2 * n_label * D * V / 1e9
    return conv + enc + proj_f + proj_b + 2 * lm_body + head + lm_w


def relaunch_if_needed(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start one rank per GPU under
    torch.distributed.run (what the driver does itself) and hand its exit code back."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    n = torch.cuda.device_count()
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
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the ta355 hot path has no CPU fallback)")
    if world != max(a.gpus, 1) and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)            # RCCL over xGMI
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
    cfg = ASRConfig(text_config=text, freeze_language_model=not a.full_ft, projector_type=a.projector, projector_hidden_dim=a.proj_hidden, audio_token_dropout=a.dropout,
                    use_lora=a.lora, freeze_projector=a.lora)
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
    wav = 0.1 * torch.randn(B, 160000, device=dev, generator=g)
    lens = torch.full((B,), 160000, device=dev, dtype=torch.int64)
    n_audio = int(model.projector.get_output_length(500))         # 125 for the frame-stacking projectors, 102 for the QFormer
    ids, att, lab, counts, n_lab = token_batch(B, n_audio, V, cfg.audio_token_id, cfg.pad_token_id, cfg.eos_token_id, L=L)
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
        torch.cuda.synchronize()

    def timed(n_steps, n_warm, **kw):
        for _ in range(n_warm):
            step(**kw)
        trainer.flush()
        trainer.allreduce_exposed_ms()                            # reset the event list
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(**kw)
        trainer.flush()                                           # every optimizer update of the K steps is inside the timed region
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    full = a.logits == "full"
    dt = timed(a.steps, a.warmup, full_logits=full)
    ar_ms = trainer.allreduce_exposed_ms() / max(a.steps, 1) if world > 1 else 0.0
    loss = trainer.last_loss()
    ms = dt / a.steps * 1e3
    value = world * B * 10.0 * a.steps / dt

    # the same step with the reference's materialised outputs.logits [B, L, V] (bf16), on record beside the default line
    logits_full = None
    if not full and not a.no_logits_full:
        k2 = max(1, min(a.steps, 4))
        dt2 = timed(k2, 1, full_logits=True)
        logits_full = {"ms_per_step": round(dt2 / k2 * 1e3, 3), "value": round(world * B * 10.0 * k2 / dt2, 1), "steps": k2,
                       "note": "additionally writes outputs.logits [B, L, V] bf16 every step, as the reference's forward does"}
        trainer.last_logits = None

    # the same step fed from HOST buffers (20.5 MB of f32 waveforms + the token tensors per step over PCIe, pinned, same stream):
    # what the boundary costs when the dataloader hands over host memory.  Never `value`.
    host_inputs = None
    if not a.no_logits_full:
        host.update(wav=wav.cpu().pin_memory(), ids=ids_d.cpu().pin_memory(), att=att_d.cpu().pin_memory(),
                    lab=lab_d.cpu().pin_memory(), cnt=counts_d.cpu().pin_memory())
        k3 = max(1, min(a.steps, 4))
        dt3 = timed(k3, 1, full_logits=full, from_host=True)
        host_inputs = {"ms_per_step": round(dt3 / k3 * 1e3, 3), "value": round(world * B * 10.0 * k3 / dt3, 1), "steps": k3,
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
            tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_gemm_traffic_b32_mlp.json")
            default_cfg = (B == 32 and a.projector == "mlp" and not a.lora and not a.full_ft and a.proj_hidden == 1024
                           and a.lm == "0.6b" and a.seq_len == 192)
            if default_cfg and os.path.exists(tj):
                with open(tj) as fh:
                    tr = json.load(fh)
                if abs(tr["launches"] / 2 - nl.value / 2) <= 2:            # same launch count per step as the measured run
                    traffic = round((tr["fetch_mb_per_launch"] + tr["write_mb_per_launch"]) * 1e6)
                    traffic_src = "profiles/pmc_gemm_traffic_b32_mlp.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)"
            roofline = {"kernel": "gemm_nt_kernel (bf16 MFMA 16x16x32, all tile / epilogue variants)", "bound": "mfma",
                        "achieved": round(achieved, 1), "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_BF16_DENSE_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "launches_per_step": nl.value // 2, "avg_launch_us": round(tms.value * 1e3 / max(nl.value, 1), 2),
                        "gemm_ms_per_step": round(tms.value / 2, 3),
                        "algorithmic_gflop_per_launch": round(tfl.value / max(nl.value, 1) / 1e9, 3),
                        "hbm_kernels": hbm_kernel_rates(B, L, cfg, fe, wav, lens)}

    cpu = None
    if not a.no_cpu_baseline and rank == 0 and world == 1 and a.projector == "mlp" and not a.lora and a.proj_hidden == 1024 and a.lm == "0.6b" and not a.full_ft:
        cpu = cpu_baseline(model, cfg, L)

    if rank == 0:
        D_, F_ = cfg.text_config.hidden_size, cfg.text_config.intermediate_size
        gf = algorithmic_gflop_per_clip(L, V, n_lab // B, full, H=a.proj_hidden, D=D_, F=F_, full_ft=a.full_ft)
        rec = {"metric": "training audio-sec/sec on 10s@16kHz clips", "value": round(value, 1), "unit": "audio-s/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
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
                          "logits": a.logits, "audio_token_dropout": a.dropout,
                          "algorithmic_gflop_per_clip": round(gf, 1),
                          "step_tflops": round(gf * world * B / (ms * 1e-3) / 1e3, 1)},
               "final_loss": round(loss, 4),
               "rccl_ranks": rccl_ranks,
               "allreduce": None if world == 1 else {
                   "ms_exposed_per_step": round(ar_ms, 4), "elements": trainer.flat.n + trainer.flat.EXTRA,
                   "mode": "async on RCCL's stream, update applied after the next step's frozen-encoder forward" if overlap
                           else "synchronous on the compute stream",
                   "note": "events on the compute stream around the collective (sync) / around the wait for it (async)"},
               "logits_full": logits_full, "host_inputs": host_inputs, "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


def hbm_kernel_rates(B, L, cfg, fe, wav, lens):
    """Achieved GB/s of the HBM-bound kernels of the step (north_star: feature / norm kernels against the HBM
    roofline), each timed over 20 launches with events on the launch stream; bytes are ALGORITHMIC."""
    import torch
    from tiny_audio_amd import ops
    dev = wav.device
    PEAK = 8000.0

    def rate(fn, nbytes, reps=20):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        return {"avg_us": round(us, 2), "achieved": round(nbytes / us / 1e3, 1), "peak": PEAK, "unit": "GB/s",
                "frac": round(nbytes / us / 1e3 / PEAK, 4), "algorithmic_mb": round(nbytes / 1e6, 2)}

    Me, H = B * 500, cfg.audio_config.hidden_size
    Ml, D, F = B * L, cfg.text_config.hidden_size, cfg.text_config.intermediate_size
    xe = torch.randn(Me, H, device=dev).to(torch.bfloat16); we, be = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    xl = torch.randn(Ml, D, device=dev).to(torch.bfloat16); wl = torch.ones(D, device=dev)
    gu = torch.randn(Ml, 2 * F, device=dev).to(torch.bfloat16)
    out = {"layernorm_kernel (encoder, bf16 residual stream in -> bf16 out)": rate(lambda: ops.layernorm(xe, we, be), Me * H * 4),
           "rmsnorm_fwd_kernel (LM, bf16 residual stream in -> bf16 out: the variant the step runs)": rate(lambda: ops.rmsnorm_fwd(xl, wl), Ml * D * 4 + Ml * 4),
           "swiglu_fwd_kernel (LM, bf16 gate|up -> bf16)": rate(lambda: ops.swiglu_fwd(gu, F), Ml * F * 6),
           "logmel (f32 wav -> f32 [128, 1000]; mixed-radix 16x25 FFT, two frames per transform; LDS-latency-bound)": rate(lambda: fe.extract(wav, lens), B * (640000 + 512000), reps=5)}
    return out


def cpu_baseline(model, cfg, L, reps=3):
    """The numpy oracle (a 'port': the reference's Python cannot travel) timed on the host cores on a bounded
    sample: ONE 10 s clip, one full-depth training step (forward + backward), fp32, same weights as the GPU model;
    one untimed warm-up pass, then the median of ``reps`` passes."""
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
    for i in range(reps + 1):
        t0 = time.perf_counter()
        wav, lens = OF.pad_batch([OW.synthetic_wave(0)])
        feats, _ = OF.log_mel(wav, lens)
        batch = dict(input_ids=ids, attention_mask=att, labels=lab, input_features=feats, audio_token_counts=counts)
        out = OM.asr_forward(batch, W, ocfg, training=True)
        OM.asr_backward(out, W, ocfg)
        if i > 0:
            times.append(time.perf_counter() - t0)
        loss = float(out["loss"])
    dt = sorted(times)[len(times) // 2]
    return {"value": round(10.0 / dt, 3), "unit": "audio-s/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "1 clip x 1 full-depth training step (log-mel + fwd + bwd, fp32 numpy/OpenBLAS oracle): median of "
                      f"{reps} passes after 1 warm-up, {dt:.1f} s each ({min(times):.1f}-{max(times):.1f}), loss {loss:.4f}"}


if __name__ == "__main__":
    main()
