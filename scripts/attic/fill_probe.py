import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tiny_audio_amd.asr_config import ASRConfig
from tiny_audio_amd.asr_modeling import ASRModel
from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
from tiny_audio_amd.synthetic import token_batch
from tiny_audio_amd.trainer import ASRTrainer, TrainingArguments
from tiny_audio_amd import ops
dev = torch.device("cuda", 0)
cfg = ASRConfig(audio_token_dropout=0.1)
model = ASRModel(cfg, device=dev, init="random", seed=0); model.train()
fe = LogMelFeatureExtractor(128, dev)
tr = ASRTrainer(model, TrainingArguments(learning_rate=1e-3))
B, L, V = 32, 192, cfg.text_config.vocab_size
wav = 0.1 * torch.randn(B, 160000, device=dev); lens = torch.full((B,), 160000, device=dev, dtype=torch.int64)
ids, att, lab, counts, n_lab = token_batch(B, 125, V, cfg.audio_token_id, cfg.pad_token_id, cfg.eos_token_id, L=L)
ids_d, att_d, lab_d, counts_d = (torch.from_numpy(x).to(dev) for x in (ids, att, lab, counts))
def step():
    feats, _ = fe.extract(wav, lens)
    rows, tg, _n = ops.label_rows(lab_d)
    tr.training_step(dict(input_ids=ids_d, input_features=feats, attention_mask=att_d, labels=lab_d, audio_token_counts=counts_d, label_meta=(rows, tg, n_lab)))
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.events() if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::full")]
for e in sorted(rows, key=lambda e: -e.device_time_total)[:12]:
    print(e.name, e.input_shapes, round(e.device_time_total, 1), [s for s in (e.stack or [])[:6]])
