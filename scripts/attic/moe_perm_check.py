import sys, os
sys.path.insert(0, "/root/repo")
import torch
from tiny_audio_amd.asr_config import ASRConfig
from tiny_audio_amd.projectors import PROJECTOR_CLASSES
torch.manual_seed(0)
cfg = ASRConfig(projector_type="moe", audio_token_dropout=0.0, router_jitter_noise=0.0)
proj = PROJECTOR_CLASSES["moe"](cfg).to("cuda") if hasattr(PROJECTOR_CLASSES["moe"](cfg), "to") else None
proj.train()
h = torch.randn(3, 500, 1280, device="cuda").to(torch.bfloat16) * 0.5
y1 = proj(h)
perm = [2, 0, 1]
y2 = proj(h[perm].contiguous())
d = (y1[perm].float() - y2.float()).abs()
print("max diff", float(d.max()), "rows differing", int((d.amax(-1) > 1e-3).sum()), "of", d.shape[0] * d.shape[1])
proj.eval()
y1 = proj(h); y2 = proj(h[perm].contiguous())
d = (y1[perm].float() - y2.float()).abs()
print("eval: max diff", float(d.max()), "rows differing", int((d.amax(-1) > 1e-3).sum()))
