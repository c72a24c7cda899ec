import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tiny_audio_amd.asr_processing import LogMelFeatureExtractor
from tiny_audio_amd import _lib
from tiny_audio_amd.ops import ptr, stream
fe = LogMelFeatureExtractor(128, "cuda")
B = 32
wav = 0.1 * torch.randn(B, 160000, device="cuda"); lens = torch.full((B,), 160000, device="cuda", dtype=torch.int64)
for _ in range(3): fe.extract(wav, lens)
# raw call so that finalize does not rescale the stamps: call, then read before finalize? finalize rescales out -> undo: x = (max(v, floor)+4)/4
f, m = fe.extract(wav, lens)
torch.cuda.synchronize()
print("stamps (after the finalize map (x+4)/4):", [round(float(v) * 4 - 4) for v in f.reshape(-1)[:6]])
