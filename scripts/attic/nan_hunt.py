import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import weights as OW
from tests.golden import recipe as R
from tiny_audio_amd.asr_config import ASRConfig
from tiny_audio_amd.asr_modeling import ASRModel
DEV = "cuda"
def poison():
    blocks = [torch.full((64 * 1024 * 1024,), float("nan"), device="cuda") for _ in range(4)]
    small = [torch.full((n,), float("nan"), device="cuda") for n in (256, 4096, 65536, 1 << 20) for _ in range(8)]
    del blocks, small
g = np.load("tests/golden/asr_small.npz")
S = R.SMALL
E, D, H = S["enc"]["hidden"], S["lm"]["hidden"], S["proj_hidden"]
ids, att, lab, counts = R.asr_tokens(g["counts"])
for mode in ("moe", "fullft"):
    poison()
    kw = dict(projector_type="moe", router_jitter_noise=0.0) if mode == "moe" else dict(freeze_language_model=False)
    cfg = ASRConfig(audio_config=S["enc"], text_config=S["lm"], projector_hidden_dim=H, audio_token_id=S["audio_token_id"], **kw)
    m = ASRModel(cfg, device=DEV, init="none")
    m.audio_tower.load_state_dict_hf(OW.init_encoder(S["enc"], 0))
    m.language_model.load_state_dict_hf(OW.init_lm(S["lm"], 1))
    pw = OW.init_moe_projector(E, D, H) if mode == "moe" else OW.init_mlp_projector(E, D, H)
    m.load_state_dict({"projector." + k: torch.from_numpy(v) for k, v in pw.items()})
    m.train()
    for trial in range(3):
        poison()
        feats = torch.from_numpy(g["input_features"]).to(DEV)
        hidden = m.audio_tower(feats).last_hidden_state
        y = m.projector(hidden)
        print(mode, trial, "enc nan", bool(torch.isnan(hidden.float()).any()), "proj nan", bool(torch.isnan(y).any()), "counts", counts.tolist(), "y shape", tuple(y.shape))
        out = m(input_ids=torch.from_numpy(ids), input_features=torch.from_numpy(g["input_features"]), attention_mask=torch.from_numpy(att),
                labels=torch.from_numpy(lab), audio_token_counts=torch.from_numpy(counts))
        print("   loss", float(out.loss), "logits nan rows", int(torch.isnan(out.logits.float()).any(-1).sum()), "of", out.logits.shape[0] * out.logits.shape[1])
        if mode == "fullft":
            b = m.language_model._bufs
            for k in ("embed_f32", "embed_bf16", "embed_t_bf16"):
                print("   ", k, tuple(b[k].shape), "nan:", bool(torch.isnan(b[k].float()).any()))
            for k, v in b.items():
                if torch.is_tensor(v) and v.is_floating_point() and torch.isnan(v.float()).any():
                    print("    NaN in buffer", k, tuple(v.shape))
