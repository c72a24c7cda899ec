"""Feature extraction + batch contract on MI355X.

``LogMelFeatureExtractor`` replaces ``WhisperFeatureExtractor`` as the reference configures it for GLM-ASR
(tiny_audio/asr_modeling.py:190-201; called at scripts/train.py:327-333 and tiny_audio/asr_processing.py:74-80):
raw 16 kHz waveforms are uploaded once and the log-mel + frame mask are computed on the GPU
(``ta_logmel_f32``) instead of in CPU dataloader workers -- the stage the reference itself flags as its
bottleneck (configs/experiments/embedded.yaml:37-41).

``ASRProcessor`` is the drop-in for tiny_audio/asr_processing.py:17-128: audio (+ optional transcript / system prompt) ->
``input_features, audio_attention_mask, input_ids, attention_mask`` through the token-count contract
(mel mask -> conv formula -> projector.get_output_length -> number of ``<audio>`` placeholders) and the tokenizer's chat template.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .asr_config import DEFAULT_ENCODER_CONV_LAYERS, compute_encoder_output_length
from .ops import F32, ptr, stream

N_FFT, HOP, N_BIN = 400, 160, 201


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)


def slaney_mel_filters(n_mels=128, n_freq=N_BIN, sr=16000, fmin=0.0, fmax=8000.0):
    """[n_freq, n_mels] float32, slaney scale + slaney area norm (TF:audio_utils.py:638-730 as Whisper asks)."""
    mel_pts = np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2)
    hz = _mel_to_hz(mel_pts)
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    diff = np.diff(hz)
    slopes = hz[None, :] - fft_freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]))
    fb *= (2.0 / (hz[2:n_mels + 2] - hz[:n_mels]))[None, :]
    return fb.astype(np.float32)


def dft_tables():
    """cos|sin twiddles [400, 402] rounded once from float64, and the periodic Hann window [400]."""
    n = np.arange(N_FFT, dtype=np.float64)[:, None]
    k = np.arange(N_BIN, dtype=np.float64)[None, :]
    ang = 2.0 * np.pi * ((n * k) % N_FFT) / N_FFT
    dft = np.concatenate([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)
    win = torch.hann_window(N_FFT, periodic=True, dtype=torch.float32).numpy()   # exactly the reference's window
    return dft, win


class LogMelFeatureExtractor:
    sampling_rate = 16000

    def __init__(self, feature_size=128, device="cuda"):
        self.feature_size = feature_size
        self.device = torch.device(device)
        self.hop_length, self.n_fft = HOP, N_FFT
        self.padding = False                                     # tiny_audio/asr_modeling.py:199-200
        dft, win = dft_tables()
        self._dft = torch.from_numpy(dft).to(self.device)
        self._win = torch.from_numpy(win).to(self.device)
        self._mel = torch.from_numpy(slaney_mel_filters(feature_size)).to(self.device)
        self._mel_ranges = None                                  # {first, end} bin per filter: computed once (ta_logmel_mel_ranges)

    def extract(self, wav: torch.Tensor, lens: torch.Tensor):
        """wav [B, Ls] f32 (zero padded, device), lens [B] int64 -> (features [B, n_mels, T], mask [B, T] int32);
        torch.ops.ta355.logmel."""
        from . import torch_ops
        return torch.ops.ta355.logmel(wav, lens, torch_ops.register_module(self))

    def _extract(self, wav: torch.Tensor, lens: torch.Tensor):
        B, Ls = wav.shape
        T = Ls // HOP
        feats = torch.empty((B, self.feature_size, T), device=self.device, dtype=F32)
        mask = torch.empty((B, T), device=self.device, dtype=torch.int32)
        if self._mel_ranges is None:
            self._mel_ranges = torch.empty(2 * self.feature_size, device=self.device, dtype=torch.int32)
            _lib.check(_lib.lib().ta_logmel_mel_ranges(ptr(self._mel), self.feature_size, ptr(self._mel_ranges), stream()),
                       "ta_logmel_mel_ranges")
        cm = torch.empty(_lib.lib().ta_logmel_scratch_floats(B, Ls, self.feature_size), device=self.device, dtype=F32)   # per-workgroup clip maxima
        _lib.check(_lib.lib().ta_logmel_f32(ptr(wav), ptr(lens), B, Ls, ptr(self._dft), ptr(self._win), ptr(self._mel),
                                            self.feature_size, ptr(feats), ptr(mask), ptr(cm), ptr(self._mel_ranges), stream()),
                   "ta_logmel_f32")
        return feats, mask

    def __call__(self, raw_speech, sampling_rate=None, padding="longest", return_attention_mask=True,
                 return_tensors="pt", **_):
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(f"LogMelFeatureExtractor was built for {self.sampling_rate} Hz audio, got {sampling_rate}")
        if isinstance(raw_speech, np.ndarray) and raw_speech.ndim == 1:
            raw_speech = [raw_speech]
        lens = np.array([len(w) for w in raw_speech], dtype=np.int64)
        Ls = int(lens.max())
        host = np.zeros((len(raw_speech), Ls), dtype=np.float32)
        for i, w in enumerate(raw_speech):
            host[i, : len(w)] = np.asarray(w, dtype=np.float32)
        wav = torch.from_numpy(host).to(self.device, non_blocking=True)
        feats, mask = self.extract(wav, torch.from_numpy(lens).to(self.device))
        out = {"input_features": feats}
        if return_attention_mask:
            out["attention_mask"] = mask
        return out


class ASRProcessor:
    """Drop-in for ``tiny_audio.asr_processing.ASRProcessor`` (tiny_audio/asr_processing.py:17-128; built by
    ``ASRModel.get_processor()``, tiny_audio/asr_modeling.py:384-396): same constructor arguments, same ``__call__`` arguments,
    same returned keys -- ``input_features`` / ``audio_attention_mask`` (only with audio), ``input_ids``, ``attention_mask``.

    The feature extractor is whatever the caller hands in; ``LogMelFeatureExtractor`` computes the log-mel on the GPU.  The
    ``<audio>`` placeholder count follows the LONGEST clip's real mel length (attention-mask sum -> conv formulas ->
    ``projector.get_output_length``, :84-87), the chat prompt is rendered by the tokenizer's own template with Qwen3's thinking
    mode off, and a generation prompt is appended exactly when no target text is given (:104-112)."""

    attributes = ["feature_extractor", "tokenizer"]
    feature_extractor_class = "AutoFeatureExtractor"
    tokenizer_class = "AutoTokenizer"
    AUDIO_TOKEN = "<audio>"
    TRANSCRIBE_PROMPT = "Transcribe the speech to text"

    def __init__(self, feature_extractor, tokenizer, projector=None, encoder_conv_layers=None):
        self.feature_extractor = feature_extractor
        self.tokenizer = tokenizer
        self.audio_token_id = tokenizer.convert_tokens_to_ids(self.AUDIO_TOKEN)
        self.projector = projector
        self.encoder_conv_layers = encoder_conv_layers or DEFAULT_ENCODER_CONV_LAYERS

    # ---- length bookkeeping
    def _compute_encoder_output_length(self, mel_length):
        return compute_encoder_output_length(mel_length, self.encoder_conv_layers)

    def audio_token_counts(self, frame_mask: torch.Tensor) -> torch.Tensor:
        """Per-clip ``<audio>`` counts from the mel frame mask (what scripts/train.py:335-342 computes in the collator)."""
        enc_lengths = self._compute_encoder_output_length(frame_mask.sum(dim=-1))
        return self.projector.get_output_length(enc_lengths).to(torch.long)

    # ---- prompt
    @classmethod
    def build_messages(cls, num_audio_tokens: int, text=None, system_prompt=None):
        """The chat turns of one request (:89-102): optional system turn, the user turn carrying the placeholders (+ the
        instruction), and the transcript as the assistant turn when training text is given."""
        user = cls.AUDIO_TOKEN * int(num_audio_tokens) if num_audio_tokens > 0 else ""
        if cls.TRANSCRIBE_PROMPT:
            user = user + " " + cls.TRANSCRIBE_PROMPT if user else cls.TRANSCRIBE_PROMPT
        turns = [{"role": "system", "content": system_prompt}] if system_prompt else []
        turns.append({"role": "user", "content": user})
        if text is not None:
            turns.append({"role": "assistant", "content": text})
        return turns

    @staticmethod
    def tokenize_messages(tokenizer, messages, add_generation_prompt: bool, return_tensors="pt") -> torch.Tensor:
        """-> ``input_ids`` [1, L] through the tokenizer's chat template (thinking mode off, :104-123); tokenizers differ in what
        ``apply_chat_template`` returns (a tensor, or a BatchEncoding / dict holding one)."""
        out = tokenizer.apply_chat_template(messages, tokenize=True, add_generation_prompt=add_generation_prompt,
                                            return_tensors=return_tensors, enable_thinking=False)
        if not isinstance(out, torch.Tensor):
            out = out["input_ids"] if isinstance(out, dict) else out.get("input_ids", getattr(out, "input_ids", None))
        ids = torch.as_tensor(out)
        return ids[None, :] if ids.dim() == 1 else ids

    def __call__(self, audio=None, text=None, system_prompt=None, return_tensors: str = "pt", **kwargs) -> dict:
        result = {}
        n_tok = 0
        if audio is not None:
            f = self.feature_extractor(audio, sampling_rate=getattr(self.feature_extractor, "sampling_rate", 16000),
                                       return_attention_mask=True, return_tensors=return_tensors, **kwargs)
            result["input_features"] = f["input_features"]
            result["audio_attention_mask"] = f["attention_mask"]
            real_mel_len = int(f["attention_mask"].sum(dim=-1).max().item())               # :85 (one host sync, as the reference)
            n_tok = self.projector.get_output_length(self._compute_encoder_output_length(real_mel_len))
        messages = self.build_messages(n_tok, text, system_prompt)
        ids = self.tokenize_messages(self.tokenizer, messages, add_generation_prompt=text is None, return_tensors=return_tensors)
        result["input_ids"] = ids
        result["attention_mask"] = torch.ones_like(ids)
        return result
