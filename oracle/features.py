"""Log-mel front end (row a1 of SURVEY.md section 8) restated in numpy.

Follows ``WhisperFeatureExtractor`` as the reference instantiates it for
GLM-ASR (``tiny_audio/asr_modeling.py:190-201``: ``feature_size=128``,
``padding=False`` attribute, but the collator calls it with
``padding="longest"`` -- ``scripts/train.py:327-333``).
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
N_MELS = 128


def _hz_to_mel_slaney(freq):
    """TF:audio_utils.py:448-481 (mel_scale="slaney")."""
    freq = np.asarray(freq, dtype=np.float64)
    mels = 3.0 * freq / 200.0
    logstep = 27.0 / np.log(6.4)
    log_region = freq >= 1000.0
    safe = np.where(log_region, freq, 1000.0)
    return np.where(log_region, 15.0 + np.log(safe / 1000.0) * logstep, mels)


def _mel_to_hz_slaney(mels):
    """TF:audio_utils.py:484-520 (mel_scale="slaney")."""
    mels = np.asarray(mels, dtype=np.float64)
    freq = 200.0 * mels / 3.0
    logstep = np.log(6.4) / 27.0
    log_region = mels >= 15.0
    return np.where(log_region, 1000.0 * np.exp(logstep * (mels - 15.0)), freq)


def mel_filter_bank(n_freq: int = N_FFT // 2 + 1, n_mels: int = N_MELS,
                    fmin: float = 0.0, fmax: float = 8000.0, sr: int = SAMPLE_RATE) -> np.ndarray:
    """[n_freq, n_mels] slaney-normalised triangular bank.

    TF:audio_utils.py:638-730 with ``norm="slaney", mel_scale="slaney"`` as
    ``WhisperFeatureExtractor.__init__`` asks for it
    (TF:models/whisper/feature_extraction_whisper.py:95-103).
    """
    mel_freqs = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    filter_freqs = _mel_to_hz_slaney(mel_freqs)
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    # TF:audio_utils.py:541-560
    filter_diff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2:n_mels + 2] - filter_freqs[:n_mels])
    return fb * enorm[None, :]


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    """torch.hann_window(n) (periodic=True), used at
    TF:models/whisper/feature_extraction_whisper.py:141."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)


def pad_batch(wavs):
    """``SequenceFeatureExtractor.pad(padding="longest")``: zero-pad on the
    right to the longest clip, sample-level attention mask of ones/zeros."""
    lens = np.array([len(w) for w in wavs], dtype=np.int64)
    lmax = int(lens.max())
    out = np.zeros((len(wavs), lmax), dtype=np.float32)
    for i, w in enumerate(wavs):
        out[i, : len(w)] = np.asarray(w, dtype=np.float32)
    return out, lens


def log_mel(wav_padded: np.ndarray, lens: np.ndarray, n_mels: int = N_MELS):
    """[B, Ls] float32 (+ true lengths) -> features [B, n_mels, T] float32,
    frame mask [B, T] int32, with T = Ls // 160.

    TF:models/whisper/feature_extraction_whisper.py:135-168 (torch.stft with
    center=True / reflect pad, periodic Hann, |X|^2, drop last frame, mel,
    clamp 1e-10, log10, per-clip max-8 floor, (x+4)/4) and :330-339 (mask
    ``[:, ::hop]``, trimmed by one when Ls % hop != 0).
    Arithmetic is done in float64 and rounded once; the reference runs the same
    chain in float32, so the pin tolerance is ~1e-5 (its own stated tolerance).
    """
    wav = np.asarray(wav_padded, dtype=np.float64)
    B, Ls = wav.shape
    half = N_FFT // 2
    padded = np.pad(wav, ((0, 0), (half, half)), mode="reflect")
    n_frames = 1 + Ls // HOP
    idx = np.arange(n_frames)[:, None] * HOP + np.arange(N_FFT)[None, :]
    frames = padded[:, idx] * hann_periodic()[None, None, :]
    spec = np.fft.rfft(frames, n=N_FFT, axis=-1)
    power = (spec.real ** 2 + spec.imag ** 2)[:, :-1, :]          # drop last frame
    fb = mel_filter_bank(n_mels=n_mels).astype(np.float32).astype(np.float64)  # TF keeps it f32
    mel = np.einsum("fm,btf->bmt", fb, power)
    logspec = np.log10(np.maximum(mel, 1e-10))
    mx = logspec.reshape(B, -1).max(axis=1)[:, None, None]
    logspec = np.maximum(logspec, mx - 8.0)
    feats = ((logspec + 4.0) / 4.0).astype(np.float32)
    T = feats.shape[2]
    sample_mask = (np.arange(Ls)[None, :] < np.asarray(lens)[:, None]).astype(np.int32)
    mask = sample_mask[:, ::HOP]
    if Ls % HOP != 0:
        mask = mask[:, :-1]
    assert mask.shape[1] == T, (mask.shape, T)
    return feats, mask


def conv_out_length(length, conv_layers=((1, 3, 1), (1, 3, 2))):
    """tiny_audio/asr_config.py:9-19."""
    for p, k, s in conv_layers:
        length = (length + 2 * p - (k - 1) - 1) // s + 1
    return length


def mlp_out_length(length, k: int = 4):
    """tiny_audio/projectors.py:52-55 (also MoE :253-255)."""
    return (length - k) // k + 1


def audio_token_counts(frame_mask: np.ndarray, k: int = 4) -> np.ndarray:
    """scripts/train.py:335-338: mask.sum -> conv formula -> projector length."""
    mel_lengths = frame_mask.sum(axis=-1).astype(np.int64)
    return mlp_out_length(conv_out_length(mel_lengths), k).astype(np.int64)
