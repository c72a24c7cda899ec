"""Configuration of the MI355X-native ASR training path.

Mirrors ``tiny_audio/asr_config.py`` (field names, defaults, the conv length formula at :9-19) without
depending on ``transformers``: sub-configs are plain objects carrying the shape fields the kernels need.
An HF ``GlmAsrEncoderConfig`` / ``Qwen3Config`` (or a dict) can be passed wherever a sub-config is expected.
"""
from __future__ import annotations

from typing import Optional

# [(padding, kernel, stride), ...]  tiny_audio/asr_config.py:6
DEFAULT_ENCODER_CONV_LAYERS = [(1, 3, 1), (1, 3, 2)]


def compute_encoder_output_length(mel_length, conv_layers=None):
    """tiny_audio/asr_config.py:9-19: (L + 2p - (k-1) - 1) // s + 1 per conv; ints or tensors."""
    layers = conv_layers if conv_layers is not None else DEFAULT_ENCODER_CONV_LAYERS
    length = mel_length
    for padding, kernel_size, stride in layers:
        length = (length + 2 * padding - (kernel_size - 1) - 1) // stride + 1
    return length


def _get(src, names, default=None):
    for n in names:
        if isinstance(src, dict):
            if n in src and src[n] is not None:
                return src[n]
        elif src is not None and getattr(src, n, None) is not None:
            return getattr(src, n)
    return default


class EncoderConfig:
    """GlmAsrEncoderConfig fields (TF:models/glmasr/configuration_glmasr.py:44-61)."""

    def __init__(self, src=None, **kw):
        src = {**(src if isinstance(src, dict) else {}), **kw} if (isinstance(src, dict) or src is None) else src
        self.hidden_size = int(_get(src, ["hidden_size", "hidden"], 1280))
        self.intermediate_size = int(_get(src, ["intermediate_size", "ffn"], 5120))
        self.num_hidden_layers = int(_get(src, ["num_hidden_layers", "layers"], 32))
        self.num_attention_heads = int(_get(src, ["num_attention_heads", "heads"], 20))
        self.num_mel_bins = int(_get(src, ["num_mel_bins", "n_mels"], 128))
        self.max_position_embeddings = int(_get(src, ["max_position_embeddings"], 1500))
        rp = _get(src, ["rope_parameters"], None) or {}
        self.rope_theta = float(_get(src, ["rope_theta"], None) or rp.get("rope_theta", 10000.0))
        self.partial_rotary_factor = float(_get(src, ["partial_rotary_factor", "partial_rotary"], None)
                                           or rp.get("partial_rotary_factor", 0.5))
        self.layer_norm_eps = float(_get(src, ["layer_norm_eps", "ln_eps"], 1e-5))
        if self.hidden_size // self.num_attention_heads != 64 or self.partial_rotary_factor != 0.5:
            raise ValueError("ta355 encoder kernels are built for head_dim 64 with partial rotary 0.5 (GLM-ASR)")


class LMConfig:
    """Qwen3Config fields; defaults = Qwen3-0.6B (SURVEY.md section 8 preamble)."""

    def __init__(self, src=None, **kw):
        src = {**(src if isinstance(src, dict) else {}), **kw} if (isinstance(src, dict) or src is None) else src
        self.vocab_size = int(_get(src, ["vocab_size", "vocab"], 151670))
        self.hidden_size = int(_get(src, ["hidden_size", "hidden"], 1024))
        self.intermediate_size = int(_get(src, ["intermediate_size", "ffn"], 3072))
        self.num_hidden_layers = int(_get(src, ["num_hidden_layers", "layers"], 28))
        self.num_attention_heads = int(_get(src, ["num_attention_heads", "heads"], 16))
        self.num_key_value_heads = int(_get(src, ["num_key_value_heads", "kv_heads"], 8))
        self.head_dim = int(_get(src, ["head_dim"], 128))
        self.rms_norm_eps = float(_get(src, ["rms_norm_eps", "rms_eps"], 1e-6))
        rp = _get(src, ["rope_parameters"], None) or {}
        self.rope_theta = float(_get(src, ["rope_theta"], None) or rp.get("rope_theta", 1e6))
        self.max_position_embeddings = int(_get(src, ["max_position_embeddings"], 4096))
        if self.head_dim != 128:
            raise ValueError("ta355 LM kernels are built for head_dim 128 (Qwen3)")


class ASRConfig:
    """Same knobs as the reference ASRConfig (tiny_audio/asr_config.py:36-220); unknown kwargs are kept as
    attributes (the reference lets e.g. ``router_z_loss_coef`` / ``router_jitter_noise`` ride along that way)."""

    model_type = "asr_model"

    def __init__(self, audio_model_id: str = "zai-org/GLM-ASR-Nano-2512", text_model_id: str = "Qwen/Qwen3-0.6B",
                 attn_implementation: str = "ta355", model_dtype: str = "bfloat16",
                 system_prompt: str = "You are a helpful assistant.", encoder_dim: Optional[int] = None,
                 llm_dim: Optional[int] = None, encoder_conv_layers: Optional[list] = None,
                 audio_sample_rate: int = 16000, projector_pool_stride: int = 4, downsample_rate: int = 5,
                 projector_hidden_dim: Optional[int] = None, projector_type: str = "mlp",
                 audio_token_dropout: float = 0.0, num_experts: int = 4, num_experts_per_tok: int = 2,
                 router_aux_loss_coef: float = 0.01, use_lora: bool = False, lora_rank: int = 8, lora_alpha: int = 32,
                 lora_dropout: float = 0.0, lora_target_modules: Optional[list] = None, freeze_projector: bool = False,
                 freeze_language_model: bool = True, max_new_tokens: int = 128, audio_config=None, text_config=None,
                 audio_token_id: int = 151669, pad_token_id: int = 151643, eos_token_id: int = 151645, **kwargs):
        self.audio_model_id = audio_model_id
        self.text_model_id = text_model_id
        self.attn_implementation = attn_implementation
        self.model_dtype = model_dtype
        self.system_prompt = system_prompt
        self.encoder_conv_layers = encoder_conv_layers or DEFAULT_ENCODER_CONV_LAYERS
        self.audio_sample_rate = audio_sample_rate
        self.projector_pool_stride = projector_pool_stride
        self.downsample_rate = downsample_rate
        self.projector_hidden_dim = projector_hidden_dim
        self.projector_type = projector_type
        self.audio_token_dropout = audio_token_dropout
        self.num_experts = num_experts
        self.num_experts_per_tok = num_experts_per_tok
        self.router_aux_loss_coef = router_aux_loss_coef
        self.use_lora = use_lora
        self.lora_rank = lora_rank
        self.lora_alpha = lora_alpha
        self.lora_dropout = lora_dropout
        self.lora_target_modules = lora_target_modules or ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj",
                                                            "up_proj", "down_proj"]
        self.freeze_projector = freeze_projector
        self.freeze_language_model = freeze_language_model
        self.max_new_tokens = max_new_tokens
        self.audio_config = audio_config if isinstance(audio_config, EncoderConfig) else EncoderConfig(audio_config)
        self.text_config = text_config if isinstance(text_config, LMConfig) else LMConfig(text_config)
        self.encoder_dim = encoder_dim or self.audio_config.hidden_size      # asr_modeling.py:259-265
        self.llm_dim = llm_dim or self.text_config.hidden_size               # asr_modeling.py:267-273
        self.audio_token_id = audio_token_id
        self.pad_token_id = pad_token_id
        self.eos_token_id = eos_token_id
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_dict(self):
        d = {k: v for k, v in self.__dict__.items() if k not in ("audio_config", "text_config")}
        d["audio_config"] = dict(self.audio_config.__dict__)
        d["text_config"] = dict(self.text_config.__dict__)
        return d
