"""Encoder architecture descriptors for the PLM backbones the hot path supports.

The reference builds its text encoder with ``AutoModel.from_pretrained(plm_model)``
(reference manner/models/components/news_encoder.py:20), so the architecture is
whatever HF config the name resolves to.  Offline there is no hub access; the
presets below restate the published architectures (BertConfig / RobertaConfig
values) and ``from_hf_config`` reads a local ``config.json``.
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass

ARCH_BERT = 0
ARCH_ROBERTA = 1


@dataclass(frozen=True)
class EncoderConfig:
    arch: int = ARCH_BERT          # position-id rule: BERT arange / RoBERTa cumsum+pad
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    vocab: int = 30522
    max_pos: int = 512
    type_vocab: int = 2
    ln_eps: float = 1e-12
    pad_id: int = 0
    # parameter naming of the HF checkpoint: "bert" (BertModel / RobertaModel) or "distilbert" (DistilBertModel: same
    # post-LayerNorm block, no token-type embedding, no pooler, q_lin/k_lin/... names) — see weights.canonical_weights
    naming: str = "bert"

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    def flops_per_news(self, length: int) -> int:
        """Algorithmic FLOPs of one encoded news of ``length`` real tokens.

        SURVEY.md §8(d): F(L) = layers*(24*L*H^2 + 4*L^2*H) for I = 4H, generalised
        here to 8*L*H^2 + 4*L*H*I + 4*L^2*H per layer.
        """
        h, i = self.hidden, self.intermediate
        per_layer = 8 * length * h * h + 4 * length * h * i + 4 * length * length * h
        return self.layers * per_layer

    def to_dict(self) -> dict:
        return asdict(self)


PRESETS = {
    "bert-base-uncased": EncoderConfig(),
    "roberta-base": EncoderConfig(arch=ARCH_ROBERTA, vocab=50265, max_pos=514, type_vocab=1,
                                  ln_eps=1e-5, pad_id=1),
    "roberta-large": EncoderConfig(arch=ARCH_ROBERTA, hidden=1024, layers=24, heads=16,
                                   intermediate=4096, vocab=50265, max_pos=514, type_vocab=1,
                                   ln_eps=1e-5, pad_id=1),
    # small architectures for fast unit tests (same code paths, head_dim 64)
    "tiny-bert": EncoderConfig(hidden=128, layers=2, heads=2, intermediate=512, vocab=2048,
                               max_pos=128),
    # two roberta-large-shaped layers (H=1024, 16 heads, I=4096): exercises the 4/12/16-column-tile GEMM
    # shapes and the 4-vector LayerNorm of configs[4] without 355 M parameters
    "mini-roberta-large": EncoderConfig(arch=ARCH_ROBERTA, hidden=1024, layers=2, heads=16, intermediate=4096,
                                        vocab=4096, max_pos=130, type_vocab=1, ln_eps=1e-5, pad_id=1),
    # the multilingual experiments of the reference use distilbert-base-multilingual-cased
    "distilbert-base-multilingual-cased": EncoderConfig(layers=6, vocab=119547, type_vocab=1, naming="distilbert"),
    "tiny-distilbert": EncoderConfig(hidden=128, layers=2, heads=2, intermediate=512, vocab=2048, max_pos=128,
                                     type_vocab=1, naming="distilbert"),
    "tiny-roberta": EncoderConfig(arch=ARCH_ROBERTA, hidden=128, layers=2, heads=2,
                                  intermediate=512, vocab=2048, max_pos=130, type_vocab=1,
                                  ln_eps=1e-5, pad_id=1),
}


def from_hf_config(path: str) -> EncoderConfig:
    """Read a local HF ``config.json`` (directory or file)."""
    if os.path.isdir(path):
        path = os.path.join(path, "config.json")
    with open(path) as f:
        c = json.load(f)
    mt = c.get("model_type", "bert")
    if mt == "distilbert":
        if c.get("activation", "gelu") != "gelu":
            raise ValueError("the HIP encoder implements exact erf GeLU only")
        return EncoderConfig(arch=ARCH_BERT, hidden=c["dim"], layers=c["n_layers"], heads=c["n_heads"],
                             intermediate=c["hidden_dim"], vocab=c["vocab_size"], max_pos=c["max_position_embeddings"],
                             type_vocab=1, ln_eps=1e-12, pad_id=c.get("pad_token_id", 0), naming="distilbert")
    if mt not in ("bert", "roberta", "xlm-roberta"):
        raise ValueError(f"unsupported PLM model_type {mt!r}: the HIP encoder implements BERT / RoBERTa / DistilBERT")
    if c.get("hidden_act", "gelu") != "gelu":
        raise ValueError("the HIP encoder implements exact erf GeLU only")
    if c.get("position_embedding_type", "absolute") != "absolute":
        raise ValueError("only absolute position embeddings are supported")
    arch = ARCH_BERT if mt == "bert" else ARCH_ROBERTA
    return EncoderConfig(
        arch=arch,
        hidden=c["hidden_size"],
        layers=c["num_hidden_layers"],
        heads=c["num_attention_heads"],
        intermediate=c["intermediate_size"],
        vocab=c["vocab_size"],
        max_pos=c["max_position_embeddings"],
        type_vocab=c["type_vocab_size"],
        ln_eps=c.get("layer_norm_eps", 1e-12),
        pad_id=c.get("pad_token_id", 0 if arch == ARCH_BERT else 1),
    )


def resolve(plm_model: str) -> EncoderConfig:
    """``plm_model`` is a preset name or a local HF directory."""
    if os.path.isdir(plm_model):
        return from_hf_config(plm_model)
    if plm_model in PRESETS:
        return PRESETS[plm_model]
    raise ValueError(
        f"PLM {plm_model!r} is neither a local HF directory nor a known architecture preset "
        f"({sorted(PRESETS)}); there is no hub access in this build")
