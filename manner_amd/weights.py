"""Deterministic synthetic PLM weights, keyed by HF parameter name.

There are no pretrained checkpoints offline, so parity and bench
runs use seeded weights that can be regenerated bit-identically on the GPU box
instead of shipping ~440 MB.  Every tensor draws from its own
``numpy.random.Generator(PCG64([seed, crc32(name)]))`` stream so the values do
not depend on generation order.

Key names follow HF ``BertModel`` / ``RobertaModel`` state_dict, which is what sits
under ``news_encoder.text_encoder.plm_model.`` in reference checkpoints
(reference manner/models/components/news_encoder.py:20).
"""
from __future__ import annotations

import hashlib
import zlib
from typing import Dict, Iterator, Tuple

import numpy as np

from .config import EncoderConfig


_DISTIL_TO_BERT = (("attention.q_lin.", "attention.self.query."), ("attention.k_lin.", "attention.self.key."),
                   ("attention.v_lin.", "attention.self.value."), ("attention.out_lin.", "attention.output.dense."),
                   ("sa_layer_norm.", "attention.output.LayerNorm."), ("ffn.lin1.", "intermediate.dense."),
                   ("ffn.lin2.", "output.dense."), ("output_layer_norm.", "output.LayerNorm."))


def canonical_weights(cfg: EncoderConfig, weights):
    """The encoder and the oracle speak BertModel parameter names.  A DistilBertModel state dict (same post-LayerNorm
    block: transformers/models/distilbert/modeling_distilbert.py) is renamed and given an all-zero token-type row."""
    if cfg.naming != "distilbert":
        return weights
    out = {}
    for k, v in weights.items():
        if k.startswith("transformer.layer."):
            k = "encoder.layer." + k[len("transformer.layer."):]
            for a, b in _DISTIL_TO_BERT:
                if a in k:
                    k = k.replace(a, b)
                    break
        out[k] = v
    ref = out["embeddings.LayerNorm.bias"]
    out["embeddings.token_type_embeddings.weight"] = ref.new_zeros((1, cfg.hidden)) if hasattr(ref, "new_zeros") \
        else np.zeros((1, cfg.hidden), np.float32)
    return out


def plm_param_shapes(cfg: EncoderConfig, with_pooler: bool = True) -> Iterator[Tuple[str, Tuple[int, ...]]]:
    h, i = cfg.hidden, cfg.intermediate
    if cfg.naming == "distilbert":
        yield "embeddings.word_embeddings.weight", (cfg.vocab, h)
        yield "embeddings.position_embeddings.weight", (cfg.max_pos, h)
        yield "embeddings.LayerNorm.weight", (h,)
        yield "embeddings.LayerNorm.bias", (h,)
        for l in range(cfg.layers):
            p = f"transformer.layer.{l}."
            for n in ("q_lin", "k_lin", "v_lin", "out_lin"):
                yield p + f"attention.{n}.weight", (h, h)
                yield p + f"attention.{n}.bias", (h,)
            yield p + "sa_layer_norm.weight", (h,)
            yield p + "sa_layer_norm.bias", (h,)
            yield p + "ffn.lin1.weight", (i, h)
            yield p + "ffn.lin1.bias", (i,)
            yield p + "ffn.lin2.weight", (h, i)
            yield p + "ffn.lin2.bias", (h,)
            yield p + "output_layer_norm.weight", (h,)
            yield p + "output_layer_norm.bias", (h,)
        return
    yield "embeddings.word_embeddings.weight", (cfg.vocab, h)
    yield "embeddings.position_embeddings.weight", (cfg.max_pos, h)
    yield "embeddings.token_type_embeddings.weight", (cfg.type_vocab, h)
    yield "embeddings.LayerNorm.weight", (h,)
    yield "embeddings.LayerNorm.bias", (h,)
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        for n in ("query", "key", "value"):
            yield p + f"attention.self.{n}.weight", (h, h)
            yield p + f"attention.self.{n}.bias", (h,)
        yield p + "attention.output.dense.weight", (h, h)
        yield p + "attention.output.dense.bias", (h,)
        yield p + "attention.output.LayerNorm.weight", (h,)
        yield p + "attention.output.LayerNorm.bias", (h,)
        yield p + "intermediate.dense.weight", (i, h)
        yield p + "intermediate.dense.bias", (i,)
        yield p + "output.dense.weight", (h, i)
        yield p + "output.dense.bias", (h,)
        yield p + "output.LayerNorm.weight", (h,)
        yield p + "output.LayerNorm.bias", (h,)
    if with_pooler:
        # computed by HF but dropped by the reference (news_encoder.py:34); kept for key parity
        yield "pooler.dense.weight", (h, h)
        yield "pooler.dense.bias", (h,)


def _stream(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def make_plm_weights(cfg: EncoderConfig, seed: int = 42, std: float = 0.02,
                     with_pooler: bool = True) -> Dict[str, np.ndarray]:
    """fp32 numpy tensors for every PLM parameter.

    Matrices ~ N(0, std^2) (HF ``initializer_range`` is 0.02; a larger ``std`` gives
    "trained-like" activations with more spread between news).  Unlike HF's init,
    biases and LayerNorm affine terms are non-trivial (bias ~ N(0, 0.02^2),
    gamma ~ 1 + N(0, 0.05^2)) so that a kernel which drops one of them fails parity.
    """
    out: Dict[str, np.ndarray] = {}
    for name, shape in plm_param_shapes(cfg, with_pooler):
        g = _stream(seed, name)
        if name.endswith(("LayerNorm.weight", "layer_norm.weight")):
            w = 1.0 + 0.05 * g.standard_normal(shape, dtype=np.float32)
        elif name.endswith(".bias"):
            w = 0.02 * g.standard_normal(shape, dtype=np.float32)
        else:
            w = std * g.standard_normal(shape, dtype=np.float32)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def make_additive_attention_weights(input_dim: int, query_dim: int, seed: int = 42,
                                    prefix: str = "additive_attention.") -> Dict[str, np.ndarray]:
    """Parameters of reference AdditiveAttention (attention.py:9-10): linear [Q,D], [Q]; query [Q]."""
    g = _stream(seed, prefix + "linear.weight")
    bound = 1.0 / np.sqrt(input_dim)
    w = g.uniform(-bound, bound, (query_dim, input_dim)).astype(np.float32)
    b = _stream(seed, prefix + "linear.bias").uniform(-bound, bound, (query_dim,)).astype(np.float32)
    q = _stream(seed, prefix + "query").uniform(-0.1, 0.1, (query_dim,)).astype(np.float32)
    return {prefix + "linear.weight": w, prefix + "linear.bias": b, prefix + "query": q}


def tensor_sha256(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_entity_weights(n_entities: int, dim: int = 100, query_dim: int = 200, hidden: int = 768, seed: int = 42,
                        with_linear: bool = True) -> Dict[str, np.ndarray]:
    """Seeded parameters of the reference's entity branch (news_encoder.py:98-113), reference key names:
    ``entity_encoder.*`` and ``linear.*`` as they sit under ``news_encoder.`` in a checkpoint."""
    out: Dict[str, np.ndarray] = {}

    def put(name, shape, scale):
        out[name] = (scale * _stream(seed, name).standard_normal(shape, dtype=np.float32)).astype(np.float32)

    put("entity_encoder.pretrained_embedding.weight", (n_entities, dim), 0.5)
    put("entity_encoder.multihead_attention.in_proj_weight", (3 * dim, dim), 0.15)
    put("entity_encoder.multihead_attention.in_proj_bias", (3 * dim,), 0.05)
    put("entity_encoder.multihead_attention.out_proj.weight", (dim, dim), 0.15)
    put("entity_encoder.multihead_attention.out_proj.bias", (dim,), 0.05)
    for k, v in make_additive_attention_weights(dim, query_dim, seed=seed, prefix="entity_encoder.additive_attention.").items():
        out[k] = v
    if with_linear:
        put("linear.weight", (hidden, hidden + dim), 0.03)
        put("linear.bias", (hidden,), 0.02)
    return out


def make_mha_pool_weights(dim: int, query_dim: int, seed: int = 42, prefix: str = "") -> Dict[str, np.ndarray]:
    """Seeded parameters of an nn.MultiheadAttention(dim, heads) + AdditiveAttention(dim, query_dim) pair, reference key
    names (``multihead_attention.*`` / ``additive_attention.*``: PLMTextEncoder news_encoder.py:146-151, NRMSUserEncoder
    user_encoder.py:30-31)."""
    out: Dict[str, np.ndarray] = {}

    def put(name, shape, scale):
        out[prefix + name] = (scale * _stream(seed, prefix + name).standard_normal(shape, dtype=np.float32)).astype(np.float32)

    s = 1.0 / np.sqrt(dim)
    put("multihead_attention.in_proj_weight", (3 * dim, dim), s)
    put("multihead_attention.in_proj_bias", (3 * dim,), 0.05)
    put("multihead_attention.out_proj.weight", (dim, dim), s)
    put("multihead_attention.out_proj.bias", (dim,), 0.05)
    for k, v in make_additive_attention_weights(dim, query_dim, seed=seed, prefix=prefix + "additive_attention.").items():
        out[k] = v
    return out
