"""Training path of MannerTextEncoder as a ``torch.autograd.Function`` over ``manner_hip_train_*``.

The reference trains through ``MannerTextEncoder.forward`` in ``train()`` mode (news_encoder.py:29-37) and
``loss.backward()`` (Lightning drives ``CRModule.training_step``, cr_module.py:140-171).  ``encode_train`` is that
forward on the HIP engine — HF dropouts included — and registers a backward that fills ``.grad`` of exactly the
parameters with ``requires_grad=True``; optimiser, scheduler and Lightning stay the reference's.

Frozen prefix: the reference freezes the *parameters* of ``frozen_layers`` but leaves the embeddings trainable, so its
backward runs through all layers (news_encoder.py:24-27).  That is what happens here when an embedding tensor requires
grad.  When every tensor below the first trainable layer ``f`` is frozen, the prefix is run once by the inference
engine (``encode_hidden``, eval arithmetic: no dropout in the frozen layers — the one documented deviation from the
reference, which keeps dropout active there) and training starts from ``hidden_states[f]``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from manner_amd import _lib, hip
from manner_amd.config import EncoderConfig
from manner_amd.weights import canonical_weights

Tensor = torch.Tensor
_TRAIN_PRECISIONS = ("fp32", "f16", "bf16")


def _cfg_c(cfg: EncoderConfig) -> _lib.EncoderConfigC:
    return _lib.EncoderConfigC(cfg.arch, cfg.hidden, cfg.layers, cfg.heads, cfg.intermediate, cfg.vocab, cfg.max_pos,
                               cfg.type_vocab, cfg.pad_id, cfg.ln_eps)


def _table(tensors: Sequence[Optional[Tensor]]):
    return (C.c_void_p * len(tensors))(*[(t.data_ptr() if t is not None else None) for t in tensors])


def dropout_mask(seed: int, site: int, p: float, n: int, device) -> Tensor:
    """The keep-bits (uint8 [n]) the training kernels use at one dropout site — for tests that feed an oracle the same mask."""
    out = torch.empty(n, dtype=torch.uint8, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().manner_hip_dropout_mask(C.c_uint64(seed), C.c_uint32(site), C.c_float(p), n, hip._ptr(out), hip._stream()))
    return out


class _EncodeTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids: Tensor, mask: Tensor, prefix: Optional[Tensor], opts: dict, *params: Tensor):
        cfg: EncoderConfig = opts["cfg"]
        lib = _lib.load()
        n, lp = ids.shape
        m_bound = (n * lp + 255) // 256 * 256 if opts.get("m_bound") is None else int(opts["m_bound"])
        start = int(opts["start_layer"])
        prec = _lib.PRECISIONS[opts["precision"]]
        cc = _cfg_c(cfg)
        dev = ids.device
        with torch.cuda.device(dev):
            saved = torch.empty(int(lib.manner_hip_train_saved_bytes(C.byref(cc), n, m_bound, start)), dtype=torch.uint8, device=dev)
            ws = torch.empty(int(lib.manner_hip_train_workspace_bytes(C.byref(cc), m_bound)), dtype=torch.uint8, device=dev)
            out = torch.empty((n, cfg.hidden), dtype=torch.float32, device=dev)
            weights = [p.detach() for p in params]
            status = hip.device_status(dev)
            _lib.check(lib.manner_hip_train_forward(
                C.byref(cc), _table(weights), len(weights), hip._ptr(ids), hip._ptr(mask), n, lp, m_bound, prec, start,
                hip._ptr(prefix), C.c_float(opts["p_hidden"]), C.c_float(opts["p_attn"]), C.c_float(opts["p_out"]),
                C.c_uint64(opts["seed"]), hip._ptr(out), hip._ptr(saved), saved.numel(), hip._ptr(ws), ws.numel(),
                hip._ptr(status.word), hip._stream()))
        ctx.opts, ctx.m_bound, ctx.prec, ctx.start = opts, m_bound, prec, start
        ctx.saved_buf, ctx.ws = saved, ws
        ctx.save_for_backward(ids, *params)
        ctx.prefix_grad = prefix is not None and prefix.requires_grad
        ctx.prefix_shape = None if prefix is None else tuple(prefix.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        ids, *params = ctx.saved_tensors
        opts, cfg = ctx.opts, ctx.opts["cfg"]
        lib = _lib.load()
        n, lp = ids.shape
        dev = ids.device
        need = ctx.needs_input_grad[4:]
        grads: List[Optional[Tensor]] = [torch.empty_like(p) if r else None for p, r in zip(params, need)]
        gprefix = torch.empty(ctx.prefix_shape, dtype=torch.float32, device=dev) if ctx.prefix_grad else None
        cc = _cfg_c(cfg)
        with torch.cuda.device(dev):
            g = grad_out.to(torch.float32).contiguous()
            _lib.check(lib.manner_hip_train_backward(
                C.byref(cc), _table([p.detach() for p in params]), len(params), hip._ptr(ids), n, lp, ctx.m_bound, ctx.prec,
                ctx.start, C.c_float(opts["p_hidden"]), C.c_float(opts["p_attn"]), C.c_float(opts["p_out"]),
                C.c_uint64(opts["seed"]), hip._ptr(g), hip._ptr(ctx.saved_buf), ctx.saved_buf.numel(), _table(grads),
                hip._ptr(gprefix), hip._ptr(ctx.ws), ctx.ws.numel(), hip._stream()))
        ctx.saved_buf = ctx.ws = None
        return (None, None, gprefix, None, *grads)


def encode_train(cfg: EncoderConfig, params: Dict[str, Tensor], ids: Tensor, mask: Tensor, *, precision: str = "f16",
                 p_hidden: float = 0.1, p_attn: float = 0.1, p_out: float = 0.2, seed: int = 0,
                 prefix_engine: Optional[hip.HipEncoder] = None, prefix_hidden: Optional[Tensor] = None,
                 start_layer: Optional[int] = None) -> Tensor:
    """[N, Lp] ids / mask -> [N, H] dropout([CLS]) with autograd into ``params`` (HF-named parameter dict).

    ``start_layer`` / ``prefix_hidden``: explicit cached prefix; by default the prefix is used automatically when no
    tensor below the first trainable layer requires grad and a ``prefix_engine`` (inference HipEncoder over the same
    weights) is given."""
    if precision not in _TRAIN_PRECISIONS:
        raise ValueError(f"training precision {precision!r}: one of {_TRAIN_PRECISIONS}")
    ids, mask = hip._dev(ids, torch.int64, "input_ids").contiguous(), hip._dev(mask, torch.int64, "attention_mask").contiguous()
    if ids.dim() != 2 or ids.shape != mask.shape:
        raise ValueError(f"input_ids {tuple(ids.shape)} / attention_mask {tuple(mask.shape)} must be equal 2-D")
    canon = canonical_weights(cfg, params)
    table = [canon[name] for name in hip.weight_table_order(cfg)]
    for name, t in zip(hip.weight_table_order(cfg), table):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise TypeError(f"{name}: training needs contiguous float32 GPU parameters")
    if start_layer is None:
        start_layer = 0
        emb_frozen = not any(t.requires_grad for t in table[:_lib.W_EMB_COUNT])
        if emb_frozen and prefix_engine is not None:
            first = cfg.layers
            for l in range(cfg.layers):
                if any(t.requires_grad for t in table[_lib.W_EMB_COUNT + l * _lib.WL_COUNT:_lib.W_EMB_COUNT + (l + 1) * _lib.WL_COUNT]):
                    first = l
                    break
            start_layer = min(first, cfg.layers - 1)
    if start_layer > 0 and prefix_hidden is None:
        if prefix_engine is None:
            raise ValueError("start_layer > 0 needs prefix_hidden or a prefix_engine")
        with torch.no_grad():
            prefix_hidden = prefix_engine.encode_hidden(ids, mask, start_layer, precision=precision if precision != "fp32" else "fp32")
    if prefix_hidden is not None:
        prefix_hidden = hip._dev(prefix_hidden, torch.float32, "prefix_hidden").contiguous()
        if tuple(prefix_hidden.shape) != (ids.shape[0], ids.shape[1], cfg.hidden):
            raise ValueError("prefix_hidden must be [N, Lp, H]")
    opts = dict(cfg=cfg, precision=precision, p_hidden=float(p_hidden), p_attn=float(p_attn), p_out=float(p_out),
                seed=int(seed) & (2 ** 64 - 1), start_layer=int(start_layer))
    return _EncodeTrain.apply(ids, mask, prefix_hidden, opts, *table)
