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


class _WeightCopyCache:
    """16-bit copies of FROZEN PLM weights kept across training steps (manner_hip_train_weight_cache, round 4).

    The 16-bit training modes convert every weight matrix to the GEMM operand type — and transpose it for the data-gradient GEMMs —
    on every call; for the layers the reference freezes (``frozen_layers: [0..7]``, news_encoder.py:24-27) the result never changes:
    8 layers x (3 conversions + the Q|K|V pack + 4 transposes) = 0.5 ms of a 15.5 ms step.  An entry belongs to the parameter OBJECTS
    it was made from (round 5): it holds weak references to them, is used only while every reference still yields the very
    tensor passed in, and is dropped by a finalizer the moment one of them is collected — a new model whose fresh parameters land on
    the same allocator addresses with the same version counters can never see a stale copy, and the copies of a model that is gone do
    not stay pinned in HBM.  Within the owners' lifetime an entry is validated by storage (address, size) and the torch version
    counters (an optimizer step, ``load_state_dict`` or any in-place write through the parameter bumps them); a write through
    ``p.data`` bypasses the counter — call ``invalidate_weight_cache()`` after one.  A parameter that requires grad is never cached.
    The copies themselves are made by the library on first use — the same kernels as without the cache, so the cached path is
    bit-identical — this class only owns the buffers and the validity flags."""

    def __init__(self):
        self._entries = {}

    def clear(self):
        self._entries.clear()

    def __len__(self):
        return len(self._entries)

    def _drop(self, key):
        self._entries.pop(key, None)

    def arrays(self, cfg, params, needs_grad, prec, start, dev):
        import weakref
        n = len(params)
        h, i_ = cfg.hidden, cfg.intermediate
        esz = 2
        slots = (C.c_void_p * (2 * n))()
        valid = (C.c_int32 * (2 * n))()
        touched = []
        base0 = _lib.W_EMB_COUNT
        for l in range(start, cfg.layers):
            b = base0 + l * _lib.WL_COUNT
            groups = (("qkv", (0, 2, 4), 0, 3 * h * h * esz, True), ("qkv_bias", (1, 3, 5), 1, 3 * h * 4, False),
                      ("ao", (6,), 6, h * h * esz, True), ("ff1", (10,), 10, i_ * h * esz, True), ("ff2", (12,), 12, h * i_ * esz, True))
            for what, deps, at, nbytes, both in groups:
                if any(needs_grad[b + d] for d in deps):
                    continue
                owners = [params[b + d] for d in deps]
                key = (str(dev), prec, l, what) + tuple(id(o) for o in owners)
                ent = self._entries.get(key)
                vers = tuple(int(o._version) for o in owners)
                ptrs = tuple((o.data_ptr(), o.numel()) for o in owners)
                if ent is not None and not all(r() is o for r, o in zip(ent["owners"], owners)):
                    ent = None                       # an id() re-used by another object: never this entry's tensors
                if ent is None or ent["ptrs"] != ptrs:
                    ent = {"ptrs": ptrs, "vers": None, "owners": [weakref.ref(o) for o in owners],
                           "bufs": [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(2 if both else 1)], "valid": [0, 0]}
                    self._entries[key] = ent
                    for o in owners:                 # the entry dies with the first of its owners
                        weakref.finalize(o, self._drop, key)
                if ent["vers"] != vers:
                    ent["valid"] = [0, 0]
                    ent["vers"] = vers
                for c in range(2 if both else 1):
                    slots[2 * (b + at) + c] = ent["bufs"][c].data_ptr()
                    valid[2 * (b + at) + c] = ent["valid"][c]
                touched.append((ent, 2 * (b + at), 2 if both else 1))
        return slots, valid, touched

    @staticmethod
    def commit(valid, touched):
        for ent, s0, cnt in touched:
            for c in range(cnt):
                ent["valid"][c] = int(valid[s0 + c])


_WCACHE = _WeightCopyCache()


def invalidate_weight_cache() -> None:
    """Forget every cached 16-bit copy of frozen weights (they are rebuilt on the next training call).  Needed only after a write that
    torch's version counter cannot see (``p.data.copy_(...)``, a raw pointer write); optimiser steps, ``load_state_dict`` and in-place
    ops through the parameter are detected by themselves."""
    _WCACHE.clear()


def _register_weight_cache(lib, cfg, params, needs_grad, opts, start, dev):
    """Hands the frozen weights' cached 16-bit copies to the next train_forward / _backward call (16-bit modes; MANNER_TRAIN_WEIGHT_CACHE=0
    switches it off for A/B).  Returns what ``_WeightCopyCache.commit`` needs after the call."""
    import os
    if opts["precision"] == "fp32" or os.environ.get("MANNER_TRAIN_WEIGHT_CACHE", "1") == "0":
        return None
    slots, valid, touched = _WCACHE.arrays(cfg, params, needs_grad, opts["precision"], start, dev)
    if not touched:
        return None
    _lib.check(lib.manner_hip_train_weight_cache(slots, valid, len(slots)))
    return slots, valid, touched


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
            saved = torch.empty(int(lib.manner_hip_train_saved_bytes_for(C.byref(cc), n, m_bound, start, prec)), dtype=torch.uint8, device=dev)
            ws = torch.empty(int(lib.manner_hip_train_workspace_bytes(C.byref(cc), m_bound)), dtype=torch.uint8, device=dev)
            out = torch.empty((n, cfg.hidden), dtype=torch.float32, device=dev)
            weights = [p.detach() for p in params]
            status = hip.device_status(dev)
            status.poll()                     # flags of earlier calls (a token_bound below the mask's count, a bad id) raise here
            needs = [bool(p.requires_grad) for p in params]
            wc = _register_weight_cache(lib, cfg, params, needs, opts, start, dev)
            _lib.check(lib.manner_hip_train_forward(
                C.byref(cc), _table(weights), len(weights), hip._ptr(ids), hip._ptr(mask), n, lp, m_bound, prec, start,
                hip._ptr(prefix), C.c_float(opts["p_hidden"]), C.c_float(opts["p_attn"]), C.c_float(opts["p_out"]),
                C.c_uint64(opts["seed"]), hip._ptr(out), hip._ptr(saved), saved.numel(), hip._ptr(ws), ws.numel(),
                hip._ptr(status.word), hip._stream()))
            ctx.layout = int(lib.manner_hip_train_layout_last())    # ABI v8: the saved buffer's layout travels with the autograd ctx
            if wc is not None:
                _WeightCopyCache.commit(wc[1], wc[2])
            status.arm()                      # snapshot behind an event: examined, without blocking, by the next poll
        ctx.opts, ctx.m_bound, ctx.prec, ctx.start = opts, m_bound, prec, start
        ctx.needs = needs
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
            wc = _register_weight_cache(lib, cfg, params, ctx.needs, opts, ctx.start, dev)
            _lib.check(lib.manner_hip_train_layout_next(ctx.layout))
            _lib.check(lib.manner_hip_train_backward(
                C.byref(cc), _table([p.detach() for p in params]), len(params), hip._ptr(ids), n, lp, ctx.m_bound, ctx.prec,
                ctx.start, C.c_float(opts["p_hidden"]), C.c_float(opts["p_attn"]), C.c_float(opts["p_out"]),
                C.c_uint64(opts["seed"]), hip._ptr(g), hip._ptr(ctx.saved_buf), ctx.saved_buf.numel(), _table(grads),
                hip._ptr(gprefix), hip._ptr(ctx.ws), ctx.ws.numel(), hip._stream()))
            if wc is not None:
                _WeightCopyCache.commit(wc[1], wc[2])
        return (None, None, gprefix, None, *grads)       # the activation buffer lives with ctx: backward(retain_graph=True) may run again


def first_trainable_layer(cfg: EncoderConfig, params: Dict[str, Tensor]) -> int:
    """Index of the layer training starts at when everything below it is frozen — 0 when an embedding table trains (the reference's
    default: the gradient then crosses every layer), else the first layer with a trainable tensor, capped at the last layer."""
    canon = canonical_weights(cfg, params)
    table = [canon[name] for name in hip.weight_table_order(cfg)]
    if any(t.requires_grad for t in table[:_lib.W_EMB_COUNT]):
        return 0
    for l in range(cfg.layers):
        if any(t.requires_grad for t in table[_lib.W_EMB_COUNT + l * _lib.WL_COUNT:_lib.W_EMB_COUNT + (l + 1) * _lib.WL_COUNT]):
            return min(l, cfg.layers - 1)
    return cfg.layers - 1


def encode_train(cfg: EncoderConfig, params: Dict[str, Tensor], ids: Tensor, mask: Tensor, *, precision: str = "f16",
                 p_hidden: float = 0.1, p_attn: float = 0.1, p_out: float = 0.2, seed: int = 0,
                 prefix_engine: Optional[hip.HipEncoder] = None, prefix_hidden: Optional[Tensor] = None,
                 start_layer: Optional[int] = None, token_bound: Optional[int] = None) -> Tensor:
    """[N, Lp] ids / mask -> [N, H] dropout([CLS]) with autograd into ``params`` (HF-named parameter dict).

    ``start_layer`` / ``prefix_hidden``: explicit cached prefix; by default the prefix is used automatically when no
    tensor below the first trainable layer requires grad and a ``prefix_engine`` (inference HipEncoder over the same
    weights) is given.  ``token_bound``: a host-known upper bound of the real tokens (e.g. the collate's sum of lengths) —
    activation buffers are sized for it instead of N * Lp; a mask with more tokens is truncated at the bound (never indexed past
    it) and raises ``RuntimeError`` through the device status word: at the next training / scoring call (non-blocking poll) or
    at ``hip.check_status`` (blocking)."""
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
                seed=int(seed) & (2 ** 64 - 1), start_layer=int(start_layer),
                m_bound=None if token_bound is None else (max(int(token_bound), 1) + 255) // 256 * 256)
    return _EncodeTrain.apply(ids, mask, prefix_hidden, opts, *table)


class _EncodeFullTrain(torch.autograd.Function):
    """HF last_hidden_state [N, Lp, H] INCLUDING the padded positions, with autograd into the PLM parameters (the PLM of
    PLMTextEncoder in train() mode, news_encoder.py:160-171)."""

    @staticmethod
    def forward(ctx, ids: Tensor, mask: Tensor, opts: dict, *params: Tensor):
        cfg: EncoderConfig = opts["cfg"]
        lib = _lib.load()
        n, lp = ids.shape
        m_bound = (n * lp + 255) // 256 * 256
        prec = _lib.PRECISIONS[opts["precision"]]
        cc = _cfg_c(cfg)
        dev = ids.device
        with torch.cuda.device(dev):
            saved = torch.empty(int(lib.manner_hip_train_saved_bytes(C.byref(cc), n, m_bound, 0)), dtype=torch.uint8, device=dev)
            ws = torch.empty(int(lib.manner_hip_train_workspace_bytes(C.byref(cc), m_bound)), dtype=torch.uint8, device=dev)
            out = torch.empty((n, lp, cfg.hidden), dtype=torch.float32, device=dev)
            weights = [p.detach() for p in params]
            status = hip.device_status(dev)
            status.poll()
            _lib.check(lib.manner_hip_train_full_forward(
                C.byref(cc), _table(weights), len(weights), hip._ptr(ids), hip._ptr(mask), n, lp, prec, C.c_float(opts["p_hidden"]),
                C.c_float(opts["p_attn"]), C.c_uint64(opts["seed"]), hip._ptr(out), hip._ptr(saved), saved.numel(), hip._ptr(ws), ws.numel(),
                hip._ptr(status.word), hip._stream()))
            ctx.layout = int(lib.manner_hip_train_layout_last())
            status.arm()
        ctx.opts, ctx.prec = opts, prec
        ctx.saved_buf, ctx.ws = saved, ws
        ctx.save_for_backward(ids, *params)
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        ids, *params = ctx.saved_tensors
        opts, cfg = ctx.opts, ctx.opts["cfg"]
        lib = _lib.load()
        n, lp = ids.shape
        need = ctx.needs_input_grad[3:]
        grads: List[Optional[Tensor]] = [torch.empty_like(p) if r else None for p, r in zip(params, need)]
        cc = _cfg_c(cfg)
        with torch.cuda.device(ids.device):
            g = grad_out.to(torch.float32).contiguous()
            _lib.check(lib.manner_hip_train_layout_next(ctx.layout))
            _lib.check(lib.manner_hip_train_full_backward(
                C.byref(cc), _table([p.detach() for p in params]), len(params), hip._ptr(ids), n, lp, ctx.prec, C.c_float(opts["p_hidden"]),
                C.c_float(opts["p_attn"]), C.c_uint64(opts["seed"]), hip._ptr(g), hip._ptr(ctx.saved_buf), ctx.saved_buf.numel(), _table(grads),
                hip._ptr(ctx.ws), ctx.ws.numel(), hip._stream()))
        return (None, None, None, *grads)


def encode_full_train(cfg: EncoderConfig, params: Dict[str, Tensor], ids: Tensor, mask: Tensor, *, precision: str = "fp32",
                      p_hidden: float = 0.1, p_attn: float = 0.1, seed: int = 0) -> Tensor:
    """[N, Lp] ids / mask -> HF ``last_hidden_state`` [N, Lp, H] including the padded positions (``hip.encode_full`` with
    autograd and HF's dropouts): what ``PLMTextEncoder`` feeds its un-masked attention in train() mode."""
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
    opts = dict(cfg=cfg, precision=precision, p_hidden=float(p_hidden), p_attn=float(p_attn), seed=int(seed) & (2 ** 64 - 1))
    return _EncodeFullTrain.apply(ids, mask, opts, *table)


# ---------------------------------------------------------------------------------------------- scorer and loss
class _LateFusion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hist: Tensor, hist_off: Tensor, cand: Tensor, cand_off: Tensor):
        b, d = hist_off.numel() - 1, hist.shape[1]
        user = torch.empty((b, d), dtype=torch.float32, device=hist.device)
        scores = torch.empty(cand.shape[0], dtype=torch.float32, device=hist.device)
        with torch.cuda.device(hist.device):
            _lib.check(_lib.load().manner_hip_late_fusion_train_forward(hip._ptr(hist), hip._ptr(hist_off), hip._ptr(cand), hip._ptr(cand_off),
                                                                        b, d, hip._ptr(user), hip._ptr(scores), hip._stream()))
        ctx.save_for_backward(user, hist_off, cand, cand_off)
        ctx.n_hist = hist.shape[0]
        return scores

    @staticmethod
    def backward(ctx, g: Tensor):
        user, hist_off, cand, cand_off = ctx.saved_tensors
        b, d = user.shape
        dhist = torch.empty((ctx.n_hist, d), dtype=torch.float32, device=user.device)
        dcand = torch.empty_like(cand)
        with torch.cuda.device(user.device):
            g = g.to(torch.float32).contiguous()
            _lib.check(_lib.load().manner_hip_late_fusion_train_backward(hip._ptr(g), hip._ptr(user), hip._ptr(hist_off), hip._ptr(cand),
                                                                         hip._ptr(cand_off), b, d, hip._ptr(dhist), hip._ptr(dcand), hip._stream()))
        return dhist, None, dcand, None


def late_fusion_scores(hist: Tensor, hist_off: Tensor, cand: Tensor, cand_off: Tensor) -> Tensor:
    """Ragged scores of CRModule.forward(late_fusion=True) on per-occurrence vectors (cr_module.py:105-131), differentiable:
    hist [sum h_i, D], cand [sum c_i, D] f32, offsets int64 [B+1] on the GPU -> scores [sum c_i]."""
    hist, cand = hip._dev(hist, torch.float32, "hist").contiguous(), hip._dev(cand, torch.float32, "cand").contiguous()
    hist_off, cand_off = hip._dev(hist_off, torch.int64, "hist_off").contiguous(), hip._dev(cand_off, torch.int64, "cand_off").contiguous()
    if hist.shape[1] != cand.shape[1] or hist_off.numel() != cand_off.numel():
        raise ValueError("late_fusion_scores: mismatching shapes")
    return _LateFusion.apply(hist, hist_off, cand, cand_off)


class _Dot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, user: Tensor, cand: Tensor):
        ctx.save_for_backward(user, cand)
        return hip.dot(user.detach(), cand.detach())

    @staticmethod
    def backward(ctx, g: Tensor):
        user, cand = ctx.saved_tensors
        b, _, d = user.shape
        c = cand.shape[2]
        du = torch.empty((b, 1, d), dtype=torch.float32, device=user.device)
        dc = torch.empty((b, d, c), dtype=torch.float32, device=user.device)
        with torch.cuda.device(user.device):
            g = g.to(torch.float32).contiguous()
            uc = user.contiguous()
            _lib.check(_lib.load().manner_hip_dot_backward(hip._ptr(g), hip._ptr(uc), hip._ptr(cand), b, c, d, cand.stride(0), cand.stride(1),
                                                           cand.stride(2), hip._ptr(du), hip._ptr(dc), hip._stream()))
        return du, dc


def dot(user: Tensor, cand: Tensor) -> Tensor:
    """DotProduct.forward (click_predictors.py:9-12) with autograd: user [B, 1, D], cand [B, D, C] (any strides)."""
    return _Dot.apply(hip._dev(user, torch.float32, "user_vector"), hip._dev(cand, torch.float32, "candidate_news_vector"))


class _Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores: Tensor, labels: Tensor, cand_off: Tensor, mode: int, temperature: float, c_max: int):
        b = cand_off.numel() - 1
        losses = torch.empty(b, dtype=torch.float32, device=scores.device)
        red = torch.empty(2, dtype=torch.float32, device=scores.device)
        grad = torch.empty_like(scores)
        with torch.cuda.device(scores.device):
            _lib.check(_lib.load().manner_hip_train_loss(hip._ptr(scores), hip._ptr(labels), hip._ptr(cand_off), b, mode, C.c_float(temperature),
                                                         c_max, hip._ptr(losses), hip._ptr(red), hip._ptr(grad), hip._stream()))
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(losses)
        return red[0], losses

    @staticmethod
    def backward(ctx, g: Tensor, _g_losses):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None


def model_step_loss(scores: Tensor, labels: Tensor, cand_off: Tensor, supcon: bool = True, temperature: float = 0.1,
                    c_max: Optional[int] = None):
    """The loss of CRModule.model_step (cr_module.py:140-171) on ragged scores, with autograd: SupConLoss on the score matrix
    (losses.py:12-40, mean over the non-zero per-impression losses) or nn.CrossEntropyLoss over the dense zero-padded rows
    (``c_max`` = the batch's largest candidate count, e.g. ``batch["cand_max"]`` from DeviceCollate).
    Returns (batch loss scalar, per-impression losses [B])."""
    scores = hip._dev(scores, torch.float32, "scores").contiguous()
    labels = hip._dev(labels, torch.float32, "labels").contiguous()
    cand_off = hip._dev(cand_off, torch.int64, "cand_off").contiguous()
    if not supcon and c_max is None:
        raise ValueError("cross-entropy mode needs c_max (the dense row width of the reference)")
    # nn.CrossEntropyLoss has no temperature (cr_module.py:169)
    return _Loss.apply(scores, labels, cand_off, 0 if supcon else 1, float(temperature) if supcon else 1.0, int(c_max or 1))


# ---------------------------------------------------------------------------------------------- small differentiable operators
# (train_small.hip) — what the reference's default `use_entities: True` and early fusion need around the text encoder
def _f32(t: Tensor, name: str) -> Tensor:
    return hip._dev(t, torch.float32, name).contiguous()


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor]):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        shape = x.shape
        y = hip.linear(x.reshape(-1, shape[-1]), weight, bias)
        return y.reshape(*shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, g: Tensor):
        x, weight = ctx.saved_tensors
        o, k = weight.shape
        x2, g2 = x.reshape(-1, k), _f32(g.reshape(-1, o), "grad")
        r = x2.shape[0]
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x2) if need_x else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(o, dtype=torch.float32, device=x.device) if need_b else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().manner_hip_linear_backward(hip._ptr(x2), hip._ptr(weight), hip._ptr(g2), r, k, o, hip._ptr(None),
                                                              hip._ptr(dx), hip._ptr(dw), hip._ptr(db), hip._stream()))
        return (dx.reshape(x.shape) if need_x else None), dw, db


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """nn.Linear with autograd on the HIP kernels (f32): x [..., K], weight [O, K]."""
    return _Linear.apply(_f32(x, "x"), _f32(weight, "weight"), None if bias is None else _f32(bias, "bias"))


class _AdditivePool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, lin_w: Tensor, lin_b: Tensor, query: Tensor):
        ctx.save_for_backward(x, lin_w, lin_b, query)
        return hip.additive_pool(x, lin_w, lin_b, query, strict=True)      # the backward recomputes the exact-f32 pre-activations

    @staticmethod
    def backward(ctx, g: Tensor):
        x, lin_w, lin_b, query = ctx.saved_tensors
        b, s, d = x.shape
        q = lin_w.shape[0]
        lib = _lib.load()
        dx, dw, db, dq = torch.empty_like(x), torch.empty_like(lin_w), torch.empty_like(lin_b), torch.empty_like(query)
        with torch.cuda.device(x.device):
            need = int(lib.manner_hip_additive_pool_backward_workspace_bytes(b, s, d, q))
            ws = torch.empty(need, dtype=torch.uint8, device=x.device)
            g = _f32(g, "grad")
            _lib.check(lib.manner_hip_additive_pool_backward(hip._ptr(x), hip._ptr(lin_w), hip._ptr(lin_b), hip._ptr(query), hip._ptr(g), b, s, d,
                                                             q, hip._ptr(dx), hip._ptr(dw), hip._ptr(db), hip._ptr(dq), hip._ptr(ws), need,
                                                             hip._stream()))
        return dx, dw, db, dq


def additive_pool(x: Tensor, lin_w: Tensor, lin_b: Tensor, query: Tensor) -> Tensor:
    """AdditiveAttention.forward (attention.py:21-27) with autograd: x [B, S, D] -> [B, D]."""
    return _AdditivePool.apply(_f32(x, "input_vector"), _f32(lin_w, "linear.weight"), _f32(lin_b, "linear.bias"), _f32(query, "query"))


class _Axis0Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv: Tensor, heads: int):
        l0, b1, e3 = qkv.shape
        out = torch.empty((l0, b1, e3 // 3), dtype=torch.float32, device=qkv.device)
        with torch.cuda.device(qkv.device):
            _lib.check(_lib.load().manner_hip_axis0_attention(hip._ptr(qkv), l0, b1, e3 // 3, heads, hip._ptr(out), hip._stream()))
        ctx.save_for_backward(qkv)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        (qkv,) = ctx.saved_tensors
        l0, b1, e3 = qkv.shape
        dqkv = torch.empty_like(qkv)
        stats = torch.empty(l0 * b1 * ctx.heads * 3, dtype=torch.float32, device=qkv.device)
        with torch.cuda.device(qkv.device):
            g = _f32(g, "grad")
            _lib.check(_lib.load().manner_hip_axis0_attention_backward(hip._ptr(qkv), hip._ptr(g), l0, b1, e3 // 3, ctx.heads, hip._ptr(dqkv),
                                                                       hip._ptr(stats), hip._stream()))
        return dqkv, None


def mha_axis0(x: Tensor, in_proj_w: Tensor, in_proj_b: Tensor, out_proj_w: Tensor, out_proj_b: Tensor, heads: int) -> Tensor:
    """nn.MultiheadAttention(batch_first=False) as the reference calls it on [batch, seq, E] without masks (quirk Q1), with
    autograd: in-projection, attention along axis 0, out-projection."""
    qkv = linear(x, in_proj_w, in_proj_b)
    return linear(_Axis0Attention.apply(qkv.contiguous(), int(heads)), out_proj_w, out_proj_b)


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids: Tensor, table: Tensor, padding_idx: int):
        out = torch.empty(tuple(ids.shape) + (table.shape[1],), dtype=torch.float32, device=table.device)
        with torch.cuda.device(table.device):
            _lib.check(_lib.load().manner_hip_embedding(hip._ptr(ids), ids.numel(), hip._ptr(table), table.shape[0], table.shape[1], hip._ptr(out),
                                                        hip._ptr(hip.device_status(table.device).word), hip._stream()))
        ctx.save_for_backward(ids)
        ctx.shape, ctx.padding_idx = tuple(table.shape), padding_idx
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        (ids,) = ctx.saved_tensors
        dt = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            g = _f32(g, "grad")
            _lib.check(_lib.load().manner_hip_embedding_backward(hip._ptr(ids), ids.numel(), hip._ptr(g), ctx.shape[0], ctx.shape[1],
                                                                 -1 if ctx.padding_idx is None else int(ctx.padding_idx), hip._ptr(dt), hip._stream()))
        return None, dt, None


def embedding(ids: Tensor, table: Tensor, padding_idx: Optional[int] = None) -> Tensor:
    """nn.Embedding lookup with autograd (row ``padding_idx`` receives no gradient)."""
    return _Embedding.apply(hip._dev(ids, torch.int64, "ids").contiguous(), _f32(table, "embedding.weight"), padding_idx)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, p: float, seed: int, site: int):
        ctx.args = (p, seed, site)
        return _Dropout._run(x, p, seed, site)

    @staticmethod
    def _run(x, p, seed, site):
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().manner_hip_dropout(hip._ptr(x), hip._ptr(out), x.numel(), C.c_uint64(seed), C.c_uint32(site), C.c_float(p), hip._stream()))
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        return _Dropout._run(_f32(g, "grad"), *ctx.args), None, None, None


def dropout(x: Tensor, p: float, seed: int, site: int = 0) -> Tensor:
    """nn.Dropout in train() mode on the training path's counter-based generator (mask = f(seed, site, element index))."""
    if p <= 0.0:
        return x
    return _Dropout.apply(_f32(x, "x"), float(p), int(seed) & (2 ** 64 - 1), int(site))


class _SupConEmbeddings(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb: Tensor, labels: Tensor, temperature: float):
        n, d = emb.shape
        lib = _lib.load()
        losses = torch.empty(n, dtype=torch.float32, device=emb.device)
        red = torch.empty(2, dtype=torch.float32, device=emb.device)
        grad = torch.empty_like(emb)
        with torch.cuda.device(emb.device):
            need = int(lib.manner_hip_supcon_embeddings_workspace_bytes(n))
            ws = torch.empty(need, dtype=torch.uint8, device=emb.device)
            _lib.check(lib.manner_hip_supcon_embeddings(hip._ptr(emb), hip._ptr(labels), n, d, C.c_float(temperature), hip._ptr(losses), hip._ptr(red),
                                                        hip._ptr(grad), hip._ptr(ws), need, hip._stream()))
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(losses)
        return red[0], losses

    @staticmethod
    def backward(ctx, g: Tensor, _g_losses):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def supcon_embedding_loss(embeddings: Tensor, labels: Tensor, temperature: float = 0.1):
    """The A-Module's criterion (a_module.py:73-75,102-108: pytorch_metric_learning SupConLoss with an un-normalised dot-product
    similarity) on news embeddings [N, D] and aspect labels [N], with autograd.  Returns (batch loss, per-anchor losses)."""
    emb = hip._dev(embeddings, torch.float32, "embeddings").contiguous()
    lab = hip._dev(labels, torch.int64, "labels").contiguous()
    if emb.dim() != 2 or lab.shape != (emb.shape[0],):
        raise ValueError("supcon_embedding_loss: embeddings [N, D], labels [N]")
    return _SupConEmbeddings.apply(emb, lab, float(temperature))
