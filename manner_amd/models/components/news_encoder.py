"""MannerTextEncoder / MannerEntityEncoder / MannerNewsEncoder — mirror of reference
manner/models/components/news_encoder.py:11-129.

Same constructor arguments, forward signature and state_dict keys (``text_encoder.plm_model.*`` are
the HF BertModel / RobertaModel names), so reference checkpoints load with
``load_state_dict``.  ``forward`` (GPU tensors) runs K1-K7 through libmanner_hip.so in eval mode and, in train() mode,
the training path of SURVEY §8f-3 (``manner_amd.train``: dropouts + autograd into the parameters); there is no CPU path.
"""
from __future__ import annotations

import json
import operator
import os
import warnings
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from manner_amd import hip, train
from manner_amd.config import EncoderConfig, resolve
from manner_amd.models.components.attention import AdditiveAttention
from manner_amd.weights import plm_param_shapes


_warned_eval_graph = False

# Optimiser steps taken in this process.  The inference handle keys its packed weight copies on (storage address, version counter) of
# the parameters — but a FUSED optimiser (torch.optim.AdamW(fused=True), with or without a GradScaler) rewrites them without bumping
# Parameter._version (measured on torch 2.10 / ROCm: 0 -> 0 while the values change, tools/version_probe.py; the foreach and the
# single-tensor forms bump it).  So ANY optimiser step — the global post-step hook below counts them — makes the next eval() forward
# repack: at most one repack per (step, eval forward) pair, i.e. nothing inside a training epoch and one rebuild when validation starts.
_OPT_STEPS = [0]


def _count_optimizer_step(*_args, **_kwargs) -> None:
    _OPT_STEPS[0] += 1


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _register_post_step
    _register_post_step(_count_optimizer_step)
except ImportError:                                      # torch < 2.0: the sampled fingerprint (weight_fingerprint) is the only tripwire
    pass


def autocast_mode(device_type: str = "cuda") -> Optional[str]:
    """The 16-bit mode the CALLER's autocast state asks for: "f16" under ``torch.autocast("cuda", torch.float16)`` — what Lightning's
    ``precision: 16-mixed`` plugin wraps every ``*_step`` in (reference configs/trainer/default.yaml:12) — "bf16" under bfloat16
    autocast (``bf16-mixed``), None outside any autocast region (``trainer.precision=32``).  The reference's modules are plain torch,
    so this state IS their arithmetic; the mirrors read the same knob instead of a switch of their own."""
    try:
        if not torch.is_autocast_enabled(device_type):
            return None
        dt = torch.get_autocast_dtype(device_type)
    except TypeError:                                    # torch < 2.4: the argument-free CUDA forms
        if not torch.is_autocast_enabled():
            return None
        dt = torch.get_autocast_gpu_dtype()
    return "f16" if dt == torch.float16 else "bf16" if dt == torch.bfloat16 else None


def _wants_graph(module: nn.Module, *inputs) -> bool:
    """Does autograd have to record this forward?  The reference's modules are plain torch: whenever grad mode is on and a
    parameter (or an input) requires grad, their output carries a grad_fn — in eval() too (dropout-free fine-tuning,
    gradient attribution).  The mirrors follow the same rule: the differentiable engine runs then, with the dropouts
    switched by ``module.training`` alone; the inference engine serves eval() under no_grad / inference_mode (what Lightning's
    validation and test loops use) or a fully frozen module."""
    if not torch.is_grad_enabled():
        return False
    return any(p.requires_grad for p in module.parameters()) or any(isinstance(t, torch.Tensor) and t.requires_grad for t in inputs)


def _warn_eval_graph(name: str) -> None:
    global _warned_eval_graph
    if not _warned_eval_graph:
        _warned_eval_graph = True
        warnings.warn(f"{name}: eval() forward with grad mode on and trainable parameters — running the differentiable engine "
                      "(dropout off), as the reference would build a graph here; wrap inference in torch.no_grad() / "
                      "inference_mode() to get the MFMA inference engine")


def _set_nested_parameter(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


class HipPLM(nn.Module):
    """Parameter tree with HF BertModel/RobertaModel names; the arithmetic lives in the HIP library."""

    def __init__(self, cfg: EncoderConfig, hidden_dropout_prob: float = 0.1, attention_probs_dropout_prob: float = 0.1) -> None:
        super().__init__()
        self.cfg = cfg
        # HF BertConfig / RobertaConfig defaults (DistilBertConfig: dropout / attention_dropout, also 0.1): train() mode only
        self.hidden_dropout_prob = hidden_dropout_prob
        self.attention_probs_dropout_prob = attention_probs_dropout_prob
        for name, shape in plm_param_shapes(cfg, with_pooler=True):
            t = torch.empty(shape, dtype=torch.float32)
            if name.endswith(("LayerNorm.weight", "layer_norm.weight")):
                t.fill_(1.0)
            elif name.endswith(".bias"):
                t.zero_()
            else:
                t.normal_(0.0, 0.02)            # HF initializer_range
            _set_nested_parameter(self, name, nn.Parameter(t))

    @property
    def base_model(self) -> "HipPLM":           # reference freezes via plm_model.base_model (news_encoder.py:24)
        return self

    @classmethod
    def from_pretrained(cls, plm_model: str) -> "HipPLM":
        """Local HF directory (config.json + model.safetensors / pytorch_model.bin) or an architecture
        preset name.  There is no hub access: a preset name yields random HF-style init and expects a
        checkpoint to be loaded afterwards (as EnsembleModule does, ensemble_module.py:33-46)."""
        model = cls(resolve(plm_model))
        if os.path.isdir(plm_model):
            with open(os.path.join(plm_model, "config.json")) as f:
                hf = json.load(f)
            model.hidden_dropout_prob = float(hf.get("hidden_dropout_prob", hf.get("dropout", 0.1)))
            model.attention_probs_dropout_prob = float(hf.get("attention_probs_dropout_prob", hf.get("attention_dropout", 0.1)))
            sd = None
            st, pt = os.path.join(plm_model, "model.safetensors"), os.path.join(plm_model, "pytorch_model.bin")
            if os.path.exists(st):
                from safetensors.torch import load_file
                sd = load_file(st)
            elif os.path.exists(pt):
                sd = torch.load(pt, map_location="cpu", weights_only=True)
            if sd is None:
                raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {plm_model}")
            own = model.state_dict()
            fixed = {}
            for k, v in sd.items():
                for pre in ("bert.", "roberta.", "distilbert.", ""):
                    if k.startswith(pre) and k[len(pre):] in own:
                        fixed[k[len(pre):]] = v
                        break
            missing = [k for k in own if k not in fixed and not k.startswith("pooler.")]
            if missing:
                raise KeyError(f"{plm_model}: checkpoint lacks {missing[:4]}... ({len(missing)} tensors)")
            model.load_state_dict(fixed, strict=False)
        else:
            warnings.warn(f"PLM {plm_model!r}: no hub access — random init of the {plm_model} architecture; "
                          "load a checkpoint before use")
        return model


class _ParamView:
    """``dict(module.named_parameters())`` without walking the module tree on every forward: 250 us for bert-base, paid twice per
    batch by the unchanged ``CRModule.forward`` right behind a host synchronisation of the reference's own code, i.e. with the GPU idle.
    The view remembers every ``_modules`` / ``_parameters`` dict of the tree and checks, per call, that each still holds the very
    objects it held and still IS the container its module owns (a submodule or Parameter replaced by assignment, an adapter added,
    ``.to()`` with overwrite-on-conversion, a wrapper that swaps ``module._parameters`` wholesale: any of these rebuilds the view) — 350 dict look-ups run by ``map`` / ``operator.is_`` without a Python-level loop.  Same names, order and
    de-duplication as ``named_parameters()``.  ``state()`` is the (storage address, version counter) fingerprint of the parameters in
    that order — what tells an optimiser step, ``load_state_dict`` or a ``p.data`` swap from "nothing changed"."""

    def __init__(self, root: nn.Module) -> None:
        self.root = root
        self._build()

    def _build(self) -> None:
        # built into locals and published by ONE attribute assignment: a second thread in get() sees the old or the new snapshot whole
        len_dicts, len_vals, dicts, keys, objs, named, seen = [], [], [], [], [], {}, set()
        owners, attrs = [], []          # the module __dict__ each container hangs in: a container REPLACED wholesale is seen too

        def walk(m: nn.Module, prefix: str) -> None:
            for a in ("_modules", "_parameters"):
                d = m.__dict__[a]
                owners.append(m.__dict__); attrs.append(a)
                len_dicts.append(d)
                len_vals.append(len(d))
            for n, p in m._parameters.items():
                dicts.append(m._parameters); keys.append(n); objs.append(p)
                if p is not None and id(p) not in seen:
                    seen.add(id(p))
                    named[prefix + n] = p
            for n, c in m._modules.items():
                dicts.append(m._modules); keys.append(n); objs.append(c)
                if c is not None:
                    walk(c, prefix + n + ".")
        walk(self.root, "")
        self._snap = (len_dicts, len_vals, dicts, keys, objs, named, list(named.values()), owners, attrs)

    @property
    def named(self) -> dict:
        return self._snap[5]

    def get(self) -> dict:
        """The name -> Parameter dict (shared between calls: do not mutate it)."""
        len_dicts, len_vals, dicts, keys, objs, named, _, owners, attrs = self._snap
        if (list(map(len, len_dicts)) != len_vals or not all(map(operator.is_, map(dict.get, dicts, keys), objs))
                or not all(map(operator.is_, map(dict.get, owners, attrs), len_dicts))):
            self._build()
            return self._snap[5]
        return named

    def state(self, named: Optional[dict] = None) -> Optional[tuple]:
        """(data_ptr, _version) of every parameter, as two lists built without a Python-level loop.  ``named``: the dict get() returned
        to this caller — None comes back if the view has been rebuilt since (another thread), so that a fingerprint never pairs with
        another snapshot's parameters."""
        snap = self._snap
        if named is not None and snap[5] is not named:
            return None
        return list(map(_DATA_PTR, snap[6])), list(map(_VERSION, snap[6]))


_DATA_PTR = torch.Tensor.data_ptr
_VERSION = operator.attrgetter("_version")


class MannerTextEncoder(nn.Module):
    """reference news_encoder.py:11-37."""

    #: Arithmetic of the HIP inference engine.  None (the default) = FOLLOW THE CALLER, as the reference's plain-torch module does:
    #: under fp16 autocast (Lightning `precision: 16-mixed`, the shipped configs/trainer/default.yaml:12) "f16" — the MFMA path on
    #: IEEE half; under bf16 autocast (`bf16-mixed`) "bf16"; outside autocast (`trainer.precision=32`) the parity-grade mode
    #: "f16x3" (f32 activations, split-operand f16 GEMMs with f32 accumulation: within 1e-4 of the reference's fp32 path, ranking
    #: indices as the oracle's).  A string pins one mode whatever the caller's state ("f16", "bf16", "fp32" = f32 MFMA, "f16x3",
    #: "bf16x3"); the environment variable MANNER_HIP_PRECISION pins it for a whole process (A/B runs).
    precision: Optional[str] = os.environ.get("MANNER_HIP_PRECISION") or None

    def resolved_precision(self) -> str:
        """The mode the next eval() forward computes in (see ``precision``)."""
        if self.precision is not None:
            return self.precision
        ac = autocast_mode()
        if ac is not None:
            return ac
        cfg = self.plm_model.cfg                       # the split-operand GEMMs tile K in 256s: tiny test models take the f32 MFMA mode
        return "f16x3" if cfg.hidden % 256 == 0 and cfg.intermediate % 256 == 0 else "fp32"

    def resolved_train_precision(self) -> str:
        """The GEMM arithmetic of the next train() forward (see ``train_precision``)."""
        if self.train_precision is not None:
            return self.train_precision
        return autocast_mode() or "fp32"

    def __init__(self, plm_model: str, frozen_layers: List[int], dropout_probability: float) -> None:
        super().__init__()
        self.plm_model = HipPLM.from_pretrained(plm_model)
        self.dropout = nn.Dropout(p=dropout_probability)
        # freeze PLM layers (same name test as the reference, news_encoder.py:24-27)
        for name, param in self.plm_model.base_model.named_parameters():
            for layer in frozen_layers:
                if "layer." + str(layer) + "." in name:
                    param.requires_grad = False
        self._hip: Optional[hip.HipEncoder] = None
        self._hip_key = None
        self._cache: Optional[hip.NewsEmbeddingCache] = None

    #: Opt-in memoisation of the eval() forward (SURVEY §8d "mode T" behind the unchanged call pattern): rows of an
    #: `hip.NewsEmbeddingCache` in HBM, 0 = off (the default: every occurrence is encoded, as in the reference).  With it a row
    #: whose real tokens were seen before under the same weights and precision is returned from the table — bit-identical to
    #: encoding it again, because the inference engine computes a news from its own tokens only.  MIND-large has 161 013 news.
    embedding_cache_rows: int = int(os.environ.get("MANNER_EMBED_CACHE_ROWS", "0"))

    def __getstate__(self):                      # the HIP handle is rebuilt lazily after copy/unpickle
        d = self.__dict__.copy()
        d["_hip"], d["_hip_key"] = None, None
        d["_cache"] = None
        d.pop("_param_view", None)
        d.pop("_hip_modes", None)
        d.pop("_cache_mode", None)
        d.pop("_fp", None)
        d["_hip_prefix"], d["_hip_prefix_key"] = None, None
        d["_prefix_cache"], d["_prefix_cache_key"] = None, None
        return d

    def _plm_params(self) -> dict:
        """``dict(self.plm_model.named_parameters())`` through the cached, identity-checked view (``_ParamView``)."""
        if os.environ.get("MANNER_PARAM_VIEW") == "0":               # A/B: the plain tree walk
            return dict(self.plm_model.named_parameters())
        view = self.__dict__.get("_param_view")
        if view is None or view.root is not self.plm_model:
            view = self.__dict__["_param_view"] = _ParamView(self.plm_model)
        return view.get()

    def _encoder(self, device: torch.device, precision: Optional[str] = None) -> hip.HipEncoder:
        params = self._plm_params()
        view = self.__dict__.get("_param_view")
        state = view.state(params) if view is not None else None
        if state is None:
            state = tuple((p.data_ptr(), p._version) for p in params.values())
        # the handle packs the weights once per mode it has been asked for: a caller that alternates autocast states (a 16-mixed fit
        # whose sanity check ran outside autocast, an A/B) grows the set instead of rebuilding the engine at every flip
        mode = precision if precision is not None else self.resolved_precision()
        key = (device, state, _OPT_STEPS[0])
        packed = self.__dict__.get("_hip_modes", ())
        if self._hip is None or self._hip_key != key or mode not in packed:
            same_weights = self._hip is not None and self._hip_key == key
            if self._hip is not None:
                try:
                    self._hip.status()           # input-validation flags still pending on the old handle surface here, not never
                finally:
                    self._hip.close()
                    self._hip = None
            if getattr(self, "_cache", None) is not None and not same_weights:    # other weights: every cached embedding is stale
                self._cache.clear()
            precisions = tuple(dict.fromkeys((("bf16", "fp32") if not same_weights else tuple(packed)) + (mode,)))
            self._hip = hip.HipEncoder(self.plm_model.cfg, {k: v.detach() for k, v in params.items()},
                                       precisions=precisions, device=device)
            self._hip_key = key
            self.__dict__["_hip_modes"] = precisions
            if not same_weights or self.__dict__.get("_fp") is None:
                self.__dict__["_fp"] = (hip.WeightFingerprint(list(params.values()), device)
                                        if self.weight_fingerprint != "0" else None)
        return self._hip

    #: How the inference handle notices parameter writes that torch's version counters do not see (``p.data.mul_(2)``, an EMA swap
    #: through ``p.data.copy_``; everything autograd-visible — optimiser steps, ``load_state_dict``, ``.to()`` — is caught by the
    #: (storage address, version) key at once).  "async" (default): a sampled device-side fingerprint of the parameters is enqueued
    #: after every eval() forward and examined, without blocking, on the way into the next ones — a bulk rewrite rebuilds the handle
    #: and RAISES at the forward after the stale one (the call after next at the latest), like the input-validation flags: never
    #: silent, no host synchronisation.  "sync": the fingerprint is read back BEFORE every forward (one small launch + a blocking
    #: 1 KB copy): the very next forward already computes with the new values.  "0": off.  After a raw ``.data`` write the exact,
    #: free tool is ``invalidate()``.  Env: MANNER_HIP_WEIGHT_FINGERPRINT.
    weight_fingerprint: str = os.environ.get("MANNER_HIP_WEIGHT_FINGERPRINT", "async")

    def invalidate(self) -> None:
        """Forget every copy made of the parameters: the inference handle's packed weights (rebuilt by the next eval() forward), the
        embedding / frozen-prefix caches computed with them and the training path's 16-bit copies of frozen weights
        (``train.invalidate_weight_cache``).  Call it after writing parameters through ``.data`` (or any other route that bypasses
        autograd's version counters)."""
        if self._hip is not None:
            try:
                self._hip.status()
            finally:
                self._hip.close()
                self._hip, self._hip_key = None, None
        self.__dict__.pop("_hip_modes", None)
        self.__dict__.pop("_fp", None)
        if getattr(self, "_cache", None) is not None:
            self._cache.clear()
        if getattr(self, "_hip_prefix", None) is not None:
            self._hip_prefix.close()
            self._hip_prefix, self._hip_prefix_key = None, None
        if getattr(self, "_prefix_cache", None) is not None:
            self._prefix_cache.clear()
        train.invalidate_weight_cache()

    #: GEMM arithmetic of the training path: "f16" / "bf16" ("16-mixed": 16-bit GEMM operands, f32 accumulation, f32
    #: activations and gradients) or "fp32"
    #: "bf16" has f32's exponent range, so activation gradients need no loss scaling.  "f16" is the arithmetic of the
    #: reference's `precision: 16-mixed` and, exactly as there, needs the caller's GradScaler (Lightning's 16-mixed plugin
    #: scales the loss before backward() and unscales .grad afterwards — the engine then sees scaled gradients): token-level
    #: gradients of a fine-tuning step routinely fall below f16's normal range (6e-5) and would flush to zero unscaled.
    #: None (the default) = follow the caller's autocast state, as for ``precision``: fp16 autocast (`16-mixed`) -> "f16" — with the
    #: caller's GradScaler present exactly as in the reference —, bf16 autocast (`bf16-mixed`) -> "bf16", no autocast
    #: (`trainer.precision=32`) -> "fp32".  A string (or MANNER_HIP_TRAIN_PRECISION for a whole process) pins one mode.
    train_precision: Optional[str] = os.environ.get("MANNER_HIP_TRAIN_PRECISION") or None

    def _prefix_encoder(self, device: torch.device, params) -> hip.HipEncoder:
        """Inference engine for the frozen prefix of the training path: rebuilt only when a FROZEN tensor changes (the
        trainable layers' packed copies go stale after every optimiser step, but encode_hidden never reaches them)."""
        tp = self.resolved_train_precision()
        key = (str(device), tp,
               tuple((p.data_ptr(), p._version) for p in params.values() if not p.requires_grad))
        if getattr(self, "_hip_prefix", None) is None or self._hip_prefix_key != key:
            if getattr(self, "_hip_prefix", None) is not None:
                self._hip_prefix.close()
            prec = tp if tp in ("f16", "bf16") else "fp32"
            self._hip_prefix = hip.HipEncoder(self.plm_model.cfg, {k: v.detach() for k, v in self.plm_model.named_parameters()},
                                              precisions=(prec,), device=device)
            self._hip_prefix_key = key
        return self._hip_prefix

    def _forward_train(self, ids: torch.Tensor, mask: torch.Tensor, dropout: bool = True) -> torch.Tensor:
        """train() mode (reference news_encoder.py:29-37 under model.train()): HF's dropouts, the [CLS] dropout and autograd
        into every parameter with requires_grad — through the frozen layers into the embeddings when those train (the
        reference's default); with the embeddings frozen too, the frozen prefix runs once on the inference engine.
        ``dropout=False``: the same differentiable engine with every dropout off (eval() with grad mode on)."""
        plm = self.plm_model
        tp = self.resolved_train_precision()
        params = {k: v for k, v in self._plm_params().items() if not k.startswith("pooler.")}
        emb_frozen = not any(p.requires_grad for k, p in params.items() if k.startswith("embeddings."))
        first_frozen = not any(p.requires_grad for k, p in params.items() if "layer.0." in k)
        engine = self._prefix_encoder(ids.device, params) if (emb_frozen and first_frozen) else None
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())          # torch's CPU generator: reproducible under manual_seed
        on = 1.0 if dropout else 0.0
        extra = {}
        if engine is not None and self.prefix_cache_rows > 0:
            # frozen prefix from the content-addressed table (constant across epochs): only unseen news run layers 0..k-1
            start = train.first_trainable_layer(plm.cfg, params)
            if 0 < start:
                pc = getattr(self, "_prefix_cache", None)
                want = (self.prefix_cache_rows, self.prefix_cache_len, str(ids.device), self._hip_prefix_key)
                if pc is None or self._prefix_cache_key != want:
                    if pc is not None and (pc.capacity, pc.max_len, str(pc.device)) == want[:3]:
                        pc.clear()                                   # same table, other frozen weights / precision
                    else:
                        pc = self._prefix_cache = hip.PrefixCache(plm.cfg.hidden, self.prefix_cache_len, self.prefix_cache_rows, ids.device)
                    self._prefix_cache_key = want
                with torch.no_grad():
                    ph = pc.hidden_states(engine, ids, mask, start, tp)
                extra = dict(prefix_hidden=ph, start_layer=start)
        return train.encode_train(plm.cfg, params, ids, mask, precision=tp, p_hidden=on * plm.hidden_dropout_prob,
                                  p_attn=on * plm.attention_probs_dropout_prob, p_out=on * self.dropout.p, seed=seed, prefix_engine=engine,
                                  **extra)

    #: Opt-in: rows of a `hip.PrefixCache` — the hidden states after the frozen layers, kept per news across steps and epochs when the
    #: embeddings and a prefix of layers are frozen.  One row is prefix_cache_len x hidden f32 (96 x 768: 295 KB).
    prefix_cache_rows: int = int(os.environ.get("MANNER_PREFIX_CACHE_ROWS", "0"))
    prefix_cache_len: int = int(os.environ.get("MANNER_PREFIX_CACHE_LEN", "96"))       # tokenizer_max_length, configs/data/mind_rec.yaml:41

    def forward(self, tokenized_text) -> torch.Tensor:
        ids, mask = tokenized_text["input_ids"], tokenized_text["attention_mask"]
        if not ids.is_cuda:
            raise RuntimeError("MannerTextEncoder.forward needs GPU tensors — the HIP hot path has no CPU fallback")
        if "token_type_ids" in tokenized_text and tokenized_text["token_type_ids"] is not None:
            if bool(torch.count_nonzero(tokenized_text["token_type_ids"])):
                raise ValueError("non-zero token_type_ids: the reference collate never passes them "
                                 "(mind_rec_dataset.py:134-137) and the HIP encoder assumes segment 0")
        # CLS slice of the last hidden state; dropout is the identity in eval().
        # Input validation (bad mask, id outside the vocabulary: ValueError / IndexError in the reference) happens on
        # the device; its flag word is snapshotted behind an event after every call and the completed snapshots are
        # examined before the next one — an invalid batch raises at the following forward (or at check_inputs()),
        # never passes silently, and the fast path has no host synchronisation.
        # Which engine: train() draws dropout masks (with or without a graph), and a graph is recorded whenever autograd
        # would record one for the reference (_wants_graph); only eval() without a graph runs the inference engine.
        if self.training or _wants_graph(self.plm_model):
            if not self.training:
                _warn_eval_graph("MannerTextEncoder")
            return self._forward_train(ids, mask, dropout=self.training)
        mode = self.resolved_precision()
        enc = self._encoder(ids.device, mode)
        fp = self.__dict__.get("_fp")
        if fp is not None:
            if self.weight_fingerprint == "sync":
                if fp.changed_now():
                    self.invalidate()
                    enc = self._encoder(ids.device, mode)
                    fp = None
            elif fp.changed():
                self.invalidate()
                raise RuntimeError("MannerTextEncoder: the PLM parameters were rewritten behind autograd's version counters (a write "
                                   "through p.data?) — up to two earlier eval() forwards computed with the old values.  The handle has "
                                   "been rebuilt (a retry runs on the new values); call .invalidate() right after such a write, or set "
                                   "MANNER_HIP_WEIGHT_FINGERPRINT=sync")
        enc.status_poll()
        if self.embedding_cache_rows > 0:
            out = self._forward_cached(enc, ids, mask, mode)
        else:
            out = enc.encode_cls(ids, mask, precision=mode)
        enc.status_arm()
        if fp is not None and self.weight_fingerprint != "sync":
            fp.arm()
        return out

    def _forward_cached(self, enc: hip.HipEncoder, ids: torch.Tensor, mask: torch.Tensor, mode: str) -> torch.Tensor:
        """eval() forward through the content-addressed cache: encode the rows not seen before, return every row from the table.
        One host read per call (how many rows are new) — the unchanged callers synchronise anyway (`to_dense_batch`, the
        per-row `torch.where` loop of cr_module.py:117-120)."""
        cache = getattr(self, "_cache", None)
        if cache is None or cache.capacity != self.embedding_cache_rows or cache.device != ids.device or cache.dim != self.plm_model.cfg.hidden:
            cache = self._cache = hip.NewsEmbeddingCache(self.plm_model.cfg.hidden, self.embedding_cache_rows, ids.device)
        if self.__dict__.get("_cache_mode") != mode:          # rows computed in another arithmetic are not this mode's rows
            if self.__dict__.get("_cache_mode") is not None:
                cache.clear()
            self.__dict__["_cache_mode"] = mode
        try:
            rows, state = cache.lookup(ids, mask)
            todo = torch.nonzero(state != 0).squeeze(1)                          # the host read
            cache.lookups += int(ids.shape[0])
            cache.encoded += int(todo.numel())
            out = None
            if todo.numel():
                fresh = enc.encode_cls(ids.index_select(0, todo), mask.index_select(0, todo), precision=mode)
                keep = state.index_select(0, todo) == 1
                cache.table.index_copy_(0, rows.index_select(0, todo)[keep].long(), fresh[keep])
                if todo.numel() == ids.shape[0]:
                    return fresh
                out = cache.table.index_select(0, rows.clamp_min(0).long())
                out.index_copy_(0, todo, fresh)                                  # state 2 rows have no table row
                return out
            return cache.table.index_select(0, rows.long())
        except BaseException:                  # KeyboardInterrupt / SystemExit too: a notebook or a resumed loop goes on using the module
            cache.clear()                      # keys of this call may point at rows that were never written
            raise

    def warm_embedding_cache(self, tokenized_text, chunk: int = 8192) -> int:
        """Encode a whole news pool into the embedding cache in large calls (mode T's table build: the engine's full rate instead
        of the latency floor of the small calls that misses inside a batch cost) — e.g. from a Lightning ``on_test_start`` hook with
        the tokenised news of the dev set; every later eval() forward then finds its rows.  Needs ``embedding_cache_rows`` > 0 and
        eval() mode; returns the number of news that were encoded."""
        if self.embedding_cache_rows <= 0:
            raise RuntimeError("warm_embedding_cache: set embedding_cache_rows first")
        if self.training:
            raise RuntimeError("warm_embedding_cache: eval() mode only (train() draws dropout masks and records a graph)")
        ids, mask = tokenized_text["input_ids"], tokenized_text["attention_mask"]
        encoded = 0
        with torch.no_grad():
            for a in range(0, ids.shape[0], chunk):
                cache = getattr(self, "_cache", None)
                e0 = cache.encoded if cache is not None else 0
                self.forward({"input_ids": ids[a:a + chunk], "attention_mask": mask[a:a + chunk]})
                e1 = self._cache.encoded
                encoded += e1 - e0 if (self._cache is cache and e1 >= e0) else e1      # the first call may create or empty the table
        return encoded

    def check_inputs(self) -> None:
        """Blocking: raise if any forward so far saw an invalid attention_mask / input id."""
        if self._hip is not None:
            self._hip.status()


    def frozen_hidden_states(self, tokenized_text, n_layers: Optional[int] = None,
                             dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
        """HF ``hidden_states[n_layers]`` [N, Lp, H] computed by the HIP encoder without autograd — by default up to
        the first trainable layer (the constructor's ``frozen_layers`` must then be a prefix 0..k-1, as in
        configs/model/cr_module.yaml:10).  These activations are constant across epochs, so a training loop can cache
        them per news and run only layers k.. in PyTorch with gradients.  Note that the
        reference's name test freezes the encoder layers only — its embedding tables keep training — so a cached
        prefix reproduces the reference exactly only when the embeddings are frozen as well."""
        if n_layers is None:
            frozen = sorted({int(n.split("layer.")[1].split(".")[0]) for n, p in self.plm_model.named_parameters()
                             if "layer." in n and not p.requires_grad})
            if frozen != list(range(len(frozen))):
                raise ValueError(f"frozen layers {frozen} are not a prefix 0..k-1; pass n_layers explicitly")
            n_layers = len(frozen)
        ids, mask = tokenized_text["input_ids"], tokenized_text["attention_mask"]
        if not ids.is_cuda:
            raise RuntimeError("MannerTextEncoder.frozen_hidden_states needs GPU tensors — no CPU fallback")
        with torch.no_grad():
            mode = self.resolved_precision()
            return self._encoder(ids.device, mode).encode_hidden(ids, mask, n_layers, precision=mode, out_dtype=dtype)


class MannerEntityEncoder(nn.Module):
    """reference news_encoder.py:40-72.

    BATCH-FAITHFUL to the reference: its nn.MultiheadAttention is batch_first=False but receives
    [N, E, D], so entity attention runs across the N news of the call at each entity slot and there is
    no key_padding_mask.  The HIP path reproduces exactly that, so — as in the
    reference — a news embedding depends on the other news of the batch when use_entities=True."""

    def __init__(self, pretrained_embedding: nn.Embedding, embedding_dim: int, num_attention_heads: int,
                 query_vector_dim: int, dropout_probability: float) -> None:
        super().__init__()
        self.pretrained_embedding = pretrained_embedding
        self.multihead_attention = nn.MultiheadAttention(embed_dim=embedding_dim, num_heads=num_attention_heads)
        self.additive_attention = AdditiveAttention(input_dim=embedding_dim, query_dim=query_vector_dim)
        self.dropout = nn.Dropout(p=dropout_probability)

    def forward(self, entity_sequence: torch.Tensor) -> torch.Tensor:
        mha, pool = self.multihead_attention, self.additive_attention
        if self.training or _wants_graph(self):
            # train() mode (news_encoder.py:60-72): embedding -> dropout -> axis-0 attention -> dropout -> additive pooler,
            # every operator with its hand-written backward (manner_amd.train / csrc/train_small.hip); eval() with a graph:
            # the same operators with the dropouts off
            if mha.dropout != 0.0:
                raise RuntimeError("attention-probability dropout inside nn.MultiheadAttention is not built (the reference uses 0)")
            p = self.dropout.p if self.training else 0.0
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0
            emb = self.pretrained_embedding
            x = train.embedding(entity_sequence, emb.weight, emb.padding_idx)
            x = train.dropout(x, p, seed, site=2)
            x = train.mha_axis0(x, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias, mha.num_heads)
            x = train.dropout(x, p, seed, site=3)
            return train.additive_pool(x, pool.linear.weight, pool.linear.bias, pool.query)
        hip.status_poll(entity_sequence.device)          # an entity index outside the table (IndexError in the reference) of an earlier call
        out = hip.entity_encode(entity_sequence, self.pretrained_embedding.weight.detach(), mha.in_proj_weight.detach(),
                                mha.in_proj_bias.detach(), mha.out_proj.weight.detach(), mha.out_proj.bias.detach(),
                                pool.linear.weight.detach(), pool.linear.bias.detach(), pool.query.detach(),
                                heads=mha.num_heads)
        hip.status_arm(entity_sequence.device)
        return out


class MannerNewsEncoder(nn.Module):
    """reference news_encoder.py:75-129."""

    def __init__(self, plm_model: str, frozen_layers: List[int], dropout_probability: float, use_entities: bool,
                 entity_embeddings: torch.Tensor, entity_embedding_dim: int, num_attention_heads: int,
                 query_vector_dim: int, text_embedding_dim: int) -> None:
        super().__init__()
        self.text_encoder = MannerTextEncoder(plm_model=plm_model, frozen_layers=frozen_layers,
                                              dropout_probability=dropout_probability)
        self.use_entities = use_entities
        if self.use_entities:
            pretrained_entity_embedding = nn.Embedding.from_pretrained(
                embeddings=torch.FloatTensor(entity_embeddings), freeze=False, padding_idx=0)
            self.entity_encoder = MannerEntityEncoder(
                pretrained_embedding=pretrained_entity_embedding, embedding_dim=entity_embedding_dim,
                num_attention_heads=num_attention_heads, query_vector_dim=query_vector_dim,
                dropout_probability=dropout_probability)
            self.linear = nn.Linear(in_features=text_embedding_dim + entity_embedding_dim,
                                    out_features=text_embedding_dim)

    def forward(self, news: Dict[str, Any]) -> torch.Tensor:
        # text embedding
        text_vector = self.text_encoder(news["text"])
        if not self.use_entities:
            return text_vector
        # entity embedding, concat, linear (news_encoder.py:119-124)
        entity_vector = self.entity_encoder(news["entities"])
        both = torch.cat([text_vector, entity_vector], dim=-1)                  # a copy; its backward is a split
        if _wants_graph(self.linear, both):
            return train.linear(both, self.linear.weight, self.linear.bias)
        return hip.linear(both, self.linear.weight.detach(), self.linear.bias.detach())


class PLMTextEncoder(nn.Module):
    """reference news_encoder.py:132-171 — the text encoder of the PLM baselines (NRMS-PLM, TANR-PLM, SentiRec-PLM, ...).

    Batch-faithful to the reference: its nn.MultiheadAttention is batch_first=False but receives [B, S, D], and neither it
    nor the additive pooler gets a mask, so (a) attention runs ACROSS THE NEWS OF THE CALL at each token position and (b)
    the hidden states AT PADDED POSITIONS take part in both — ``hip.encode_full`` (and, with autograd, ``train.encode_full_train``)
    therefore computes them as HF does."""

    #: GEMM arithmetic of the PLM in this class.  None (default) = the caller's autocast state, as MannerTextEncoder: fp16 autocast ->
    #: "f16", bf16 autocast -> "bf16", none -> "fp32" (these are baselines, not the throughput path); a string pins one mode
    precision: Optional[str] = None

    def __init__(self, plm_model: str, frozen_layers: List[int], text_embedding_dim: int, num_attention_heads: int,
                 query_vector_dim: int, dropout_probability: float) -> None:
        super().__init__()
        self.plm_model = HipPLM.from_pretrained(plm_model)
        self.multihead_attention = nn.MultiheadAttention(embed_dim=text_embedding_dim, num_heads=num_attention_heads)
        self.additive_attention = AdditiveAttention(input_dim=text_embedding_dim, query_dim=query_vector_dim)
        self.dropout = nn.Dropout(p=dropout_probability)
        for name, param in self.plm_model.base_model.named_parameters():
            for layer in frozen_layers:
                if "layer." + str(layer) + "." in name:
                    param.requires_grad = False

    #: GEMM arithmetic of the PLM in train() mode ("fp32", "bf16" or "f16"; None = the caller's autocast state, "fp32" outside it: see
    #: MannerTextEncoder.train_precision)
    train_precision: Optional[str] = None

    def forward(self, tokenized_text) -> torch.Tensor:
        ids, mask = tokenized_text["input_ids"], tokenized_text["attention_mask"]
        if not ids.is_cuda:
            raise RuntimeError("PLMTextEncoder.forward needs GPU tensors — the HIP path has no CPU fallback")
        if self.training or _wants_graph(self):
            # train() mode (news_encoder.py:160-171; trained by baselines/nrms_plm_module.py:119-135): the PLM with HF's dropouts and
            # autograd over "full rows" (padded positions included), dropout, the axis-0 attention, dropout, the pooler — every
            # operator with its hand-written backward; eval() with a graph: the same operators, dropouts off
            mha, pool = self.multihead_attention, self.additive_attention
            if mha.dropout != 0.0:
                raise RuntimeError("attention-probability dropout inside nn.MultiheadAttention is not built (the reference uses 0)")
            if not self.training:
                _warn_eval_graph("PLMTextEncoder")
            on = 1.0 if self.training else 0.0
            plm = self.plm_model
            params = {k: v for k, v in plm.named_parameters() if not k.startswith("pooler.")}
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            x = train.encode_full_train(plm.cfg, params, ids, mask, precision=self.train_precision or autocast_mode() or "fp32",
                                        p_hidden=on * plm.hidden_dropout_prob,
                                        p_attn=on * plm.attention_probs_dropout_prob, seed=seed)
            x = train.dropout(x, on * self.dropout.p, seed, site=4)
            x = train.mha_axis0(x, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias, mha.num_heads)
            x = train.dropout(x, on * self.dropout.p, seed, site=5)
            return train.additive_pool(x, pool.linear.weight, pool.linear.bias, pool.query)
        params = {k: v.detach() for k, v in self.plm_model.named_parameters() if not k.startswith("pooler.")}
        hidden = hip.encode_full(self.plm_model.cfg, params, ids, mask, precision=self.precision or autocast_mode() or "fp32")   # [B, S, D], pads incl.
        mha, pool = self.multihead_attention, self.additive_attention
        mixed = hip.mha_axis0(hidden, mha.in_proj_weight.detach(), mha.in_proj_bias.detach(), mha.out_proj.weight.detach(),
                              mha.out_proj.bias.detach(), mha.num_heads)
        return hip.additive_pool(mixed, pool.linear.weight.detach(), pool.linear.bias.detach(), pool.query.detach())
