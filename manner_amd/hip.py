"""Torch-tensor front end of the C ABI: pointer/stream plumbing only.

PyTorch-ROCm provides device memory and the current HIP stream; all arithmetic happens in
libmanner_hip.so.  Every function raises on non-GPU tensors — there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .config import EncoderConfig
from .weights import canonical_weights, plm_param_shapes

Tensor = torch.Tensor


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: Tensor, dtype: torch.dtype, name: str) -> Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor — the MANNeR HIP hot path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _ptr(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


# ------------------------------------------------------------------ device-side input validation
_STATUS_TEXT = {_lib.STATUS_MASK: "attention_mask is not a right-padded 0/1 prefix mask",
                _lib.STATUS_TOKEN: "input id / position outside the embedding tables",
                _lib.STATUS_INDEX: "news or entity index outside the table (IndexError in the reference)",
                _lib.STATUS_LENGTHS: "host_lengths disagree with attention_mask"}


def _status_message(flag: int) -> str:
    return "; ".join(t for b, t in _STATUS_TEXT.items() if flag & b) or f"device status 0x{flag:x}"


class _FlagRing:
    """Snapshots of a device status word in a small ring of pinned host slots, each behind an event: ``completed()``
    returns the OR of the snapshots whose copy has finished, without ever blocking."""

    SLOTS = 32

    def __init__(self):
        self._host = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
        self._pending = []                     # (slot, event), oldest first
        self._next = 0

    def slot(self):
        """-> (flags carried over from a recycled slot, pinned one-element view to copy the word into)."""
        carried = 0
        i = self._next
        self._next = (i + 1) % self.SLOTS
        for j, (sl, ev) in enumerate(self._pending):
            if sl == i:                        # the slot comes round again before its snapshot was consumed: take it now
                ev.synchronize()
                carried = int(self._host[sl])
                del self._pending[j]
                break
        return carried, i, self._host[i:i + 1]

    def push(self, i: int) -> None:
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((i, ev))

    def completed(self, wait: bool = False) -> int:
        """OR of the finished snapshots.  ``wait``: block for all of them.  Otherwise every snapshot EXCEPT the newest is waited for
        (those copies were enqueued at least one call earlier — behind work the device has long finished or is about to —
        so a raised bit surfaces at the call after next at the latest instead of "whenever the event happens to have completed");
        the newest is taken only if it has already completed, so the host still runs one call ahead of the device."""
        flag, keep = 0, []
        last = len(self._pending) - 1
        for j, (i, ev) in enumerate(self._pending):
            if wait or j < last:
                ev.synchronize()
            if wait or j < last or ev.query():
                flag |= int(self._host[i])
            else:
                keep.append((i, ev))
        self._pending = keep
        return flag


class DeviceStatus:
    """The int32 status word the scoring kernels raise bits in (``MANNER_HIP_STATUS_*``) where the reference would
    raise an exception (an out-of-range table index).  ``check()`` is blocking; ``poll()`` never is: every ``arm()``
    enqueues a copy of the word into pinned host memory and an event, and ``poll()`` raises for the snapshots that have
    completed, waiting for every one but the newest — so a bad batch surfaces at the next call when the device has kept up,
    at the call after next at the latest, and the host never waits for the call it has just enqueued."""

    def __init__(self, device: torch.device):
        self.device = device
        self.word = torch.zeros(1, dtype=torch.int32, device=device)
        self._ring = _FlagRing()
        self._carry = 0

    def arm(self) -> None:
        with torch.cuda.device(self.device):
            carried, i, host = self._ring.slot()
            self._carry |= carried
            host.copy_(self.word, non_blocking=True)
            self.word.zero_()
            self._ring.push(i)

    def _raise(self, flag: int) -> None:
        flag |= self._carry
        self._carry = 0
        if flag:
            raise RuntimeError(f"manner_hip input error: {_status_message(flag)}")

    def poll(self) -> None:
        self._raise(self._ring.completed())

    def check(self) -> None:
        flag = int(self.word.item())           # synchronises with the kernels that may still raise bits
        self.word.zero_()
        self._raise(flag | self._ring.completed(wait=True))


_status: Dict[str, DeviceStatus] = {}


def device_status(device) -> DeviceStatus:
    key = str(torch.device(device))
    if key not in _status:
        _status[key] = DeviceStatus(torch.device(device))
    return _status[key]


def check_status(device=None) -> None:
    """Blocking: raise if any scoring / entity kernel on ``device`` (default: every device used so far) saw an
    out-of-range index since the last check."""
    for st in ([device_status(device)] if device is not None else list(_status.values())):
        st.check()


def status_poll(device) -> None:
    """Non-blocking: raise for every armed snapshot of ``device``'s status word whose copy has completed.  The composed
    entry points (hotpath.*, train.encode_train, the module mirrors) call this on the way in and ``status_arm`` on the
    way out, so an out-of-range index / a token_bound below the mask's token count surfaces at the next call when the device
    has kept up and at the call AFTER next at the latest (every snapshot but the newest is waited for) — the host still runs one
    call ahead of the device; ``check_status`` is the blocking form, to be called when a loop ends (epoch end)."""
    device_status(device).poll()


def status_arm(device) -> None:
    device_status(device).arm()


def weight_table_order(cfg: EncoderConfig) -> Sequence[str]:
    """HF parameter names in the order of the ``weights`` table of manner_hip_encoder_create."""
    names = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
             "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]
    per_layer = ["attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight",
                 "attention.self.key.bias", "attention.self.value.weight", "attention.self.value.bias",
                 "attention.output.dense.weight", "attention.output.dense.bias",
                 "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
                 "intermediate.dense.weight", "intermediate.dense.bias", "output.dense.weight",
                 "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias"]
    for l in range(cfg.layers):
        names += [f"encoder.layer.{l}.{n}" for n in per_layer]
    return names


class HipEncoder:
    """Owns a manner_hip_encoder handle (packed PLM weights) and a growable workspace."""

    def __init__(self, cfg: EncoderConfig, weights: Dict[str, Tensor], precisions: Sequence[str] = ("bf16", "fp32"),
                 device: Optional[torch.device] = None):
        lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise RuntimeError("HipEncoder needs a GPU device — there is no CPU fallback")
        weights = canonical_weights(cfg, weights)           # DistilBERT names -> BERT names (+ zero token-type row)
        shapes = dict(plm_param_shapes(cfg if cfg.naming == "bert" else EncoderConfig(**{**cfg.to_dict(), "naming": "bert"})))
        names = weight_table_order(cfg)
        keep = []
        with torch.cuda.device(self.device):
            for n in names:
                w = weights[n]
                if isinstance(w, np.ndarray):
                    w = torch.from_numpy(w)
                w = w.detach().to(device=self.device, dtype=torch.float32).contiguous()
                if tuple(w.shape) != tuple(shapes[n]):
                    raise ValueError(f"{n}: shape {tuple(w.shape)} != {shapes[n]}")
                keep.append(w)
            table = (C.c_void_p * len(keep))(*[w.data_ptr() for w in keep])
            cc = _lib.EncoderConfigC(cfg.arch, cfg.hidden, cfg.layers, cfg.heads, cfg.intermediate, cfg.vocab,
                                     cfg.max_pos, cfg.type_vocab, cfg.pad_id, cfg.ln_eps)
            mask = 0
            for p in precisions:
                mask |= 1 << _lib.PRECISIONS[p]
            handle = C.c_void_p()
            _lib.check(lib.manner_hip_encoder_create(C.byref(cc), table, len(keep), mask, _stream(), C.byref(handle)))
        self._handle = handle
        self._ws: Optional[Tensor] = None
        self._ring: Optional[_FlagRing] = None
        self._carry = 0
        del keep                      # the handle owns private packed copies

    def close(self) -> None:
        if getattr(self, "_handle", None):
            _lib.load().manner_hip_encoder_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _workspace(self, n_news: int, tokens: int, prec: int) -> Tuple[Tensor, int]:
        """(buffer, bytes to hand to the library).  The library sizes its chunks from the BYTES it is given, so a buffer that an
        earlier call in a wider arithmetic left larger must not change this call's chunking (and with it the per-launch figures)."""
        need = int(_lib.load().manner_hip_encoder_workspace_bytes(self._handle, n_news, tokens, prec))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    def encode_cls(self, ids: Tensor, mask: Tensor, precision: str = "bf16",
                   host_lengths: Optional[np.ndarray] = None, max_chunk_tokens: int = 65536,
                   out: Optional[Tensor] = None) -> Tensor:
        """[N, Lp] int64 ids/mask -> [N, H] float32 CLS embeddings."""
        ids, mask = _dev(ids, torch.int64, "input_ids"), _dev(mask, torch.int64, "attention_mask")
        if ids.dim() != 2 or ids.shape != mask.shape:
            raise ValueError(f"input_ids {tuple(ids.shape)} / attention_mask {tuple(mask.shape)} must be equal 2-D")
        ids, mask = ids.contiguous(), mask.contiguous()
        n, lp = ids.shape
        prec = _lib.PRECISIONS[precision]
        if out is None:
            out = torch.empty((n, self.cfg.hidden), dtype=torch.float32, device=ids.device)
        else:
            _dev(out, torch.float32, "out")
            assert out.is_contiguous() and tuple(out.shape) == (n, self.cfg.hidden)
        if n == 0:
            return out
        hl = None
        if host_lengths is not None:
            hl = np.ascontiguousarray(host_lengths, dtype=np.int32)
            assert hl.shape == (n,)
            tokens = min(int(hl.sum()), max_chunk_tokens)
        else:
            tokens = min(n * lp, max_chunk_tokens)
        tokens = max(tokens, lp, 256)
        with torch.cuda.device(ids.device):
            ws, ws_bytes = self._workspace(min(n, tokens), tokens, prec)
            _lib.check(_lib.load().manner_hip_encode_cls(
                self._handle, _ptr(ids), _ptr(mask), C.c_void_p(hl.ctypes.data if hl is not None else 0), n, lp, prec,
                _ptr(out), _ptr(ws), ws_bytes, _stream()))
        return out

    def encode_hidden(self, ids: Tensor, mask: Tensor, n_layers: int, precision: str = "bf16",
                      out_dtype: torch.dtype = torch.float32, host_lengths: Optional[np.ndarray] = None,
                      max_chunk_tokens: int = 65536) -> Tensor:
        """HF ``hidden_states[n_layers]`` as [N, Lp, H] (zeros at padded positions): the cacheable output of the
        frozen layers (``frozen_layers: [0..7]`` -> ``n_layers=8``) for a training loop that only updates the rest."""
        ids, mask = _dev(ids, torch.int64, "input_ids"), _dev(mask, torch.int64, "attention_mask")
        if ids.dim() != 2 or ids.shape != mask.shape:
            raise ValueError(f"input_ids {tuple(ids.shape)} / attention_mask {tuple(mask.shape)} must be equal 2-D")
        if out_dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("out_dtype must be float32 or bfloat16")
        ids, mask = ids.contiguous(), mask.contiguous()
        n, lp = ids.shape
        prec = _lib.PRECISIONS[precision]
        out = torch.empty((n, lp, self.cfg.hidden), dtype=out_dtype, device=ids.device)
        if n == 0:
            return out
        hl = None
        if host_lengths is not None:
            hl = np.ascontiguousarray(host_lengths, dtype=np.int32)
            assert hl.shape == (n,)
            tokens = min(int(hl.sum()), max_chunk_tokens)
        else:
            tokens = min(n * lp, max_chunk_tokens)
        tokens = max(tokens, lp, 256)
        with torch.cuda.device(ids.device):
            ws, ws_bytes = self._workspace(min(n, tokens), tokens, prec)
            _lib.check(_lib.load().manner_hip_encode_hidden(
                self._handle, _ptr(ids), _ptr(mask), C.c_void_p(hl.ctypes.data if hl is not None else 0), n, lp, prec,
                int(n_layers), 0 if out_dtype == torch.float32 else 1, _ptr(out), _ptr(ws), ws_bytes, _stream()))
        return out

    def profile(self, enable: bool) -> None:
        """Bracket every launch of encode_cls with HIP events on the launch stream (opt-in)."""
        _lib.check(_lib.load().manner_hip_encoder_profile(self._handle, int(enable)))

    def profile_read(self) -> Dict[str, Tuple[float, int]]:
        """{kernel class: (total ms, launches)} accumulated since the last read; synchronises."""
        n = len(_lib.PROF_CLASSES)
        ms, cnt = (C.c_double * n)(), (C.c_int64 * n)()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().manner_hip_encoder_profile_read(self._handle, _stream(), ms, cnt))
        return {c: (ms[i], cnt[i]) for i, c in enumerate(_lib.PROF_CLASSES)}

    def status(self) -> None:
        """Blocking check of the device-side input validation flag (raises on bad masks/ids)."""
        carried = 0
        if self._ring is not None:
            carried = self._ring.completed(wait=True) | self._carry
            self._carry = 0
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().manner_hip_encoder_status(self._handle, _stream()))
        if carried:
            raise RuntimeError(f"manner_hip input error in an earlier encode call: {_status_message(carried)}")

    def status_arm(self) -> None:
        """Non-blocking: snapshot (and reset) the flag word into pinned host memory behind an event on the current
        stream; ``status_poll`` raises for completed snapshots.  The module mirror calls both around every forward."""
        if self._ring is None:
            self._ring = _FlagRing()
        with torch.cuda.device(self.device):
            carried, i, host = self._ring.slot()
            self._carry |= carried
            _lib.check(_lib.load().manner_hip_encoder_status_async(self._handle, C.c_void_p(host.data_ptr()), _stream()))
            self._ring.push(i)

    def status_poll(self) -> None:
        flag = self._carry | (self._ring.completed() if self._ring is not None else 0)
        self._carry = 0
        if flag:
            raise RuntimeError(f"manner_hip input error in an earlier encode call: {_status_message(flag)}")


class WeightFingerprint:
    """Tripwire for writes autograd's version counters cannot see (``p.data.mul_(2)``, ``p.data.copy_(ema)``: the ``.data`` alias has a
    version counter of its own) under a host that caches packed copies of the parameters — the module mirror's inference handle.

    One launch of ``manner_hip_fingerprint`` hashes up to ``SAMPLES`` evenly strided words of every tensor into one word per tensor.
    ``baseline`` is taken (blocking) when the packed copies are made; ``arm()`` enqueues a fresh fingerprint and its copy into a pinned
    slot behind an event, ``changed()`` compares the snapshots that have completed — every one but the newest is waited for, as
    ``_FlagRing`` does for the status word — so a bulk rewrite surfaces at the next call when the device has kept up and at the call
    after next at the latest, with no host synchronisation per forward; ``changed_now()`` is the blocking form.  A poke into single
    elements between the samples is not seen: documented limit, ``MannerTextEncoder.invalidate()`` is the exact tool."""

    SAMPLES = 1024
    SLOTS = 8

    def __init__(self, tensors: Sequence[Tensor], device: torch.device):
        self.device = torch.device(device)
        self._keep = [t for t in tensors]
        n = len(self._keep)
        # (normal tensors even when the first forward runs under inference_mode: they are updated in place by later no_grad calls)
        with torch.inference_mode(False), torch.cuda.device(self.device):
            self._ptrs = torch.tensor([t.data_ptr() for t in self._keep], dtype=torch.int64, device=self.device)
            self._counts = torch.tensor([t.numel() * t.element_size() // 4 for t in self._keep], dtype=torch.int64, device=self.device)
            self._out = torch.empty(n, dtype=torch.int32, device=self.device)
            self._host = torch.zeros((self.SLOTS, n), dtype=torch.int32).pin_memory()
        self._pending: list = []
        self._next = 0
        self.baseline = self._blocking()

    def _launch(self) -> None:
        _lib.check(_lib.load().manner_hip_fingerprint(_ptr(self._ptrs), _ptr(self._counts), len(self._keep), self.SAMPLES, _ptr(self._out), _stream()))

    def _blocking(self) -> Tensor:
        with torch.cuda.device(self.device):
            self._launch()
            return self._out.cpu()

    def changed_now(self) -> bool:
        return not torch.equal(self._blocking(), self.baseline)

    def arm(self) -> None:
        with torch.cuda.device(self.device):
            i = self._next
            self._next = (i + 1) % self.SLOTS
            stale = False
            for j, (sl, ev) in enumerate(self._pending):
                if sl == i:                    # the slot comes round before its snapshot was looked at: look now
                    ev.synchronize()
                    stale = not torch.equal(self._host[sl], self.baseline)
                    del self._pending[j]
                    break
            self._carry = getattr(self, "_carry", False) or stale
            self._launch()
            self._host[i].copy_(self._out, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((i, ev))

    def changed(self) -> bool:
        hit, keep = getattr(self, "_carry", False), []
        self._carry = False
        last = len(self._pending) - 1
        for j, (i, ev) in enumerate(self._pending):
            if j < last:
                ev.synchronize()
            if j < last or ev.query():
                hit = hit or not torch.equal(self._host[i], self.baseline)
            else:
                keep.append((i, ev))
        self._pending = keep
        return hit


class NewsEmbeddingCache:
    """Content-addressed table of text-encoder outputs (csrc/cache.hip; SURVEY.md §8d mode T behind the drop-in call pattern).

    ``lookup(ids, mask)`` keys every row by its real tokens and returns (rows int32 [N], state int32 [N]): state 0 — the embedding is
    (or, for a repeat of a key new in this call, will be) at ``table[rows]``; 1 — new, the caller encodes it and stores it at
    ``table[rows]``; 2 — encode, do not store (table full).  Everything lives in HBM: ``capacity`` rows of ``dim`` f32 plus
    2 x capacity hash slots of 20 bytes (161 013 news x 768 f32 = 495 MB — the whole MIND-large table fits 500 times over).
    The owner clears it whenever the weights or the arithmetic mode change."""

    def __init__(self, dim: int, capacity: int, device: torch.device, zero: bool = False):
        if capacity < 1:
            raise ValueError("NewsEmbeddingCache: capacity must be positive")
        self.dim, self.capacity, self.device = int(dim), int(capacity), torch.device(device)
        n_slots = 2
        while n_slots < 2 * self.capacity:
            n_slots *= 2
        self.n_slots = n_slots
        # ordinary tensors even when the first forward runs under torch.inference_mode() (Lightning's test loop): an inference
        # tensor could not be updated in place by a later call under no_grad
        with torch.inference_mode(False):
            self.table = (torch.zeros if zero else torch.empty)((self.capacity, self.dim), dtype=torch.float32, device=self.device)
            self.slot_keys = torch.zeros((2, n_slots), dtype=torch.int64, device=self.device)     # uint64 bit patterns
            self.slot_rows = torch.full((n_slots,), -1, dtype=torch.int32, device=self.device)
            self.row_count = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.lookups = 0          # host-side counters (rows asked for / rows encoded), for the hit rate
        self.encoded = 0

    def clear(self) -> None:
        self.slot_keys.zero_()
        self.slot_rows.fill_(-1)
        self.row_count.zero_()
        self.lookups = self.encoded = 0

    def keys(self, ids: Tensor, mask: Tensor) -> Tensor:
        ids = _dev(ids, torch.int64, "input_ids").contiguous()
        mask = _dev(mask, torch.int64, "attention_mask").contiguous()
        if ids.dim() != 2 or ids.shape != mask.shape:
            raise ValueError("NewsEmbeddingCache: input_ids / attention_mask must be [n_news, padded_len]")
        out = torch.empty((ids.shape[0], 2), dtype=torch.int64, device=ids.device)
        with torch.cuda.device(ids.device):
            _lib.check(_lib.load().manner_hip_news_key128(_ptr(ids), _ptr(mask), ids.shape[0], ids.shape[1], _ptr(out), _stream()))
        return out

    def lookup(self, ids: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
        keys = self.keys(ids, mask)
        n = keys.shape[0]
        rows = torch.empty(n, dtype=torch.int32, device=keys.device)
        state = torch.empty(n, dtype=torch.int32, device=keys.device)
        scratch = torch.empty(2 * max(n, 1), dtype=torch.int32, device=keys.device)
        with torch.cuda.device(keys.device):
            _lib.check(_lib.load().manner_hip_news_cache_lookup(_ptr(keys), n, _ptr(self.slot_keys), _ptr(self.slot_rows), self.n_slots,
                                                                 _ptr(self.row_count), self.capacity, _ptr(rows), _ptr(state), _ptr(scratch),
                                                                 _stream()))
        return rows, state


class PrefixCache(NewsEmbeddingCache):
    """The same table with one row = the hidden states of a news after the frozen layers, [max_len, hidden] f32, zeros at padded
    positions ("activations after layer 7 are constant across epochs").  65 238 news x 96 tokens x 768 f32 =
    19.2 GB, 161 013 news = 47.5 GB: HBM the MI355X has to spare."""

    def __init__(self, hidden: int, max_len: int, capacity: int, device: torch.device):
        super().__init__(int(hidden) * int(max_len), capacity, device, zero=True)
        self.hidden, self.max_len = int(hidden), int(max_len)

    def clear(self) -> None:
        super().clear()
        self.table.zero_()                  # a row stored at a narrower padded width must read zeros beyond it

    def hidden_states(self, engine: "HipEncoder", ids: Tensor, mask: Tensor, n_layers: int, precision: str) -> Tensor:
        """``engine.encode_hidden(ids, mask, n_layers)`` [N, Lp, H] f32 with every row whose tokens were seen before taken from the table
        (bit-identical: a row's hidden states do not depend on the other rows of the call).  Wider batches than ``max_len`` bypass it."""
        n, lp = ids.shape
        if lp > self.max_len or n == 0:
            return engine.encode_hidden(ids, mask, n_layers, precision=precision)
        try:
            rows, state = self.lookup(ids, mask)
            todo = torch.nonzero(state != 0).squeeze(1)                          # one host read: how many rows are new
            self.lookups += int(n)
            self.encoded += int(todo.numel())
            tbl = self.table.view(self.capacity, self.max_len, self.hidden)
            fresh = None
            if todo.numel():
                fresh = engine.encode_hidden(ids.index_select(0, todo), mask.index_select(0, todo), n_layers, precision=precision)
                if todo.numel() == n and not bool((state == 1).any()):
                    return fresh
                keep = state.index_select(0, todo) == 1
                tbl[rows.index_select(0, todo)[keep].long(), :lp] = fresh[keep]
                if todo.numel() == n:
                    return fresh
            out = tbl[rows.clamp_min(0).long(), :lp]                             # gather: a new [N, Lp, H] tensor
            if fresh is not None:
                out.index_copy_(0, todo, fresh)                                  # rows that could not be stored
            return out
        except BaseException:                  # also KeyboardInterrupt / SystemExit: the table is torch.empty, a claimed key must not survive
            self.clear()
            raise


def additive_pool(x: Tensor, lin_w: Tensor, lin_b: Tensor, query: Tensor, strict: bool = False) -> Tensor:
    """AdditiveAttention.forward (reference attention.py:21-27).  Default: the one-pass kernel where the shape allows it (D = 768,
    S <= 128, Q <= 320) — x read once and kept on the CU as power-of-two-scaled IEEE-half hi/lo pairs, logits as split (x3) products
    on the f16 matrix pipe, tanh by exp/rcp: pooled vectors within 1e-4 of the reference (measured 2e-7 on the goldens, 1.1e-5 on a
    peaked-softmax stress input; csrc/pool.hip).  ``strict=True`` (or MANNER_HIP_POOL_STRICT=1): the exact-f32 two-pass path (f32
    matrix pipe logits + apply), which train() mode always uses."""
    x = _dev(x, torch.float32, "input_vector").contiguous()
    b, s, d = x.shape
    lin_w = _dev(lin_w, torch.float32, "linear.weight").contiguous()
    lin_b = _dev(lin_b, torch.float32, "linear.bias").contiguous()
    query = _dev(query, torch.float32, "query").contiguous()
    q = lin_w.shape[0]
    assert lin_w.shape == (q, d) and lin_b.shape == (q,) and query.shape == (q,)
    out = torch.empty((b, d), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = int(lib.manner_hip_additive_pool_workspace_bytes(b, s, d, q))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=x.device)
    off = (-ws.data_ptr()) % 256
    with torch.cuda.device(x.device):
        _lib.check(lib.manner_hip_additive_pool_fused(_ptr(x), _ptr(lin_w), _ptr(lin_b), _ptr(query), b, s, d, q, _ptr(out),
                                                      ws.data_ptr() + off, nbytes, 1 if strict else 0, _stream()))
    return out


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """nn.Linear forward on f32 rows: x [R, K] -> [R, O]."""
    x = _dev(x, torch.float32, "input").contiguous()
    weight = _dev(weight, torch.float32, "weight").contiguous()
    r, k = x.shape
    o = weight.shape[0]
    assert weight.shape[1] == k
    if bias is not None:
        bias = _dev(bias, torch.float32, "bias").contiguous()
    y = torch.empty((r, o), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().manner_hip_linear(_ptr(x), _ptr(weight), _ptr(bias), r, k, o, _ptr(y), _stream()))
    return y


def entity_encode(entity_ids: Tensor, table: Tensor, in_proj_w: Tensor, in_proj_b: Tensor, out_proj_w: Tensor,
                  out_proj_b: Tensor, pool_w: Tensor, pool_b: Tensor, pool_q: Tensor, heads: int) -> Tensor:
    """MannerEntityEncoder.forward (eval), batch-faithful to the reference: [N, E] ids -> [N, D]."""
    ids = _dev(entity_ids, torch.int64, "entities").contiguous()
    n, e = ids.shape
    ts = [_dev(t, torch.float32, nm).contiguous() for t, nm in
          ((table, "pretrained_embedding.weight"), (in_proj_w, "in_proj_weight"), (in_proj_b, "in_proj_bias"),
           (out_proj_w, "out_proj.weight"), (out_proj_b, "out_proj.bias"), (pool_w, "linear.weight"),
           (pool_b, "linear.bias"), (pool_q, "query"))]
    d, q = ts[0].shape[1], ts[5].shape[0]
    out = torch.empty((n, d), dtype=torch.float32, device=ids.device)
    if n == 0:
        return out
    lib = _lib.load()
    need = int(lib.manner_hip_entity_workspace_bytes(n, e, d))
    ws = torch.empty(need, dtype=torch.uint8, device=ids.device)
    with torch.cuda.device(ids.device):
        _lib.check(lib.manner_hip_entity_encode(_ptr(ids), n, e, _ptr(ts[0]), ts[0].shape[0], d, heads, _ptr(ts[1]), _ptr(ts[2]),
                                                _ptr(ts[3]), _ptr(ts[4]), _ptr(ts[5]), _ptr(ts[6]), _ptr(ts[7]), q, _ptr(out),
                                                _ptr(ws), need, _ptr(device_status(ids.device).word), _stream()))
    return out


def mha_axis0(x: Tensor, in_proj_w: Tensor, in_proj_b: Tensor, out_proj_w: Tensor, out_proj_b: Tensor, heads: int) -> Tensor:
    """nn.MultiheadAttention(batch_first=False) exactly as the reference calls it on a [batch, seq, E] tensor without masks
    (PLMTextEncoder news_encoder.py:163-165, NRMSUserEncoder user_encoder.py:35-37): attention along AXIS 0 of x [L0, B1, E]."""
    x = _dev(x, torch.float32, "x").contiguous()
    l0, b1, e = x.shape
    ts = [_dev(t, torch.float32, nm).contiguous() for t, nm in ((in_proj_w, "in_proj_weight"), (in_proj_b, "in_proj_bias"),
                                                                (out_proj_w, "out_proj.weight"), (out_proj_b, "out_proj.bias"))]
    out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    lib = _lib.load()
    need = int(lib.manner_hip_mha_axis0_workspace_bytes(l0, b1, e))
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.manner_hip_mha_axis0(_ptr(x), l0, b1, e, heads, _ptr(ts[0]), _ptr(ts[1]), _ptr(ts[2]), _ptr(ts[3]), _ptr(out),
                                            _ptr(ws), need, _stream()))
    return out


def encode_full(cfg: EncoderConfig, weights: Dict[str, Tensor], ids: Tensor, mask: Tensor, precision: str = "fp32") -> Tensor:
    """HF ``last_hidden_state`` [N, Lp, H] f32 INCLUDING the padded positions (what PLMTextEncoder consumes); ``weights`` is the
    HF-named dict of float32 GPU tensors, read in place."""
    if precision not in ("fp32", "f16", "bf16"):
        raise ValueError("encode_full precision: fp32, f16 or bf16")
    ids, mask = _dev(ids, torch.int64, "input_ids").contiguous(), _dev(mask, torch.int64, "attention_mask").contiguous()
    if ids.dim() != 2 or ids.shape != mask.shape:
        raise ValueError(f"input_ids {tuple(ids.shape)} / attention_mask {tuple(mask.shape)} must be equal 2-D")
    canon = canonical_weights(cfg, weights)
    table = [_dev(canon[n].detach(), torch.float32, n).contiguous() for n in weight_table_order(cfg)]
    n, lp = ids.shape
    out = torch.empty((n, lp, cfg.hidden), dtype=torch.float32, device=ids.device)
    if n == 0:
        return out
    lib = _lib.load()
    cc = _lib.EncoderConfigC(cfg.arch, cfg.hidden, cfg.layers, cfg.heads, cfg.intermediate, cfg.vocab, cfg.max_pos, cfg.type_vocab,
                             cfg.pad_id, cfg.ln_eps)
    with torch.cuda.device(ids.device):
        need = int(lib.manner_hip_encode_full_workspace_bytes(C.byref(cc), n, lp))
        ws = torch.empty(need, dtype=torch.uint8, device=ids.device)
        tab = (C.c_void_p * len(table))(*[t.data_ptr() for t in table])
        _lib.check(lib.manner_hip_encode_full(C.byref(cc), tab, len(table), _ptr(ids), _ptr(mask), n, lp, _lib.PRECISIONS[precision],
                                              _ptr(out), _ptr(ws), need, _ptr(device_status(ids.device).word), _stream()))
    return out


def dot(user: Tensor, cand: Tensor) -> Tensor:
    """DotProduct contract: user [B,1,D], cand [B,D,C] (any strides, e.g. a permuted [B,C,D]) -> [B,C]."""
    user, cand = _dev(user, torch.float32, "clicked_news_vector"), _dev(cand, torch.float32, "candidate_news_vector")
    b, one, d = user.shape
    assert one == 1 and cand.shape[0] == b and cand.shape[1] == d
    c = cand.shape[2]
    u = user.reshape(b, d).contiguous()
    if c == 0:
        return torch.empty((b, 0), dtype=torch.float32, device=user.device)
    if min(cand.stride()) < 0 or (cand.stride(1) != 1 and cand.stride(2) != 1):
        cand = cand.contiguous()
    out = torch.empty((b, c), dtype=torch.float32, device=user.device)
    with torch.cuda.device(user.device):
        _lib.check(_lib.load().manner_hip_dot(_ptr(u), _ptr(cand), b, c, d, cand.stride(0), cand.stride(1),
                                              cand.stride(2), _ptr(out), _stream()))
    return out


class HalfTable:
    """IEEE-half copy of a news-embedding table for the scorer: ``rows`` float16 [n, D] and, when centred, ``mean`` float32 [D]
    (rows = half(T - mean)); see ``table_to_f16``."""

    def __init__(self, rows: Tensor, mean: Optional[Tensor]):
        self.rows, self.mean = rows, mean
        self.shape, self.device = rows.shape, rows.device


def score_late_fusion(table, hist_idx: Tensor, hist_off: Tensor, cand_idx: Tensor, cand_off: Tensor,
                      total_cand: Optional[int] = None, out: Optional[Tensor] = None) -> Tensor:
    """Ragged scores [sum c_i] of impressions given as CSR index lists into ``table`` [n, D] — float32, or the float16
    copy made by ``table_to_f16`` (half the bytes per gathered row; Infinity-Cache resident at the MIND-large shape)."""
    mean = None
    half = isinstance(table, HalfTable) or (isinstance(table, torch.Tensor) and table.dtype == torch.float16)
    if half:
        if isinstance(table, HalfTable):
            table, mean = table.rows, table.mean
        table = _dev(table, torch.float16, "table").contiguous()
    else:
        table = _dev(table, torch.float32, "table").contiguous()
    hist_idx, cand_idx = _dev(hist_idx, torch.int32, "hist_idx"), _dev(cand_idx, torch.int32, "cand_idx")
    hist_off, cand_off = _dev(hist_off, torch.int64, "hist_off"), _dev(cand_off, torch.int64, "cand_off")
    nb = hist_off.numel() - 1
    assert cand_off.numel() == nb + 1
    total = int(cand_idx.numel()) if total_cand is None else total_cand
    if out is None:
        out = torch.empty((total,), dtype=torch.float32, device=table.device)
    else:
        _dev(out, torch.float32, "out")
        assert out.is_contiguous() and out.numel() == total
    # bind the contiguous copies to locals: a temporary's storage could be handed to the next temporary
    # before the kernel has read it
    hist_idx, hist_off, cand_idx, cand_off = (hist_idx.contiguous(), hist_off.contiguous(), cand_idx.contiguous(),
                                              cand_off.contiguous())
    with torch.cuda.device(table.device):
        if half:
            _lib.check(_lib.load().manner_hip_score_late_fusion_f16(
                _ptr(table), _ptr(mean), table.shape[0], table.shape[1], _ptr(hist_idx), _ptr(hist_off), _ptr(cand_idx), _ptr(cand_off),
                nb, _ptr(out), _ptr(device_status(table.device).word), _stream()))
        else:
            _lib.check(_lib.load().manner_hip_score_late_fusion(
                _ptr(table), table.shape[0], table.shape[1], _ptr(hist_idx), _ptr(hist_off), _ptr(cand_idx), _ptr(cand_off),
                nb, _ptr(out), _ptr(device_status(table.device).word), _stream()))
    return out


def table_to_f16(table: Tensor, centre: bool = False, out: Optional[Tensor] = None):
    """IEEE-half copy of a news-embedding table for ``score_late_fusion`` (rows rounded to 11 mantissa bits).
    ``centre=False`` -> the float16 tensor half(T).  ``centre=True`` -> ``HalfTable(half(T - mean), mean)``: the rows are
    centred on the table's column mean first, which is what keeps the rounding away from the scores when the rows are
    nearly collinear (the usual state of one encoder's [CLS] vectors)."""
    table = _dev(table, torch.float32, "table").contiguous()
    n, d = table.shape
    if out is None:
        out = torch.empty(table.shape, dtype=torch.float16, device=table.device)
    else:
        _dev(out, torch.float16, "out")
        assert out.is_contiguous() and out.shape == table.shape
    lib = _lib.load()
    with torch.cuda.device(table.device):
        if not centre:
            _lib.check(lib.manner_hip_table_to_f16(_ptr(table), n, d, None, _ptr(out), None, 0, _stream()))
            return out
        mean = torch.empty(d, dtype=torch.float32, device=table.device)
        need = int(lib.manner_hip_table_to_f16_workspace_bytes(d))
        ws = torch.empty(need, dtype=torch.uint8, device=table.device)
        _lib.check(lib.manner_hip_table_to_f16(_ptr(table), n, d, _ptr(mean), _ptr(out), _ptr(ws), need, _stream()))
    return HalfTable(out, mean)


def score_user(table: Tensor, user: Tensor, cand_idx: Tensor, cand_off: Tensor, total_cand: Optional[int] = None) -> Tensor:
    """Ragged scores [sum c_i] = <user[i], table[cand_idx[j]]> for the candidates j of impression i (the early-fusion
    tail of CRModule.forward, cr_module.py:125-129, without the dense candidate tensor)."""
    table = _dev(table, torch.float32, "table").contiguous()
    user = _dev(user, torch.float32, "user_vector").contiguous()
    cand_idx = _dev(cand_idx, torch.int32, "cand_idx").contiguous()
    cand_off = _dev(cand_off, torch.int64, "cand_off").contiguous()
    nb = cand_off.numel() - 1
    assert user.shape == (nb, table.shape[1])
    total = int(cand_idx.numel()) if total_cand is None else total_cand
    out = torch.empty((total,), dtype=torch.float32, device=table.device)
    with torch.cuda.device(table.device):
        _lib.check(_lib.load().manner_hip_score_user(_ptr(table), table.shape[0], table.shape[1], _ptr(user), _ptr(cand_idx),
                                                     _ptr(cand_off), nb, _ptr(out), _ptr(device_status(table.device).word),
                                                     _stream()))
    return out


def to_dense(x: Tensor, off: Tensor, width: int, fill: Optional[Tensor] = None, with_mask: bool = False):
    """K9 (``to_dense_batch``): ragged rows x [sum n_i, *] + offsets int64 [B+1] -> dense [B, width, *]; padded slots
    hold 0 (or ``fill[b]``).  ``width`` is given by the caller — no device read-back.  With ``with_mask`` also the
    bool [B, width] mask of real slots."""
    x = _dev(x, torch.float32, "x").contiguous()
    off = _dev(off, torch.int64, "off").contiguous()
    nb = off.numel() - 1
    inner = tuple(x.shape[1:])
    d = 1
    for v in inner:
        d *= int(v)
    if fill is not None:
        fill = _dev(fill, torch.float32, "fill").contiguous()
        assert fill.numel() == nb
    dense = torch.empty((nb, width) + inner, dtype=torch.float32, device=x.device)
    mask = torch.empty((nb, width), dtype=torch.uint8, device=x.device) if with_mask else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().manner_hip_to_dense(_ptr(x), _ptr(off), nb, int(width), d, _ptr(fill), _ptr(dense), _ptr(mask),
                                                   _stream()))
    return (dense, mask.bool()) if with_mask else dense


def zscore_fuse(planes: Tensor, weights: Sequence[float], cand_off: Tensor, with_pad_value: bool = False):
    """planes [K, total] (module 0 = CR); weights of modules 1..K-1 -> fused ragged scores [total].  With
    ``with_pad_value`` also the per-impression value [B] that the reference's dense matrix holds in padded slots
    (its z-score runs over the zero-padded row, ensemble_module.py:145-149)."""
    planes = _dev(planes, torch.float32, "scores")
    if planes.dim() != 2 or planes.stride(1) != 1:
        planes = planes.contiguous()               # planes may be row views of a wider buffer (stride(0) > total)
    cand_off = _dev(cand_off, torch.int64, "cand_off").contiguous()
    k, total = planes.shape
    assert len(weights) == k - 1
    w = (C.c_float * max(1, k - 1))(*[float(v) for v in weights])
    out = torch.empty((total,), dtype=torch.float32, device=planes.device)
    pad = torch.empty((cand_off.numel() - 1,), dtype=torch.float32, device=planes.device) if with_pad_value else None
    with torch.cuda.device(planes.device):
        _lib.check(_lib.load().manner_hip_zscore_fuse(_ptr(planes), planes.stride(0) if k > 1 else total, k, w, _ptr(cand_off),
                                                      cand_off.numel() - 1, _ptr(out), _ptr(pad), _stream()))
    return (out, pad) if with_pad_value else out


def rank_ndcg(scores: Tensor, labels: Optional[Tensor], cand_off: Tensor, k: int = 10, with_mrr: bool = False):
    """Per impression: top-k candidate positions int32 [B,k] (-1 padded), nDCG@k float32 [B] and, with
    ``with_mrr``, the reciprocal rank of the best-ranked positive float32 [B]."""
    scores = _dev(scores, torch.float32, "scores").contiguous()
    cand_off = _dev(cand_off, torch.int64, "cand_off").contiguous()
    nb = cand_off.numel() - 1
    topk = torch.empty((nb, k), dtype=torch.int32, device=scores.device)
    ndcg = mrr = None
    if labels is not None:
        labels = _dev(labels, torch.float32, "labels").contiguous()
        ndcg = torch.empty((nb,), dtype=torch.float32, device=scores.device)
        if with_mrr:
            mrr = torch.empty((nb,), dtype=torch.float32, device=scores.device)
    with torch.cuda.device(scores.device):
        _lib.check(_lib.load().manner_hip_rank_ndcg(_ptr(scores), _ptr(labels), _ptr(cand_off), nb, k, _ptr(topk),
                                                    _ptr(ndcg), _ptr(mrr), _stream()))
    return (topk, ndcg, mrr) if with_mrr else (topk, ndcg)


def score_fuse_rank(tables: Sequence[Tensor], weights: Sequence[float], hist_idx: Tensor, hist_off: Tensor, cand_idx: Tensor, cand_off: Tensor,
                    labels: Optional[Tensor] = None, k: int = 10, with_pad_value: bool = False):
    """SURVEY §8e phase C in ONE launch (manner_hip_score_fuse_rank): K module tables [n, D] (D = 768 / 1024, float32) -> fused ragged
    scores [sum c_i] (K == 1: the raw late-fusion scores), top-k positions int32 [B, k], nDCG@k and MRR float32 [B] (with labels);
    bit-identical to score_late_fusion x K -> zscore_fuse -> rank_ndcg.  Returns a dict like hotpath.score_impressions."""
    tables = [_dev(t, torch.float32, "table").contiguous() for t in tables]
    kk = len(tables)
    assert kk >= 1 and len(weights) == kk - 1 and all(t.shape == tables[0].shape for t in tables)
    dev = tables[0].device
    hist_idx, cand_idx = _dev(hist_idx, torch.int32, "hist_idx").contiguous(), _dev(cand_idx, torch.int32, "cand_idx").contiguous()
    hist_off, cand_off = _dev(hist_off, torch.int64, "hist_off").contiguous(), _dev(cand_off, torch.int64, "cand_off").contiguous()
    nb, total = hist_off.numel() - 1, int(cand_idx.numel())
    scores = torch.empty((total,), dtype=torch.float32, device=dev)
    topk = torch.empty((nb, k), dtype=torch.int32, device=dev)
    pad = torch.empty((nb,), dtype=torch.float32, device=dev) if with_pad_value else None
    ndcg = mrr = None
    if labels is not None:
        labels = _dev(labels, torch.float32, "labels").contiguous()
        ndcg, mrr = torch.empty((nb,), dtype=torch.float32, device=dev), torch.empty((nb,), dtype=torch.float32, device=dev)
    lib = _lib.load()
    nbytes = int(lib.manner_hip_score_fuse_rank_workspace_bytes(kk, total))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    tp = (C.c_void_p * kk)(*[t.data_ptr() for t in tables])
    w = (C.c_float * max(1, kk - 1))(*[float(v) for v in weights])
    with torch.cuda.device(dev):
        _lib.check(lib.manner_hip_score_fuse_rank(tp, kk, w, tables[0].shape[0], tables[0].shape[1], _ptr(hist_idx), _ptr(hist_off), _ptr(cand_idx),
                                                  _ptr(cand_off), nb, total, _ptr(labels), k, _ptr(scores), _ptr(pad), _ptr(topk), _ptr(ndcg),
                                                  _ptr(mrr), ws.data_ptr(), nbytes, _ptr(device_status(dev).word), _stream()))
    res = {"scores": scores, "topk": topk, "ndcg": ndcg, "mrr": mrr}
    if with_pad_value:
        res["pad"] = pad
    return res


def aspect_metrics(topk: Tensor, cand_aspect: Tensor, cand_off: Tensor, num_classes: int,
                   hist_aspect: Optional[Tensor] = None, hist_off: Optional[Tensor] = None):
    """Aspect Diversity@k (and, with the history aspects, Personalization@k) per impression from the
    top-k positions of ``rank_ndcg``: float32 [B] each."""
    topk = _dev(topk, torch.int32, "topk").contiguous()
    cand_aspect = _dev(cand_aspect, torch.int32, "cand_aspect").contiguous()
    cand_off = _dev(cand_off, torch.int64, "cand_off").contiguous()
    nb, k = topk.shape
    div = torch.empty((nb,), dtype=torch.float32, device=topk.device)
    pers = None
    if hist_aspect is not None:
        hist_aspect = _dev(hist_aspect, torch.int32, "hist_aspect").contiguous()
        hist_off = _dev(hist_off, torch.int64, "hist_off").contiguous()
        pers = torch.empty((nb,), dtype=torch.float32, device=topk.device)
    with torch.cuda.device(topk.device):
        _lib.check(_lib.load().manner_hip_aspect_metrics(_ptr(topk), _ptr(cand_aspect), _ptr(hist_aspect), _ptr(cand_off),
                                                         _ptr(hist_off), nb, k, num_classes, _ptr(div), _ptr(pers), _stream()))
    return div, pers


def auc(scores: Tensor, labels: Tensor, sigmoid_rule: bool = True, return_counts: bool = False):
    """Global binary AUC over every (score, label) pair — torchmetrics ``AUROC(task="binary")`` as the
    reference feeds it (cr_module.py:81, :267-273).  Returns a float64 scalar tensor on the device (and, with
    ``return_counts``, the exact int64 triple ``[2U, P, N]``)."""
    scores = _dev(scores, torch.float32, "scores").contiguous().reshape(-1)
    labels = _dev(labels, torch.float32, "labels").contiguous().reshape(-1)
    n = scores.numel()
    if labels.numel() != n or n == 0:
        raise ValueError("auc: scores and labels must be non-empty and of equal size")
    lib = _lib.load()
    ws_bytes = lib.manner_hip_auc_workspace_bytes(n)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=scores.device)
    out = torch.empty((1,), dtype=torch.float64, device=scores.device)
    counts = torch.empty((3,), dtype=torch.int64, device=scores.device)
    with torch.cuda.device(scores.device):
        _lib.check(lib.manner_hip_auc(_ptr(scores), _ptr(labels), n, int(bool(sigmoid_rule)), _ptr(ws), ws_bytes, _ptr(out),
                                      _ptr(counts), _stream()))
    return (out[0], counts) if return_counts else out[0]


# ---------------------------------------------------------------- device-side collate
def collate_segments(off: Tensor) -> Tensor:
    """``repeat_interleave(arange(B), sizes)`` from int64 offsets [B+1] (reference _make_batch_assignees,
    mind_rec_dataset.py:171-174).  ``off[-1]`` is read on the host once to size the output."""
    off = _dev(off, torch.int64, "off").contiguous()
    nb = off.numel() - 1
    total = int(off[-1]) if nb > 0 else 0
    seg = torch.empty((total,), dtype=torch.int64, device=off.device)
    with torch.cuda.device(off.device):
        _lib.check(_lib.load().manner_hip_collate_segments(_ptr(off), nb, total, _ptr(seg), _stream()))
    return seg


def collate_segments_sized(off: Tensor, total: int) -> Tensor:
    """As ``collate_segments`` when the host already knows ``off[-1]`` (no device read)."""
    off = _dev(off, torch.int64, "off").contiguous()
    seg = torch.empty((total,), dtype=torch.int64, device=off.device)
    with torch.cuda.device(off.device):
        _lib.check(_lib.load().manner_hip_collate_segments(_ptr(off), off.numel() - 1, total, _ptr(seg), _stream()))
    return seg


def collate_text(store_ids: Tensor, store_len: Tensor, rows: Tensor, padded_len: int, pad_id: int):
    """BatchEncoding tensors (ids, mask: int64 [M, padded_len]) of the store rows ``rows``."""
    store_ids = _dev(store_ids, torch.int32, "store_ids").contiguous()
    store_len = _dev(store_len, torch.int32, "store_len").contiguous()
    rows = _dev(rows, torch.int32, "rows").contiguous()
    m = rows.numel()
    ids = torch.empty((m, padded_len), dtype=torch.int64, device=rows.device)
    mask = torch.empty((m, padded_len), dtype=torch.int64, device=rows.device)
    with torch.cuda.device(rows.device):
        _lib.check(_lib.load().manner_hip_collate_text(_ptr(store_ids), _ptr(store_len), store_ids.shape[0], store_ids.shape[1],
                                                       _ptr(rows), m, padded_len, pad_id, _ptr(ids), _ptr(mask), _stream()))
    return ids, mask


def collate_entities(store_ent: Tensor, store_cnt: Tensor, rows: Tensor, width: int) -> Tensor:
    """Entity index matrix int64 [M, width], right-padded with 0 (reference _tokenize_entities)."""
    store_ent = _dev(store_ent, torch.int32, "store_ent").contiguous()
    store_cnt = _dev(store_cnt, torch.int32, "store_cnt").contiguous()
    rows = _dev(rows, torch.int32, "rows").contiguous()
    m = rows.numel()
    out = torch.empty((m, width), dtype=torch.int64, device=rows.device)
    with torch.cuda.device(rows.device):
        _lib.check(_lib.load().manner_hip_collate_entities(_ptr(store_ent), _ptr(store_cnt), store_ent.shape[0],
                                                           store_ent.shape[1], _ptr(rows), m, width, _ptr(out), _stream()))
    return out


def collate_aspects(category: Tensor, sentiment: Tensor, sentiment_score: Tensor, rows: Tensor):
    """category / sentiment int64 [M] and sentiment_score float32 [M] of the store rows ``rows``."""
    category = _dev(category, torch.int32, "category").contiguous()
    sentiment = _dev(sentiment, torch.int32, "sentiment").contiguous()
    sentiment_score = _dev(sentiment_score, torch.float32, "sentiment_score").contiguous()
    rows = _dev(rows, torch.int32, "rows").contiguous()
    m = rows.numel()
    ocat = torch.empty((m,), dtype=torch.int64, device=rows.device)
    osent = torch.empty((m,), dtype=torch.int64, device=rows.device)
    oscore = torch.empty((m,), dtype=torch.float32, device=rows.device)
    with torch.cuda.device(rows.device):
        _lib.check(_lib.load().manner_hip_collate_aspects(_ptr(category), _ptr(sentiment), _ptr(sentiment_score),
                                                          category.numel(), _ptr(rows), m, _ptr(ocat), _ptr(osent), _ptr(oscore),
                                                          _stream()))
    return ocat, osent, oscore


def eval_loss(scores: Tensor, labels: Tensor, cand_off: Tensor, supcon: bool = True, temperature: float = 0.1,
              c_max: Optional[int] = None, reduce: bool = True):
    """Loss of ``CRModule.model_step`` (cr_module.py:140-171) from the ragged scores: SupCon on the score matrix
    (``supcon=True``, losses.py:12-40; batch value = mean of the non-zero per-impression losses, the
    pytorch_metric_learning default reducer for SupConLoss) or CrossEntropyLoss with probability targets over the
    zero-padded row of width ``c_max`` (default: the batch maximum; batch value = mean)."""
    scores = _dev(scores, torch.float32, "scores").contiguous()
    labels = _dev(labels, torch.float32, "labels").contiguous()
    cand_off = _dev(cand_off, torch.int64, "cand_off").contiguous()
    nb = cand_off.numel() - 1
    if not supcon and c_max is None:
        c_max = int((cand_off[1:] - cand_off[:-1]).max())
    losses = torch.empty((nb,), dtype=torch.float32, device=scores.device)
    with torch.cuda.device(scores.device):
        _lib.check(_lib.load().manner_hip_eval_loss(_ptr(scores), _ptr(labels), _ptr(cand_off), nb, 0 if supcon else 1,
                                                    C.c_float(temperature if supcon else 1.0), int(c_max or 1), _ptr(losses), _stream()))
    if not reduce:
        return losses
    if not supcon:
        return losses.mean()
    if supcon and (not bool((labels > 0.5).any()) or not bool((labels <= 0.5).any())):
        return losses.sum() * 0                       # losses.py:24-25, 40: no positive or no negative pair in the batch
    nz = losses > 0
    return torch.where(nz.any(), (losses * nz).sum() / nz.sum().clamp(min=1), losses.sum() * 0)
