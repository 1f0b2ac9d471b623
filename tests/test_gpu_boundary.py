"""The drop-in boundary's two host-side contracts added in round 6:

* the arithmetic of the mirrors follows the CALLER's autocast state — the reference's own `trainer.precision` knob
  (configs/trainer/default.yaml:12: `16-mixed` -> torch.autocast(float16) around every *_step; `bf16-mixed`; `32` = none), reference
  news_encoder.py:29-37 being plain torch whose arithmetic IS that state;
* the inference handle's packed weight copies notice parameter writes that bypass autograd's version counters (`p.data.mul_`)."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from manner_amd.config import PRESETS  # noqa: E402
from manner_amd.synth import synth_news_tokens  # noqa: E402
from manner_amd.weights import make_plm_weights  # noqa: E402

DEV = "cuda:0"
FP32_TOL = 1e-4


def _text_encoder(preset, seed, std, frozen=()):
    from manner_amd.models.components.news_encoder import MannerTextEncoder
    cfg = PRESETS[preset]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc = MannerTextEncoder(preset, list(frozen), 0.2)
    w = make_plm_weights(cfg, seed=seed, std=std)
    enc.load_state_dict({"plm_model." + k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    return enc.to(DEV), cfg


def test_eval_arithmetic_follows_the_callers_autocast_state(golden_dir):
    """16-mixed -> "f16", bf16-mixed -> "bf16", trainer.precision=32 -> the parity-grade "f16x3" (within 1e-4 of the reference's fp32
    output); a pinned mode overrides the state."""
    from manner_amd.models.components.news_encoder import autocast_mode
    z = np.load(os.path.join(golden_dir, "enc_bert_base.npz"))
    meta = json.loads(str(z["meta"]))
    enc, _ = _text_encoder(meta["preset"], meta["seed"], meta["std"])
    enc.eval()
    x = {"input_ids": torch.from_numpy(z["ids"]).to(DEV), "attention_mask": torch.from_numpy(z["mask"]).to(DEV)}
    assert enc.precision is None and enc.train_precision is None, "unset MANNER_HIP_(TRAIN_)PRECISION for this test"
    explicit = {}
    with torch.no_grad():
        for mode in ("f16", "bf16", "f16x3"):
            enc.precision = mode
            explicit[mode] = enc(x).clone()
        enc.precision = None
        assert autocast_mode() is None and enc.resolved_precision() == "f16x3"
        out32 = enc(x)                                                          # trainer.precision=32
        with torch.autocast("cuda", dtype=torch.float16):                        # 16-mixed
            assert autocast_mode() == "f16" and enc.resolved_precision() == "f16"
            out16 = enc(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):                       # bf16-mixed
            assert enc.resolved_precision() == "bf16"
            outbf = enc(x)
            enc.precision = "f16"                                               # a pinned mode wins over the state
            pinned = enc(x)
            enc.precision = None
        again32 = enc(x)                                                        # and back: the handle keeps every mode it has packed
        handle = enc._hip
        with torch.autocast("cuda", dtype=torch.float16):
            enc(x)
        assert enc._hip is handle, "alternating autocast states must not rebuild the engine once both modes are packed"
    enc.check_inputs()
    assert torch.equal(out16, explicit["f16"]) and torch.equal(outbf, explicit["bf16"]) and torch.equal(pinned, explicit["f16"])
    assert torch.equal(out32, explicit["f16x3"]) and torch.equal(again32, out32)
    err = np.abs(out32.cpu().numpy() - z["out"]).max()
    print(f"no autocast (f16x3) vs reference fp32 golden: {err:.3e}")
    assert err < FP32_TOL
    assert not torch.equal(out16, outbf) and not torch.equal(out16, out32)


def test_embedding_cache_rows_are_per_mode():
    """Rows cached under one autocast state are not served under another."""
    enc, cfg = _text_encoder("mini-roberta-large", 5, 0.03)
    enc.eval()
    enc.embedding_cache_rows = 64
    ids, mask = synth_news_tokens(12, cfg, seed=5, max_len=24)
    x = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    with torch.no_grad():
        a32 = enc(x).clone()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            abf = enc(x).clone()
        b32 = enc(x).clone()
        enc.embedding_cache_rows = 0
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ref_bf = enc(x)
        ref32 = enc(x)
    assert torch.equal(a32, ref32) and torch.equal(b32, ref32) and torch.equal(abf, ref_bf) and not torch.equal(ref32, ref_bf)


def test_training_arithmetic_follows_the_callers_autocast_state():
    """train(): fp16 autocast -> the "f16" engine (the caller's GradScaler scales the loss, as Lightning's 16-mixed plugin does),
    bf16 autocast -> "bf16", none -> "fp32" — forward values and gradients equal to the pinned-mode calls, bit for bit."""
    enc, cfg = _text_encoder("tiny-bert", 51, 0.05, frozen=[0])
    enc.train()
    enc.dropout.p = 0.0
    enc.plm_model.hidden_dropout_prob = enc.plm_model.attention_probs_dropout_prob = 0.0
    ids, mask = synth_news_tokens(8, cfg, seed=2, max_len=24)
    x = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    r = torch.randn(8, cfg.hidden, device=DEV, generator=torch.Generator(DEV).manual_seed(1))
    name = "plm_model.encoder.layer.1.output.dense.weight"

    def run(ctx, pinned, scale=1.0):
        enc.train_precision = pinned
        enc.zero_grad(set_to_none=True)
        with ctx:
            mode = enc.resolved_train_precision()
            out = enc(x)
        ((out * r).sum() * scale).backward()
        return mode, out.detach().clone(), dict(enc.named_parameters())[name].grad.clone()

    import contextlib
    for ctx, want, scale in ((torch.autocast("cuda", dtype=torch.float16), "f16", 1024.0),
                             (torch.autocast("cuda", dtype=torch.bfloat16), "bf16", 1.0), (contextlib.nullcontext(), "fp32", 1.0)):
        mode, out_a, g_a = run(ctx, None, scale)
        assert mode == want
        _, out_p, g_p = run(contextlib.nullcontext(), want, scale)
        assert torch.equal(out_a, out_p) and torch.equal(g_a, g_p), want
        assert torch.isfinite(g_a).all()
    enc.train_precision = None


def test_baseline_text_encoder_follows_autocast_too():
    from manner_amd.models.components.news_encoder import PLMTextEncoder
    cfg = PRESETS["tiny-bert"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc = PLMTextEncoder("tiny-bert", [], cfg.hidden, 2, 32, 0.2).to(DEV).eval()
    ids, mask = synth_news_tokens(6, cfg, seed=3, max_len=16)
    x = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    with torch.no_grad():
        o32 = enc(x)
        with torch.autocast("cuda", dtype=torch.float16):
            o16 = enc(x)
        enc.precision = "f16"
        p16 = enc(x)
        enc.precision = "fp32"
        p32 = enc(x)
    assert torch.equal(o32, p32) and torch.equal(o16, p16) and not torch.equal(o32, o16)


def test_eval_handle_notices_parameter_writes_through_p_data():
    """`p.data.mul_(2)` does not bump `p._version` (the .data alias has its own counter): the packed copies of the inference handle
    would stay stale.  async (default): the sampled device fingerprint raises at the forward after the stale one and rebuilds;
    `invalidate()`: the very next forward is right; sync: the very next forward is right without any call."""
    enc, cfg = _text_encoder("tiny-bert", 7, 0.05)
    enc.eval()
    assert enc.weight_fingerprint == "async"
    ids, mask = synth_news_tokens(8, cfg, seed=7, max_len=20)
    x = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    p = dict(enc.named_parameters())["plm_model.encoder.layer.1.intermediate.dense.weight"]
    with torch.no_grad():
        out0 = enc(x).clone()
        assert torch.equal(enc(x), out0)
        v = p._version
        p.data.mul_(2.0)
        assert p._version == v                                 # the write autograd does not see
        enc(x)                                                 # may still be the old values (the snapshot in flight predates the write)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="invalidate"):
            enc(x)
        out2 = enc(x).clone()                                  # rebuilt handle
        assert not torch.equal(out2, out0)
        fresh, _ = _text_encoder("tiny-bert", 7, 0.05)
        fresh.eval()
        dict(fresh.named_parameters())["plm_model.encoder.layer.1.intermediate.dense.weight"].mul_(2.0)
        assert torch.equal(fresh(x), out2)
        # the exact tool: invalidate() right after the write
        p.data.mul_(0.5)
        enc.invalidate()
        assert torch.equal(enc(x), out0)
        assert torch.equal(enc(x), out0)
        # sync mode: the very next forward
        enc.weight_fingerprint = "sync"
        p.data.mul_(2.0)
        assert torch.equal(enc(x), out2)
        p.data.mul_(0.5)
        assert torch.equal(enc(x), out0)
        # autograd-visible writes never needed any of this
        enc.weight_fingerprint = "0"
        enc.invalidate()
        enc(x)
        p.mul_(2.0)
        assert torch.equal(enc(x), out2)
    enc.check_inputs()


def test_a_fused_optimizer_step_reaches_the_eval_handle():
    """torch.optim.AdamW(fused=True) rewrites the parameters WITHOUT bumping Parameter._version (tools/version_probe.py: 0 -> 0 on
    torch 2.10 / ROCm) — the (address, version) key of the inference handle cannot see it.  The global optimiser post-step hook can:
    the eval() forward after a fused step computes with the new values, and nothing raises."""
    enc, cfg = _text_encoder("tiny-bert", 9, 0.05)
    ids, mask = synth_news_tokens(8, cfg, seed=9, max_len=20)
    x = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    enc.eval()
    with torch.no_grad():
        out0 = enc(x).clone()
    p = dict(enc.named_parameters())["plm_model.encoder.layer.1.output.dense.weight"]
    opt = torch.optim.AdamW([p], lr=1e-2, fused=True)
    v0 = p._version
    p.grad = torch.ones_like(p)
    opt.step()
    fused_bumps = p._version != v0                          # False on torch 2.10: exactly the case this test is about
    with torch.no_grad():
        out1 = enc(x).clone()
        out2 = enc(x).clone()                               # and no tripwire error on the forward after
    assert not torch.equal(out1, out0) and torch.equal(out2, out1), fused_bumps
    fresh, _ = _text_encoder("tiny-bert", 9, 0.05)
    fresh.eval()
    with torch.no_grad():
        dict(fresh.named_parameters())["plm_model.encoder.layer.1.output.dense.weight"].copy_(p)
        assert torch.equal(fresh(x), out1)
    enc.check_inputs()
