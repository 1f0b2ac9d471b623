#!/usr/bin/env python3
"""Development aid behind DESIGN.md §2: which rounding points of the 16-bit encoder modes cost how
much ranking agreement.  A torch-CPU re-statement of the encoder with a `round` inserted exactly where the HIP kernels
round — GEMM operands (weights and activations), the residual stream, the stored activations (Q|K|V, attention output,
FFN intermediate) — for bf16 and f16, against the unrounded fp32 computation, on 256 title-profile news and 128
impressions over them, for both seeded weight sets.  Uses the oracle's embedding function only; not product code.

    python tools/precision_sim.py [n_news]
"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch.nn.functional as F
from manner_amd.config import PRESETS
from manner_amd.weights import make_plm_weights
from manner_amd.synth import synth_news_tokens, synth_impressions
import manner_oracle as O
torch.set_num_threads(8)
cfg = PRESETS['bert-base-uncased']
def rnd(t, dt):
    return t if dt is None else t.to(dt).float()
E4M3_MAX = 448.0
def q_e4m3(t, mode):
    """Round-trip through e4m3 (torch.float8_e4m3fn, the OCP format of v_mfma_*_f8f6f4): 'tensor' = one f32 scale for the
    whole tensor (amax -> 448), 'row' = one scale per row (per token / per output channel: folds into the GEMM epilogue),
    'block32' = one power-of-two (E8M0) scale per 32 consecutive K elements, the MX block format of
    v_mfma_scale_f32_16x16x128_f8f6f4."""
    if mode == 'tensor':
        s = t.abs().max().clamp(min=1e-30) / E4M3_MAX
    elif mode == 'row':
        s = t.abs().amax(-1, keepdim=True).clamp(min=1e-30) / E4M3_MAX
    else:
        k = t.shape[-1]
        b = t.reshape(*t.shape[:-1], k // 32, 32)
        s = torch.exp2(torch.ceil(torch.log2(b.abs().amax(-1, keepdim=True).clamp(min=1e-30) / E4M3_MAX)))
        return ((b / s).to(torch.float8_e4m3fn).float() * s).reshape(t.shape)
    return (t / s).to(torch.float8_e4m3fn).float() * s
def enc(ids, mask, w, op_dt, res_dt, act_dt, fp8=None, fp8_mode='tensor'):
    # fp8: None | 'ffn2' (only output.dense: K = I) | 'ffn' (both FFN GEMMs) | 'all' (all four GEMMs) take e4m3 operands
    # op_dt: GEMM operand dtype; res_dt: residual stream storage; act_dt: stored activations qkv/ctx/ffn
    ids, mask = torch.from_numpy(ids).long(), torch.from_numpy(mask)
    w = {k: torch.as_tensor(v) for k, v in w.items()}
    x = O.embeddings(ids, w, cfg)
    add = torch.zeros(mask.shape).masked_fill(mask == 0, torch.finfo(torch.float32).min)[:, None, None, :]
    n, s, h = x.shape; a, d = cfg.heads, cfg.head_dim
    x = rnd(x, res_dt)
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        def lin(t, name):
            use8 = fp8 == 'all' or (fp8 == 'ffn' and name in ("intermediate.dense", "output.dense")) or (fp8 == 'ffn2' and name == "output.dense")
            if use8:
                return F.linear(q_e4m3(rnd(t, op_dt), fp8_mode), q_e4m3(w[p+name+".weight"], 'row' if fp8_mode == 'tensor' else fp8_mode), w[p+name+".bias"])
            return F.linear(rnd(t, op_dt), rnd(w[p+name+".weight"], op_dt), w[p+name+".bias"])
        q = rnd(lin(x, "attention.self.query"), act_dt).view(n, s, a, d).transpose(1, 2)
        k = rnd(lin(x, "attention.self.key"), act_dt).view(n, s, a, d).transpose(1, 2)
        v = rnd(lin(x, "attention.self.value"), act_dt).view(n, s, a, d).transpose(1, 2)
        att = F.softmax(q @ k.transpose(2, 3) * d ** -0.5 + add, -1)
        ctx = rnd((rnd(att, op_dt) @ v).transpose(1, 2).reshape(n, s, h), act_dt)
        x = rnd(F.layer_norm(lin(ctx, "attention.output.dense") + x, (h,), w[p+"attention.output.LayerNorm.weight"], w[p+"attention.output.LayerNorm.bias"], cfg.ln_eps), res_dt)
        inter = rnd(F.gelu(lin(x, "intermediate.dense")), act_dt)
        x = rnd(F.layer_norm(lin(inter, "output.dense") + x, (h,), w[p+"output.LayerNorm.weight"], w[p+"output.LayerNorm.bias"], cfg.ln_eps), res_dt)
    return x[:, 0].contiguous()
FP8 = "fp8" in sys.argv[1:]
X2 = "x2" in sys.argv[1:]
sys.argv = [a for a in sys.argv if a not in ("fp8", "x2")]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ids, mask = synth_news_tokens(N, cfg, seed=42, max_len=96, profile='title')
imp = synth_impressions(128, N, seed=5)
def scores(tab):
    out = []
    for i in range(128):
        h = tab[imp['hist_idx'][imp['hist_off'][i]:imp['hist_off'][i+1]].astype(np.int64)].mean(0)
        c = tab[imp['cand_idx'][imp['cand_off'][i]:imp['cand_off'][i+1]].astype(np.int64)]
        out.append(c @ h)
    return out
def ndcg10(s_list):
    flat = torch.cat(s_list)
    return O.ndcg_at_k(flat, torch.from_numpy(imp['labels']), imp['cand_off'].tolist(), 10)[0]
def split22(t):
    """What an f16 [hi | lo] operand pair carries of an f32 tensor: hi = f16(t), lo = f16(t - hi) — 22 significant bits."""
    hi = t.to(torch.float16).float()
    return hi + (t - hi).to(torch.float16).float()
def enc_split(ids, mask, w, mode):
    """The x3 family (f32 residual stream, f32 stored activations and LayerNorm / softmax statistics, f32 accumulation; attention kept at
    the three-product precision in every variant — it is 7 % of the mode's time):
      x3   a_hi w_hi + a_hi w_lo + a_lo w_hi   (lo x lo dropped)       = the shipped parity-grade mode (K depth 3K)
      x2a  activations [hi | lo] x weights hi only  = a22 . f16(w)     (K depth 2K)
      x2w  weights [hi | lo] x activations hi only  = f16(a) . w22     (K depth 2K)"""
    ids, mask = torch.from_numpy(ids).long(), torch.from_numpy(mask)
    w = {k: torch.as_tensor(v) for k, v in w.items()}
    x = O.embeddings(ids, w, cfg)
    add = torch.zeros(mask.shape).masked_fill(mask == 0, torch.finfo(torch.float32).min)[:, None, None, :]
    n, s, h = x.shape; a, d = cfg.heads, cfg.head_dim
    f16 = lambda t: t.to(torch.float16).float()
    def mm(t, wt, b):
        if mode == "x3":
            th, wh = f16(t), f16(wt)
            tl, wl = f16(t - th), f16(wt - wh)
            return F.linear(th, wh) + F.linear(th, wl) + F.linear(tl, wh) + b
        if mode == "x2a":
            return F.linear(split22(t), f16(wt), b)
        return F.linear(f16(t), split22(wt), b)
    for l in range(cfg.layers):
        p = f"encoder.layer.{l}."
        lin = lambda t, name: mm(t, w[p+name+".weight"], w[p+name+".bias"])
        q = lin(x, "attention.self.query").view(n, s, a, d).transpose(1, 2)
        k = lin(x, "attention.self.key").view(n, s, a, d).transpose(1, 2)
        v = lin(x, "attention.self.value").view(n, s, a, d).transpose(1, 2)
        att = F.softmax(split22(q) @ split22(k).transpose(2, 3) * d ** -0.5 + add, -1)
        ctx = (split22(att) @ split22(v)).transpose(1, 2).reshape(n, s, h)
        x = F.layer_norm(lin(ctx, "attention.output.dense") + x, (h,), w[p+"attention.output.LayerNorm.weight"], w[p+"attention.output.LayerNorm.bias"], cfg.ln_eps)
        inter = F.gelu(lin(x, "intermediate.dense"))
        x = F.layer_norm(lin(inter, "output.dense") + x, (h,), w[p+"output.LayerNorm.weight"], w[p+"output.LayerNorm.bias"], cfg.ln_eps)
    return x[:, 0].contiguous()
if X2:
    # can a TWO-product split mode carry the parity grade?  Reference: the unrounded fp32 computation; float64 arbiter
    # for the reference's own f32 noise (how often the f32 reference's top-10 differs from the float64 evaluation of the same model).
    for std in (0.02, 0.05):
        w = make_plm_weights(cfg, seed=42, std=std)
        with torch.no_grad():
            ref = enc(ids, mask, w, None, None, None)
            sref = scores(ref)
            nref = ndcg10(sref)
            for name in ("x3", "x2a", "x2w"):
                t = enc_split(ids, mask, w, name)
                s = scores(t)
                agree = np.mean([torch.equal(torch.argsort(x, descending=True, stable=True)[:10], torch.argsort(y, descending=True, stable=True)[:10]) for x, y in zip(s, sref)])
                top1 = np.mean([int(torch.argmax(x)) == int(torch.argmax(y)) for x, y in zip(s, sref)])
                serr = max(float((x-y).abs().max()) for x, y in zip(s, sref))
                print(f"std {std} {name:4s} emb max err {float((t-ref).abs().max()):.3e}  score err {serr:.3e} (scale {float(sref[0].abs().max()):.0f}) "
                      f"top10 identical {agree:.3f} ({int(round(agree * 128))} / 128) top1 {top1:.3f}  |dnDCG@10| {abs(ndcg10(s) - nref):.2e}", flush=True)
            for name, (o, r, a_) in {"f16 all (the 16-mixed mode)": (torch.float16,)*3}.items():
                t = enc(ids, mask, w, o, r, a_)
                s = scores(t)
                agree = np.mean([torch.equal(torch.argsort(x, descending=True, stable=True)[:10], torch.argsort(y, descending=True, stable=True)[:10]) for x, y in zip(s, sref)])
                serr = max(float((x-y).abs().max()) for x, y in zip(s, sref))
                print(f"std {std} {name:4s} emb max err {float((t-ref).abs().max()):.3e}  score err {serr:.3e} top10 identical {agree:.3f}  |dnDCG@10| {abs(ndcg10(s) - nref):.2e}", flush=True)
    sys.exit(0)
if FP8:
    # e4m3 operands, measured.  f16 everywhere else (the headline arithmetic); per-tensor activation scale +
    # per-output-channel weight scale ("tensor"), per-row activation scale ("row"), MX block scales ("block32").
    for std in (0.02, 0.05):
        w = make_plm_weights(cfg, seed=42, std=std)
        with torch.no_grad():
            ref = enc(ids, mask, w, None, None, None)
            sref = scores(ref)
            nref = ndcg10(sref)
            legs = [("f16 all (headline arithmetic)", None, 'tensor')] + [(f"f16 + e4m3 {which} [{mode}]", which, mode)
                                                                          for which in ("ffn2", "ffn", "all") for mode in ("tensor", "row", "block32")]
            for name, which, mode in legs:
                t = enc(ids, mask, w, torch.float16, torch.float16, torch.float16, fp8=which, fp8_mode=mode)
                s = scores(t)
                agree = np.mean([torch.equal(torch.argsort(x, descending=True, stable=True)[:10], torch.argsort(y, descending=True, stable=True)[:10]) for x, y in zip(s, sref)])
                top1 = np.mean([int(torch.argmax(x)) == int(torch.argmax(y)) for x, y in zip(s, sref)])
                serr = max(float((x-y).abs().max()) for x, y in zip(s, sref))
                print(f"std {std} {name:36s} emb max err {float((t-ref).abs().max()):.3e}  score err {serr:.3e} (scale {float(sref[0].abs().max()):.0f}) "
                      f"top10 identical {agree:.3f} top1 {top1:.3f}  |dnDCG@10| {abs(ndcg10(s) - nref):.2e}", flush=True)
    sys.exit(0)
for std in (0.02, 0.05):
    w = make_plm_weights(cfg, seed=42, std=std)
    with torch.no_grad():
        ref = enc(ids, mask, w, None, None, None)
        sref = scores(ref)
        for name, (o, r, a_) in {"bf16 all": (torch.bfloat16,)*3, "bf16 ops, f32 resid": (torch.bfloat16, None, torch.bfloat16),
                                 "bf16 ops only (f32 resid+acts)": (torch.bfloat16, None, None),
                                 "f16 all": (torch.float16,)*3, "f16 ops, f32 resid": (torch.float16, None, torch.float16)}.items():
            t = enc(ids, mask, w, o, r, a_)
            s = scores(t)
            agree = np.mean([torch.equal(torch.argsort(x, descending=True, stable=True)[:10], torch.argsort(y, descending=True, stable=True)[:10]) for x, y in zip(s, sref)])
            serr = max(float((x-y).abs().max()) for x, y in zip(s, sref))
            print(f"std {std} {name:34s} emb max err {float((t-ref).abs().max()):.3e}  score err {serr:.3e} (scale {float(sref[0].abs().max()):.0f}) top10 agree {agree:.3f}", flush=True)
