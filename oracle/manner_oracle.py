"""CPU ORACLE for the MANNeR news-encoding + candidate-scoring hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / reported CPU baseline.  The product path
(``manner_amd``) never routes through it and has no CPU fallback.

It is a plain PyTorch fp32 (CPU) restatement of the reference's algorithm; each
function cites the reference lines it follows.  The encoder arithmetic lives in
a third-party dependency of the reference (``transformers`` ``BertModel`` /
``RobertaModel``, pinned only as ``>=4.29.2`` in reference requirements.txt:25;
the container has 5.15.0), so that part restates the published BERT algorithm
and is anchored on the reference's call site news_encoder.py:20,30-34.
``to_dense_batch`` (torch_geometric) and ``RetrievalNormalizedDCG`` (torchmetrics)
are not installed and are restated from their documented semantics.

Pinning: the reference ships no tests or golden vectors.  The
oracle is pinned by outputs of the reference itself, run in the build container:
``tests/golden/make_golden.py`` imports the reference's leaf modules
(MannerNewsEncoder / AdditiveAttention / NAMLUserEncoder / DotProduct) and HF
``BertModel`` / ``RobertaModel`` on seeded inputs and commits inputs + outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against them.
The Lightning-level forwards (cr_module.py:105-131, ensemble_module.py:95-151)
cannot be imported here (lightning / torch_geometric / torchmetrics absent) and
are restated line by line; for those the fixtures pin self-consistency only.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
ARCH_BERT, ARCH_ROBERTA = 0, 1


def _t(x) -> Tensor:
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(x)


# --------------------------------------------------------------------------- encoder (K1-K7)

_DISTIL = (("attention.q_lin.", "attention.self.query."), ("attention.k_lin.", "attention.self.key."),
           ("attention.v_lin.", "attention.self.value."), ("attention.out_lin.", "attention.output.dense."),
           ("sa_layer_norm.", "attention.output.LayerNorm."), ("ffn.lin1.", "intermediate.dense."),
           ("ffn.lin2.", "output.dense."), ("output_layer_norm.", "output.LayerNorm."))


def bert_named(w: Dict[str, Tensor], cfg) -> Dict[str, Tensor]:
    """DistilBertModel (transformers/models/distilbert/modeling_distilbert.py) is the BertModel block without
    token-type embeddings and pooler: Embeddings.forward = LN(word + position) (:90-118), TransformerBlock.forward =
    sa_layer_norm(attn + x), output_layer_norm(ffn + .) (:270-300), q scaled by 1/sqrt(d_head) before the scores
    (:180-190: same product).  Its state dict is renamed to BertModel names with an all-zero token-type row, so the
    functions below serve both."""
    if getattr(cfg, "naming", "bert") != "distilbert":
        return w
    out = {}
    for k, v in w.items():
        if k.startswith("transformer.layer."):
            k = "encoder.layer." + k[len("transformer.layer."):]
            for a, b in _DISTIL:
                if a in k:
                    k = k.replace(a, b)
                    break
        out[k] = v
    out["embeddings.token_type_embeddings.weight"] = torch.zeros((1, cfg.hidden))
    return out


def position_ids(ids: Tensor, arch: int, pad_id: int) -> Tensor:
    """BERT: arange(L) (transformers/models/bert/modeling_bert.py:65,83-84).
    RoBERTa: cumsum(ids != pad) * (ids != pad) + pad
    (transformers/models/roberta/modeling_roberta.py:142-155)."""
    n, l = ids.shape
    if arch == ARCH_BERT:
        return torch.arange(l).unsqueeze(0).expand(n, l)
    m = ids.ne(pad_id).int()
    return (torch.cumsum(m, dim=1).type_as(m) * m).long() + pad_id


def embeddings(ids: Tensor, w: Dict[str, Tensor], cfg) -> Tensor:
    """K1: LN(word[ids] + type[0] + pos[position_ids]) — modeling_bert.py:68-108.
    token_type is all zeros because the reference collate passes
    return_token_type_ids=False (mind_rec_dataset.py:134-137)."""
    pos = position_ids(ids, cfg.arch, cfg.pad_id)
    x = w["embeddings.word_embeddings.weight"][ids]
    x = x + w["embeddings.token_type_embeddings.weight"][0]
    x = x + w["embeddings.position_embeddings.weight"][pos]
    return F.layer_norm(x, (cfg.hidden,), w["embeddings.LayerNorm.weight"],
                        w["embeddings.LayerNorm.bias"], cfg.ln_eps)


def encoder_layer(x: Tensor, add_mask: Tensor, w: Dict[str, Tensor], l: int, cfg) -> Tensor:
    """K2-K6: one BertLayer (modeling_bert.py:175-203, 289-293, 334-337, 347-351)."""
    p = f"encoder.layer.{l}."
    n, s, h = x.shape
    a, d = cfg.heads, cfg.head_dim

    def lin(t, name):
        return F.linear(t, w[p + name + ".weight"], w[p + name + ".bias"])

    q = lin(x, "attention.self.query").view(n, s, a, d).transpose(1, 2)
    k = lin(x, "attention.self.key").view(n, s, a, d).transpose(1, 2)
    v = lin(x, "attention.self.value").view(n, s, a, d).transpose(1, 2)
    att = torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5) + add_mask       # :128-131
    att = F.softmax(att, dim=-1)
    ctx = torch.matmul(att, v).transpose(1, 2).reshape(n, s, h)
    x = F.layer_norm(lin(ctx, "attention.output.dense") + x, (h,),
                     w[p + "attention.output.LayerNorm.weight"],
                     w[p + "attention.output.LayerNorm.bias"], cfg.ln_eps)
    inter = F.gelu(lin(x, "intermediate.dense"))                             # exact erf GeLU
    x = F.layer_norm(lin(inter, "output.dense") + x, (h,),
                     w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], cfg.ln_eps)
    return x


def encode_tokens(ids: Tensor, mask: Tensor, w: Dict[str, Tensor], cfg,
                  layers: Optional[int] = None) -> Tensor:
    """last_hidden_state [N,L,H] of BertModel.forward (modeling_bert.py:623-686) in eval mode."""
    ids, mask = _t(ids).long(), _t(mask)
    w = bert_named({k: _t(v) for k, v in w.items()}, cfg)
    x = embeddings(ids, w, cfg)
    # additive padding mask over keys (create_bidirectional_mask, modeling_bert.py:704-708)
    add_mask = torch.zeros(mask.shape, dtype=torch.float32)
    add_mask = add_mask.masked_fill(mask == 0, torch.finfo(torch.float32).min)[:, None, None, :]
    for l in range(cfg.layers if layers is None else layers):
        x = encoder_layer(x, add_mask, w, l, cfg)
    return x


def encode_cls(ids, mask, w, cfg) -> Tensor:
    """K7: MannerTextEncoder.forward (reference news_encoder.py:29-37): CLS slice of the
    PLM's last hidden state; dropout is identity in eval()."""
    with torch.no_grad():
        return encode_tokens(ids, mask, w, cfg)[:, 0, :].contiguous()


def encode_cls_train(ids, mask, w: Dict[str, Tensor], cfg, p_hidden: float = 0.0, p_attn: float = 0.0, p_out: float = 0.0,
                     keep=None, start_layer: int = 0, prefix: Optional[Tensor] = None) -> Tensor:
    """MannerTextEncoder.forward in train() mode (reference news_encoder.py:29-37), differentiable: torch autograd over
    this function is the gradient oracle of SURVEY §8f-3.  HF BertModel's dropouts sit after the embedding LayerNorm
    (modeling_bert.py:107), on the attention probabilities (:140), after attention.output.dense (:183) and after
    output.dense (:351); the reference adds one on the [CLS] vector (news_encoder.py:35).  ``keep(site, kind)`` supplies
    the 0/1 keep masks (float, padded layout: kind "rows" [N, L, H], "attn" [N, heads, L, L], "cls" [N, H]) so that a
    test can replay the masks of the implementation under test; site numbering: 0 embeddings, 1 [CLS], 8*(layer+1)+{0
    attention probabilities, 1 attention output, 2 FFN output}.  ``w`` tensors may require grad.
    ``start_layer`` / ``prefix``: start from hidden_states[start_layer] = prefix [N, L, H]."""
    ids, mask = _t(ids).long(), _t(mask)
    w = bert_named(dict(w), cfg)
    n, s = ids.shape
    h, a, d = cfg.hidden, cfg.heads, cfg.head_dim

    def drop(x, p, site, kind):
        if p <= 0.0:
            return x
        return x * keep(site, kind) / (1.0 - p)

    if start_layer == 0:
        x = drop(embeddings(ids, w, cfg), p_hidden, 0, "rows")
    else:
        x = prefix
    add_mask = torch.zeros(mask.shape, dtype=torch.float32)
    add_mask = add_mask.masked_fill(mask == 0, torch.finfo(torch.float32).min)[:, None, None, :]
    for l in range(start_layer, cfg.layers):
        p = f"encoder.layer.{l}."

        def lin(t, name):
            return F.linear(t, w[p + name + ".weight"], w[p + name + ".bias"])

        q = lin(x, "attention.self.query").view(n, s, a, d).transpose(1, 2)
        k = lin(x, "attention.self.key").view(n, s, a, d).transpose(1, 2)
        v = lin(x, "attention.self.value").view(n, s, a, d).transpose(1, 2)
        att = F.softmax(torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5) + add_mask, dim=-1)
        att = drop(att, p_attn, 8 * (l + 1), "attn")
        ctx = torch.matmul(att, v).transpose(1, 2).reshape(n, s, h)
        x = F.layer_norm(drop(lin(ctx, "attention.output.dense"), p_hidden, 8 * (l + 1) + 1, "rows") + x, (h,),
                         w[p + "attention.output.LayerNorm.weight"], w[p + "attention.output.LayerNorm.bias"], cfg.ln_eps)
        inter = F.gelu(lin(x, "intermediate.dense"))
        x = F.layer_norm(drop(lin(inter, "output.dense"), p_hidden, 8 * (l + 1) + 2, "rows") + x, (h,),
                         w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], cfg.ln_eps)
    return drop(x[:, 0, :], p_out, 1, "cls")


# --------------------------------------------------------------------------- entity branch (K8)

def entity_encoder(entity_ids: Tensor, w: Dict[str, Tensor], heads: int) -> Tensor:
    """MannerEntityEncoder.forward (reference news_encoder.py:60-72), eval mode, INCLUDING quirk Q1: the
    nn.MultiheadAttention is batch_first=False (news_encoder.py:52-54) but receives [N, E, D], so the
    sequence axis is the N news of the call and the batch axis the entity slot; no key_padding_mask.
    ``w`` uses the reference's keys: pretrained_embedding.weight, multihead_attention.{in_proj_weight,
    in_proj_bias, out_proj.weight, out_proj.bias}, additive_attention.{linear.weight, linear.bias, query}."""
    w = {k: _t(v) for k, v in w.items()}
    ev = w["pretrained_embedding.weight"][_t(entity_ids).long()]                       # [N, E, D]
    n, e, d = ev.shape
    dh = d // heads
    qkv = ev @ w["multihead_attention.in_proj_weight"].T + w["multihead_attention.in_proj_bias"]
    q, k, v = qkv.split(d, dim=-1)

    def split_heads(t):                                   # torch MHA: (L, B*heads, dh) with L = N, B = E
        return t.reshape(n, e * heads, dh).transpose(0, 1)

    att = torch.softmax((split_heads(q) * dh ** -0.5) @ split_heads(k).transpose(1, 2), dim=-1)
    o = (att @ split_heads(v)).transpose(0, 1).reshape(n, e, d)
    o = o @ w["multihead_attention.out_proj.weight"].T + w["multihead_attention.out_proj.bias"]
    return additive_attention(o, w["additive_attention.linear.weight"], w["additive_attention.linear.bias"],
                              w["additive_attention.query"])


def mha_axis0(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor, heads: int) -> Tensor:
    """nn.MultiheadAttention(embed_dim, heads) in eval mode called as the reference does — batch_first=False on a
    [batch, seq, E] tensor, no masks (news_encoder.py:163-165; user_encoder.py:35-37; torch/nn/functional.py
    multi_head_attention_forward): the sequence axis is axis 0."""
    x = _t(x)
    l0, b1, e = x.shape
    dh = e // heads
    q, k, v = (x @ _t(in_w).T + _t(in_b)).split(e, dim=-1)

    def split_heads(t):                                   # (L0, B1 * heads, dh) -> (B1 * heads, L0, dh)
        return t.reshape(l0, b1 * heads, dh).transpose(0, 1)

    att = torch.softmax((split_heads(q) * dh ** -0.5) @ split_heads(k).transpose(1, 2), dim=-1)
    o = (att @ split_heads(v)).transpose(0, 1).reshape(l0, b1, e)
    return o @ _t(out_w).T + _t(out_b)


def plm_text_encoder(ids, mask, w: Dict[str, Tensor], cfg, mha: Dict[str, Tensor], pool: Tuple[Tensor, Tensor, Tensor], heads: int) -> Tensor:
    """PLMTextEncoder.forward (reference news_encoder.py:158-171), eval mode: HF last_hidden_state (padded positions
    included — encode_tokens computes them as HF does), un-masked axis-0 attention, un-masked additive pooler.
    ``mha``: in_proj_weight / in_proj_bias / out_proj.weight / out_proj.bias."""
    with torch.no_grad():
        hidden = encode_tokens(ids, mask, w, cfg)
        mixed = mha_axis0(hidden, mha["in_proj_weight"], mha["in_proj_bias"], mha["out_proj.weight"], mha["out_proj.bias"], heads)
        return additive_attention(mixed, *pool)


def nrms_user_encoder(clicked: Tensor, mha: Dict[str, Tensor], pool: Tuple[Tensor, Tensor, Tensor], heads: int) -> Tensor:
    """NRMSUserEncoder.forward (reference user_encoder.py:33-42), eval mode."""
    with torch.no_grad():
        return additive_attention(mha_axis0(clicked, mha["in_proj_weight"], mha["in_proj_bias"], mha["out_proj.weight"],
                                            mha["out_proj.bias"], heads), *pool)


def news_encoder_train(ids, mask, entity_ids, w_text: Dict[str, Tensor], w_ent: Dict[str, Tensor], cfg, heads: int, **train_kw) -> Tensor:
    """MannerNewsEncoder.forward with use_entities=True in train() mode (reference news_encoder.py:115-129 over :60-72),
    differentiable, entity-side dropout off (the text side takes encode_cls_train's keyword arguments): the gradient oracle
    of the entity branch.  ``w_ent``: the reference's keys (entity_encoder.pretrained_embedding.weight,
    entity_encoder.multihead_attention.*, entity_encoder.additive_attention.*, linear.weight, linear.bias); the embedding
    is built with padding_idx=0 (news_encoder.py:99-103), so row 0 receives no gradient."""
    text = encode_cls_train(ids, mask, w_text, cfg, **train_kw)
    p = "entity_encoder."
    ev = F.embedding(_t(entity_ids).long(), w_ent[p + "pretrained_embedding.weight"], padding_idx=0)
    ev = mha_axis0(ev, w_ent[p + "multihead_attention.in_proj_weight"], w_ent[p + "multihead_attention.in_proj_bias"],
                   w_ent[p + "multihead_attention.out_proj.weight"], w_ent[p + "multihead_attention.out_proj.bias"], heads)
    ev = additive_attention(ev, w_ent[p + "additive_attention.linear.weight"], w_ent[p + "additive_attention.linear.bias"],
                            w_ent[p + "additive_attention.query"])
    return F.linear(torch.cat([text, ev], dim=-1), w_ent["linear.weight"], w_ent["linear.bias"])


def news_encoder_with_entities(text_vec: Tensor, entity_vec: Tensor, lin_w: Tensor, lin_b: Tensor) -> Tensor:
    """MannerNewsEncoder.forward, use_entities=True (reference news_encoder.py:119-124):
    linear(cat([text_vector, entity_vector], dim=-1))."""
    return F.linear(torch.cat([_t(text_vec), _t(entity_vec)], dim=-1), _t(lin_w), _t(lin_b))


# --------------------------------------------------------------------------- pooler / scorer

def additive_attention(x: Tensor, lin_w: Tensor, lin_b: Tensor, query: Tensor) -> Tensor:
    """K11: AdditiveAttention.forward (reference attention.py:21-27), no padding mask (Q2)."""
    x, lin_w, lin_b, query = _t(x), _t(lin_w), _t(lin_b), _t(query)
    att = torch.tanh(F.linear(x, lin_w, lin_b))
    wts = F.softmax(torch.matmul(att, query), dim=1)
    return torch.bmm(wts.unsqueeze(1), x).squeeze(1)


def dot_product(user: Tensor, cand: Tensor) -> Tensor:
    """K12: DotProduct.forward (reference click_predictors.py:9-12): user [B,1,D], cand [B,D,C]."""
    return torch.bmm(_t(user), _t(cand)).squeeze(1)


def to_dense_batch(x: Tensor, batch: Tensor) -> Tuple[Tensor, Tensor]:
    """K9: torch_geometric.utils.to_dense_batch restated: ``batch`` sorted
    ascending segment ids; output zero-filled [B, max_count, *] + bool mask."""
    x, batch = _t(x), _t(batch).long()
    b = int(batch.max()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=b)
    mx = int(counts.max()) if b else 0
    starts = torch.cumsum(counts, 0) - counts
    pos = torch.arange(batch.numel()) - starts[batch]
    out = torch.zeros((b, mx) + tuple(x.shape[1:]), dtype=x.dtype)
    mask = torch.zeros((b, mx), dtype=torch.bool)
    out[batch, pos] = x
    mask[batch, pos] = True
    return out, mask


def cr_scores(hist_vec: Tensor, batch_hist: Tensor, cand_vec: Tensor, batch_cand: Tensor,
              late_fusion: bool = True,
              user_encoder: Optional[Tuple[Tensor, Tensor, Tensor]] = None) -> Tensor:
    """CRModule.forward after the two news_encoder calls (reference cr_module.py:108-131).
    Returns scores [B, Cmax]; padded candidate slots score exactly 0."""
    hist_agg, mask_hist = to_dense_batch(hist_vec, batch_hist)
    cand_agg, _ = to_dense_batch(cand_vec, batch_cand)
    if late_fusion:
        hist_size = mask_hist.sum(dim=1)                                    # :117-120
        user = torch.div(hist_agg.sum(dim=1), hist_size.unsqueeze(-1))      # :121-123
    else:
        user = additive_attention(hist_agg, *user_encoder)                  # :125
    return dot_product(user.unsqueeze(1), cand_agg.permute(0, 2, 1))        # :127-129


def zscore(scores: Tensor, mask_cand: Tensor) -> Tensor:
    """K13: EnsembleModule._submodel_forward z-normalisation (reference ensemble_module.py:138-149):
    mean = sum over the PADDED row / c_i (Q3), std = unbiased torch.std over valid entries."""
    cand_size = mask_cand.sum(dim=1)
    std = torch.stack([torch.std(scores[i][mask_cand[i]]) for i in range(mask_cand.shape[0])]).unsqueeze(-1)
    mean = torch.div(torch.sum(scores, dim=1), cand_size).unsqueeze(-1).expand_as(scores)
    return torch.div(scores - mean, std)


def ensemble_scores(module_vecs: Sequence[Tuple[Tensor, Tensor]], batch_hist: Tensor,
                    batch_cand: Tensor, weights: Sequence[float]) -> Tensor:
    """K14: EnsembleModule.forward (reference ensemble_module.py:95-109).
    ``module_vecs[k] = (hist_vec, cand_vec)`` of module k (CR first); ``weights[k-1]`` is the
    weight of module k>=1; a zero weight skips the module exactly as the reference does."""
    _, mask_cand = to_dense_batch(torch.zeros(_t(batch_cand).numel()), batch_cand)

    def sub(k):
        s = cr_scores(module_vecs[k][0], batch_hist, module_vecs[k][1], batch_cand, late_fusion=True)
        return zscore(s, mask_cand)

    scores = sub(0)
    for k, wk in enumerate(weights, start=1):
        if wk != 0:
            scores = scores + wk * sub(k)      # reference does `scores += w * s` in place
    return scores


def ragged(scores: Tensor, batch_cand: Tensor) -> Tensor:
    """Flatten [B,Cmax] back to the ragged candidate order (what cr_module.py:267-273 feeds
    to the metrics via the candidate mask)."""
    _, mask = to_dense_batch(torch.zeros(_t(batch_cand).numel()), batch_cand)
    return scores[mask]


# --------------------------------------------------------------------------- ranking / nDCG (K15)

def topk_indices(scores: Tensor, offsets: Sequence[int], k: int) -> List[List[int]]:
    """Per impression, candidate positions sorted by descending score (stable: ties keep the
    lower position first), truncated to k."""
    out = []
    for i in range(len(offsets) - 1):
        s = scores[offsets[i]:offsets[i + 1]]
        order = torch.argsort(s, descending=True, stable=True)
        out.append(order[:k].tolist())
    return out


def ndcg_at_k(scores: Tensor, target: Tensor, offsets: Sequence[int], k: int) -> Tuple[float, Tensor]:
    """RetrievalNormalizedDCG(top_k=k) restated (constructed at reference cr_module.py:83-84):
    per impression DCG@k = sum_{r<k} target_r / log2(r+2) on preds sorted descending, IDCG@k on
    targets sorted descending, impressions without a positive score 0.0
    (empty_target_action='neg'), mean over impressions.  Tie-free semantics."""
    per = []
    for i in range(len(offsets) - 1):
        s = scores[offsets[i]:offsets[i + 1]]
        t = target[offsets[i]:offsets[i + 1]].double()
        if t.sum() == 0:
            per.append(0.0)
            continue
        order = torch.argsort(s, descending=True, stable=True)[:k]
        disc = 1.0 / torch.log2(torch.arange(order.numel(), dtype=torch.float64) + 2.0)
        dcg = (t[order] * disc).sum()
        ideal = torch.sort(t, descending=True).values[:k]
        idcg = (ideal * disc[: ideal.numel()]).sum()
        per.append(float(dcg / idcg))
    per_t = torch.tensor(per, dtype=torch.float64)
    return float(per_t.mean()) if per else 0.0, per_t


def mrr(scores: Tensor, target: Tensor, offsets: Sequence[int]) -> Tuple[float, Tensor]:
    """RetrievalMRR restated (constructed at reference cr_module.py:82): per impression the reciprocal
    rank of the first positive in descending-score order, 0 without a positive; mean over impressions."""
    per = []
    for i in range(len(offsets) - 1):
        s, t = scores[offsets[i]:offsets[i + 1]], target[offsets[i]:offsets[i + 1]]
        hit = torch.nonzero(t[torch.argsort(s, descending=True, stable=True)] > 0)
        per.append(1.0 / (int(hit[0]) + 1) if hit.numel() else 0.0)
    per_t = torch.tensor(per, dtype=torch.float64)
    return float(per_t.mean()) if per else 0.0, per_t


def binary_auroc(scores: Tensor, target: Tensor, sigmoid_rule: bool = True) -> Tuple[float, Tuple[int, int, int]]:
    """AUROC(task="binary") restated (constructed at reference cr_module.py:81; torchmetrics >= 0.11.4,
    requirements.txt:4, not installed here — SURVEY.md §8c "[external]"): ONE ROC curve over all pairs.
    Format step: if any pred lies outside [0, 1] all preds go through ``sigmoid``.  Curve: preds sorted
    descending, one point per DISTINCT threshold (cumulative tps / fps), origin prepended, trapezoid area;
    no positive or no negative -> 0.  The trapezoid is evaluated in float64 here (torchmetrics uses float32
    ratios, i.e. agrees to ~1e-7).  Also returns the exact integers (2U, P, N) of the equivalent
    Mann-Whitney form, U = #(pos > neg) + #(pos == neg) / 2."""
    p = scores.detach().reshape(-1).to(torch.float32)
    t = (target.detach().reshape(-1) > 0.5).to(torch.int64)
    if sigmoid_rule and not bool(torch.all((p >= 0) & (p <= 1))):
        p = p.sigmoid()
    order = torch.argsort(p, descending=True, stable=True)
    p, t = p[order], t[order]
    last = torch.cat([torch.nonzero(p[1:] - p[:-1]).reshape(-1), torch.tensor([p.numel() - 1])])   # end of each tie run
    tps = torch.cumsum(t, 0)[last]
    fps = 1 + last - tps
    n_pos, n_neg = int(tps[-1]), int(fps[-1])
    tps0 = torch.cat([torch.zeros(1, dtype=torch.int64), tps])
    fps0 = torch.cat([torch.zeros(1, dtype=torch.int64), fps])
    twice_u = int(((fps0[1:] - fps0[:-1]) * (tps0[1:] + tps0[:-1])).sum())        # trapezoid, scaled by 2 P N
    if n_pos == 0 or n_neg == 0:
        return 0.0, (twice_u, n_pos, n_neg)
    return twice_u / (2.0 * n_pos * n_neg), (twice_u, n_pos, n_neg)


def diversity_at_k(scores: Tensor, cand_aspect: Tensor, cand_off: Sequence[int], num_classes: int, k: int) -> Tensor:
    """Diversity@k per impression: reference manner/metrics/functional.py:8-28 inside the grouping of
    torchmetrics RetrievalMetric.compute (``if not mini_target.sum()`` -> 0, empty_target_action='neg')."""
    out = []
    for i in range(len(cand_off) - 1):
        s, a = scores[cand_off[i]:cand_off[i + 1]], cand_aspect[cand_off[i]:cand_off[i + 1]].long()
        if not a.sum():
            out.append(0.0)
            continue
        top = a[torch.argsort(s, dim=-1, descending=True, stable=True)][:k]            # functional.py:19
        cnt = F.pad(torch.bincount(top), pad=(0, num_classes - int(top.max()) - 1))     # :20-21
        prob = cnt / cnt.shape[0]                                                        # :22 (divides by num_classes)
        ent = torch.distributions.Categorical(prob).entropy()                            # :23 (renormalises)
        out.append(float(ent / torch.log(torch.tensor(float(num_classes)))))             # :25-27
    return torch.tensor(out, dtype=torch.float64)


def personalization_at_k(scores: Tensor, cand_aspect: Tensor, hist_aspect: Tensor, cand_off: Sequence[int],
                         hist_off: Sequence[int], num_classes: int, k: int) -> Tensor:
    """Personalization@k per impression: reference manner/metrics/functional.py:31-70 inside the grouping of
    manner/metrics/base.py:92-129 (same empty-target rule)."""
    out = []
    for i in range(len(cand_off) - 1):
        s, a = scores[cand_off[i]:cand_off[i + 1]], cand_aspect[cand_off[i]:cand_off[i + 1]].long()
        h = hist_aspect[hist_off[i]:hist_off[i + 1]].long()
        if not a.sum():
            out.append(0.0)
            continue
        top = a[torch.argsort(s, dim=-1, descending=True, stable=True)][:k]
        pc = torch.bincount(top, minlength=num_classes)
        tc = torch.bincount(h, minlength=num_classes)
        out.append(float(torch.min(pc, tc).sum() / torch.max(pc, tc).sum()))            # generalized_jaccard :65-70
    return torch.tensor(out, dtype=torch.float64)


# --------------------------------------------------------------------------- whole-path drivers

def offsets_to_batch(offsets: Sequence[int]) -> Tensor:
    """_make_batch_assignees (reference mind_rec_dataset.py:171-174) from CSR offsets."""
    sizes = torch.tensor([offsets[i + 1] - offsets[i] for i in range(len(offsets) - 1)])
    return torch.repeat_interleave(torch.arange(len(sizes)), sizes)


def reference_faithful_scores(ids: Tensor, mask: Tensor, hist_idx: Tensor, hist_off: Sequence[int],
                              cand_idx: Tensor, cand_off: Sequence[int], w, cfg,
                              chunk: int = 64) -> Tensor:
    """Mode R: every history and candidate OCCURRENCE is encoded, as
    cr_module.py:107,113 does, then late-fusion mean + dot.  ``ids/mask`` is the news pool
    [Nn, Lp]; ``*_idx`` index into it.  Returns ragged scores [sum c_i]."""

    def enc(idx):
        outs = []
        for s in range(0, idx.numel(), chunk):
            j = idx[s:s + chunk]
            m = mask[j]
            lp = int(m.sum(dim=1).max())           # tokenizer pads to the batch max (mind_rec_dataset.py:136)
            outs.append(encode_cls(ids[j][:, :lp], m[:, :lp], w, cfg))
        return torch.cat(outs)

    ids, mask = _t(ids), _t(mask)
    hv, cv = enc(_t(hist_idx).long()), enc(_t(cand_idx).long())
    bh, bc = offsets_to_batch(hist_off), offsets_to_batch(cand_off)
    return ragged(cr_scores(hv, bh, cv, bc, late_fusion=True), bc)


# ---------------------------------------------------------------------------------------------- collate
def make_batch_assignees(items: Sequence[Sequence]) -> Tensor:
    """reference manner/data/components/mind_rec_dataset.py:171-174"""
    sizes = torch.tensor([len(x) for x in items], dtype=torch.int64)
    return torch.repeat_interleave(torch.arange(len(items)), sizes)


def load_behaviors_frame(path: str, uid2index=None):
    """The reference's two read paths for a behaviours file, restated with pandas (mind_dataframe.py:278-288 for
    the cached frame, :291-357 for raw behaviors.tsv).  Returns columns user / history / candidates / labels."""
    import pandas as pd
    with open(path) as f:
        first = f.readline().rstrip("\n").split("\t")
    if first[:4] == ["user", "history", "candidates", "labels"]:
        return pd.read_table(path, converters={
            "history": lambda x: x.strip("[]").replace("'", "").split(", "),
            "candidates": lambda x: x.strip("[]").replace("'", "").split(", "),
            "labels": lambda x: list(map(int, x.strip("[]").split(", ")))})
    names = ["impid", "uid", "time", "history", "impressions"]
    b = pd.read_table(path, header=None, names=names, usecols=range(len(names)))
    b["history"] = b["history"].fillna("").str.split()
    b["impressions"] = b["impressions"].str.split()
    b["candidates"] = b["impressions"].apply(lambda x: [i.split("-")[0] for i in x])
    b["labels"] = b["impressions"].apply(lambda x: [int(i.split("-")[1]) for i in x])
    b = b[b["history"].apply(len) > 0].reset_index(drop=True)
    b["user"] = b["uid"].apply(lambda x: (uid2index or {}).get(x, 0))
    return b[["user", "history", "candidates", "labels"]]


def collate(news: dict, batch_rows: Sequence[dict], max_history_length: int, pad_id: int) -> dict:
    """MINDRecDatasetTest.__getitem__ + MINDCollate.__call__ (mind_rec_dataset.py:87-99, 114-168) with the tokenizer
    replaced by its stored output: ``news[nid] = {"tokens": [...], "entities": [...], "category", "sentiment",
    "sentiment_score"}``; ``batch_rows`` = behaviour rows {"user", "history", "candidates", "labels"}.
    Padding follows tokenizer(padding=True): to the longest news of the concatenated frame; entities are
    right-padded with 0 to the longest list (F.pad, :141-143)."""
    def side(id_lists):
        flat = [news[n] for ids in id_lists for n in ids]
        lp = max((len(r["tokens"]) for r in flat), default=0)
        ids = torch.full((len(flat), lp), pad_id, dtype=torch.int64)
        mask = torch.zeros((len(flat), lp), dtype=torch.int64)
        for i, r in enumerate(flat):
            ids[i, :len(r["tokens"])] = torch.tensor(r["tokens"], dtype=torch.int64)
            mask[i, :len(r["tokens"])] = 1
        width = max((len(r["entities"]) for r in flat), default=0)
        ent = torch.zeros((len(flat), width), dtype=torch.int64)
        for i, r in enumerate(flat):
            ent[i, :len(r["entities"])] = torch.tensor(r["entities"], dtype=torch.int64)
        return {"text": {"input_ids": ids, "attention_mask": mask}, "entities": ent,
                "category": torch.tensor([r["category"] for r in flat], dtype=torch.int64),
                "sentiment": torch.tensor([r["sentiment"] for r in flat], dtype=torch.int64),
                "sentiment_score": torch.tensor([r["sentiment_score"] for r in flat], dtype=torch.float32)}
    hist = [list(r["history"])[:max_history_length] for r in batch_rows]
    cand = [list(r["candidates"]) for r in batch_rows]
    return {"batch_hist": make_batch_assignees(hist), "batch_cand": make_batch_assignees(cand),
            "x_hist": side(hist), "x_cand": side(cand),
            "labels": torch.tensor([l for r in batch_rows for l in r["labels"]], dtype=torch.float32),
            "users": torch.tensor([int(r["user"]) for r in batch_rows], dtype=torch.int64)}


# ---------------------------------------------------------------------------------------------- evaluation loss
def model_step_loss(scores: Tensor, labels: Tensor, offsets: Sequence[int], supcon: bool, temperature: float = 0.1):
    """The loss of CRModule.model_step restated on dense matrices exactly as the reference builds them
    (cr_module.py:140-171): to_dense_batch of the labels, positive / negative index lists (the negatives are the first
    c_i - n_pos zero-label columns, i.e. the real negatives), then the reference's SupConLoss on the SCORE matrix
    (manner/models/components/losses.py:12-40) or nn.CrossEntropyLoss with probability targets.
    pytorch_metric_learning is not installed: its pieces are restated — mat-based pair loss masks
    (pos_mask[a1, p] = 1, neg_mask[a2, n] = 1), lmu.logsumexp with keep_mask (masked entries -> -inf, rows without any
    kept entry -> 0), c_f.small_val = finfo.tiny, default reducer of SupConLoss = AvgNonZeroReducer (mean of the
    per-row losses that are > 0; 0 if none).  Returns (batch loss, per-impression losses)."""
    sizes = [offsets[i + 1] - offsets[i] for i in range(len(offsets) - 1)]
    seg = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    mat, mask = to_dense_batch(scores.reshape(-1, 1), seg)
    mat = mat.squeeze(-1)
    y_true, _ = to_dense_batch(labels.reshape(-1, 1), seg)
    y_true = y_true.squeeze(-1)
    if not supcon:
        per = -(y_true * F.log_softmax(mat, dim=1)).sum(1)
        return F.cross_entropy(mat, y_true), per
    pos_mask, neg_mask = torch.zeros_like(mat), torch.zeros_like(mat)
    n_pairs = [0, 0]
    for i in range(mat.shape[0]):
        pos = torch.where(y_true[i])[0]                                                     # cr_module.py:146
        neg = torch.where(~y_true[i].bool())[0][: int(mask[i].sum()) - len(pos)]            # :152-157
        pos_mask[i, pos] = 1
        neg_mask[i, neg] = 1
        n_pairs[0] += len(pos)
        n_pairs[1] += len(neg)
    zero = torch.zeros(mat.shape[0])
    if all(n <= 1 for n in n_pairs) or not (pos_mask.bool().any() and neg_mask.bool().any()):   # losses.py:14-15, 21, 40
        return torch.tensor(0.0), zero
    m = mat / temperature
    m = m - m.max(dim=1, keepdim=True)[0]
    keep = (pos_mask + neg_mask).bool()
    den = torch.logsumexp(m.masked_fill(~keep, float("-inf")), dim=1, keepdim=True)
    den = den.masked_fill(~keep.any(dim=1, keepdim=True), 0)
    log_prob = m - den
    per = -((pos_mask * log_prob).sum(1) / (pos_mask.sum(1) + torch.finfo(m.dtype).tiny))
    nz = per > 0
    return (per[nz].mean() if bool(nz.any()) else per.sum() * 0), per


def supcon_embedding_loss(embeddings: Tensor, labels: Tensor, temperature: float = 0.1):
    """The A-Module's criterion (reference a_module.py:73-75,102-108): pytorch_metric_learning.losses.SupConLoss(temperature,
    distance=DotProductSimilarity(normalize_embeddings=False)) called as criterion(embeddings, labels).  The library is not
    installed here (requirements.txt:6 pins >= 2.1.1): restated from its generic-pair-loss path — all pairs from the labels
    (positives: same label, the anchor itself excluded; negatives: different label), mat = E E^T, then the very
    `_compute_loss` the reference copies into manner/models/components/losses.py:21-40 (temperature, row maximum over the
    whole row detached, logsumexp over positives + negatives, mean positive log-probability) and the loss's default
    AvgNonZeroReducer.  Differentiable.  Returns (batch loss, per-anchor losses)."""
    e, y = _t(embeddings), _t(labels).long()
    n = e.shape[0]
    same = y[:, None] == y[None, :]
    eye = torch.eye(n, dtype=torch.bool)
    pos_mask, neg_mask = (same & ~eye).float(), (~same).float()
    if not (pos_mask.bool().any() and neg_mask.bool().any()):
        return e.sum() * 0, torch.zeros(n)
    m = (e @ e.T) / temperature
    m = m - m.max(dim=1, keepdim=True)[0].detach()
    keep = (pos_mask + neg_mask).bool()
    den = torch.logsumexp(m.masked_fill(~keep, float("-inf")), dim=1, keepdim=True)
    den = den.masked_fill(~keep.any(dim=1, keepdim=True), 0)
    per = -((pos_mask * (m - den)).sum(1) / (pos_mask.sum(1) + torch.finfo(m.dtype).tiny))
    nz = per > 0
    return (per[nz].mean() if bool(nz.any()) else per.sum() * 0), per
