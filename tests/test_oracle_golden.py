"""The CPU oracle against vectors produced by the reference itself (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

import manner_oracle as O
from manner_amd.config import PRESETS
from manner_amd.weights import (make_additive_attention_weights, make_plm_weights, plm_param_shapes,
                                tensor_sha256)


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return z, json.loads(str(z["meta"]))


GOLDEN_ENC = ["enc_tiny_bert", "enc_tiny_roberta", "enc_tiny_distilbert", "enc_bert_base", "enc_bert_base_spread", "enc_roberta_base",
              # round 2: 64 news incl. lengths 2 and 96; [title, abstract] pair inputs through a real tokenizer call (Q4)
              "enc_bert_base_64", "enc_roberta_base_64", "enc_pair_bert_base", "enc_pair_tiny_bert",
              # round 3: BASELINE configs[4] at its FULL architecture — roberta-large, 24 layers, H = 1024, 16 heads; lengths {2, 33, 96}
              "enc_roberta_large"]


@pytest.mark.parametrize("name", GOLDEN_ENC)
def test_encoder_matches_reference(golden_dir, name):
    z, meta = _load(golden_dir, name)
    cfg = PRESETS[meta["preset"]]
    w = make_plm_weights(cfg, seed=meta["seed"], std=meta["std"])
    for k, h in meta.get("sha256", {}).items():   # the seeded weights regenerate bit-identically
        assert tensor_sha256(w[k]) == h, k
    out = O.encode_cls(z["ids"], z["mask"], w, cfg).numpy()
    assert out.shape == z["out"].shape
    # same arithmetic, different op order/threads: fp32 rounding only
    assert np.abs(out - z["out"]).max() < 2e-5


def test_encoder_padding_invariance(golden_dir):
    """Q5: the CLS output does not depend on how far the batch is padded (what makes
    varlen packing in the HIP path legitimate)."""
    z, meta = _load(golden_dir, "enc_tiny_bert")
    cfg = PRESETS[meta["preset"]]
    w = make_plm_weights(cfg, seed=meta["seed"], std=meta["std"])
    ids = np.pad(z["ids"], ((0, 0), (0, 17)), constant_values=cfg.pad_id)
    mask = np.pad(z["mask"], ((0, 0), (0, 17)))
    out = O.encode_cls(ids, mask, w, cfg).numpy()
    assert np.abs(out - z["out"]).max() < 2e-5


def test_additive_attention_matches_reference(golden_dir):
    z, meta = _load(golden_dir, "additive_attention")
    aw = make_additive_attention_weights(meta["input_dim"], meta["query_dim"], seed=meta["seed"])
    p = [aw["additive_attention." + k] for k in ("linear.weight", "linear.bias", "query")]
    assert np.abs(O.additive_attention(z["x"], *p).numpy() - z["out"]).max() < 1e-5
    assert np.abs(O.additive_attention(z["x1"], *p).numpy() - z["out1"]).max() < 1e-6


def test_dot_product_matches_reference(golden_dir):
    z, _ = _load(golden_dir, "dot_product")
    assert np.abs(O.dot_product(z["user"], z["cand"]).numpy() - z["out"]).max() < 1e-4


def test_entity_branch_matches_reference(golden_dir):
    """K8 incl. quirk Q1 (attention across the news of the batch) against the reference's own
    MannerNewsEncoder(use_entities=True)."""
    from manner_amd.weights import make_entity_weights
    z, meta = _load(golden_dir, "entities")
    cfg = PRESETS[meta["preset"]]
    w = make_plm_weights(cfg, seed=meta["seed"], std=meta["std"])
    ew = make_entity_weights(meta["n_entities"], 100, meta["query_dim"], cfg.hidden, seed=meta["seed"])
    sub = {k[len("entity_encoder."):]: v for k, v in ew.items() if k.startswith("entity_encoder.")}
    ent = O.entity_encoder(z["entities"], sub, meta["heads"])
    assert np.abs(ent.numpy() - z["entity_vec"]).max() < 1e-5
    text = O.encode_cls(z["ids"], z["mask"], w, cfg)
    out = O.news_encoder_with_entities(text, ent, ew["linear.weight"], ew["linear.bias"]).numpy()
    assert np.abs(out - z["out"]).max() < 2e-5
    # Q1 negative: the first news encoded alone is NOT its row of the batch (entity attention mixes news)
    alone = O.news_encoder_with_entities(text[:1], O.entity_encoder(z["entities"][:1], sub, meta["heads"]),
                                         ew["linear.weight"], ew["linear.bias"]).numpy()
    assert np.abs(alone - z["single0"]).max() < 2e-5 and np.abs(alone[0] - z["out"][0]).max() > 1e-3


def test_state_dict_keys_match_reference(golden_dir):
    """Our weight naming is the reference checkpoint naming."""
    with open(os.path.join(golden_dir, "state_dict_keys.json")) as f:
        keys = json.load(f)
    for preset in ("tiny-bert", "bert-base-uncased", "tiny-distilbert"):
        ours = sorted("text_encoder.plm_model." + n for n, _ in plm_param_shapes(PRESETS[preset]))
        assert ours == keys[preset]
    assert keys["user_encoder"] == sorted(
        ["additive_attention.linear.bias", "additive_attention.linear.weight", "additive_attention.query"])


def test_to_dense_batch_semantics():
    x = torch.arange(10, dtype=torch.float32).view(5, 2)
    batch = torch.tensor([0, 0, 2, 2, 2])           # segment 1 is empty
    dense, mask = O.to_dense_batch(x, batch)
    assert dense.shape == (3, 3, 2) and mask.tolist() == [[True, True, False], [False] * 3, [True] * 3]
    assert torch.equal(dense[0, :2], x[:2]) and torch.equal(dense[2], x[2:]) and dense[1].abs().sum() == 0


def test_pipeline_regression(golden_dir):
    z, _ = _load(golden_dir, "pipeline")
    tables = [torch.from_numpy(t) for t in z["tables"]]
    hi, ci = torch.from_numpy(z["hist_idx"]).long(), torch.from_numpy(z["cand_idx"]).long()
    ho, co = z["hist_off"].tolist(), z["cand_off"].tolist()
    bh, bc = O.offsets_to_batch(ho), O.offsets_to_batch(co)
    vecs = [(t[hi], t[ci]) for t in tables]
    late = O.cr_scores(vecs[0][0], bh, vecs[0][1], bc)
    assert np.array_equal(O.ragged(late, bc).numpy(), z["late"])
    # hand check of late fusion for impression 0: mean of history rows, dot with candidates
    u0 = tables[0][hi[ho[0]:ho[1]]].mean(0)
    ref0 = tables[0][ci[co[0]:co[1]]] @ u0
    assert np.abs(ref0.numpy() - z["late"][co[0]:co[1]]).max() < 1e-5
    ue = (z["ue_w"], z["ue_b"], z["ue_q"])
    early = O.cr_scores(vecs[0][0], bh, vecs[0][1], bc, late_fusion=False, user_encoder=ue)
    assert np.allclose(O.ragged(early, bc).numpy(), z["early"], atol=1e-6)
    ens = O.ragged(O.ensemble_scores(vecs, bh, bc, (-0.3, 0.2)), bc)
    assert np.allclose(ens.numpy(), z["ens_-0.3_0.2"], atol=1e-6)
    # z-score hand check: CR-only ensemble row has mean 0 / unbiased std 1 per impression
    e0 = z["ens_0_0"][co[1]:co[2]]
    assert abs(e0.mean()) < 1e-5 and abs(e0.std(ddof=1) - 1) < 1e-5
    n10, per10 = O.ndcg_at_k(ens, torch.from_numpy(z["labels"]), co, 10)
    assert abs(n10 - float(z["ndcg10"])) < 1e-12
    top = O.topk_indices(ens, co, 10)
    for i, t in enumerate(top):
        assert t == [v for v in z["top10"][i].tolist() if v >= 0]


def test_ndcg_hand_example():
    scores = torch.tensor([0.1, 0.9, 0.5, 0.3, 0.2, 0.8])
    target = torch.tensor([0.0, 1.0, 0.0, 1.0, 0.0, 0.0])
    off = [0, 3, 6]
    val, per = O.ndcg_at_k(scores, target, off, 10)
    # imp 0: positive ranked 1st -> 1.0 ; imp 1: order (0.8, 0.3, 0.2) -> positive at rank 2
    assert abs(per[0] - 1.0) < 1e-12 and abs(per[1] - 1 / np.log2(3)) < 1e-12
    assert abs(val - per.mean().item()) < 1e-12
    val0, _ = O.ndcg_at_k(scores, torch.zeros(6), off, 10)
    assert val0 == 0.0


def test_mrr_and_aspect_metrics_hand_examples():
    """Restated metrics against values worked out by hand from reference metrics/functional.py."""
    scores = torch.tensor([0.1, 0.9, 0.5, 0.3, 0.2, 0.8, 0.7])
    target = torch.tensor([0.0, 0.0, 1.0, 0.0, 1.0, 0.0, 0.0])
    off = [0, 3, 7]
    val, per = O.mrr(scores, target, off)
    assert per.tolist() == [0.5, 0.25] and abs(val - 0.375) < 1e-12      # positives ranked 2nd and 4th
    aspects = torch.tensor([2, 1, 1, 0, 3, 3, 1])
    # impression 0, k=2: top-2 = items 1, 2 -> classes {1, 1}: entropy 0 ; impression 1, k=3: top-3 = items
    # 5, 6, 3 -> classes {3, 1, 0}: uniform over 3 of 4 classes -> ln3 / ln4
    d = O.diversity_at_k(scores, aspects, off, 4, 2)
    assert abs(d[0]) < 1e-6               # Categorical clamps probabilities at eps: not exactly 0
    d3 = O.diversity_at_k(scores, aspects, off, 4, 3)
    assert abs(d3[1] - np.log(3) / np.log(4)) < 1e-6
    hist = torch.tensor([1, 1, 2, 3, 3, 3, 0])
    hoff = [0, 3, 7]
    # impression 1, k=3: predicted counts [1,1,0,1], history counts [1,0,0,3] -> min 2 / max 5
    p = O.personalization_at_k(scores, aspects, hist, off, hoff, 4, 3)
    assert abs(p[1] - 2 / 5) < 1e-7              # float32 division in the reference functional
    # impression 0, k=3: predicted [0,2,1,0] vs history [0,2,1,0] -> 1.0
    assert abs(p[0] - 1.0) < 1e-7
    # quirk: candidate class ids that sum to 0 (all class 0) count as "no target" -> 0
    z = torch.zeros(7, dtype=torch.long)
    assert O.diversity_at_k(scores, z, off, 4, 3).tolist() == [0.0, 0.0]
    assert O.personalization_at_k(scores, z, hist, off, hoff, 4, 3).tolist() == [0.0, 0.0]


def test_binary_auroc_against_sklearn_and_brute_force():
    """The AUROC restatement is pinned by an independent implementation (scikit-learn's roc_auc_score: same
    trapezoid through distinct thresholds) and by the O(n^2) Mann-Whitney count, with heavy ties."""
    from sklearn.metrics import roc_auc_score
    g = np.random.Generator(np.random.PCG64(11))
    for n, levels in ((50, 7), (400, 40), (3000, 0)):
        s = g.random(n).astype(np.float32)
        if levels:
            s = (np.floor(s * levels) / levels).astype(np.float32)           # many exact ties, all inside [0, 1]
        y = (g.random(n) < 0.2).astype(np.float32)
        y[0], y[1] = 1.0, 0.0
        auc, (u2, p, q) = O.binary_auroc(torch.from_numpy(s), torch.from_numpy(y))
        pos, neg = s[y > 0.5], s[y <= 0.5]
        brute = 2 * int((pos[:, None] > neg[None, :]).sum()) + int((pos[:, None] == neg[None, :]).sum())
        assert (u2, p, q) == (brute, len(pos), len(neg))
        assert abs(auc - roc_auc_score(y, s)) < 1e-12
    # format step: scores outside [0, 1] are squashed first, so everything above ~17 ties at 1.0
    s = torch.tensor([30.0, 25.0, 18.0, 2.0, -1.0, -40.0])
    y = torch.tensor([0.0, 1.0, 1.0, 0.0, 1.0, 0.0])
    auc_sig, (u2, p, q) = O.binary_auroc(s, y)
    auc_raw, _ = O.binary_auroc(s, y, sigmoid_rule=False)
    assert (p, q) == (3, 3) and u2 == 2 * (1 + 1 + 1 + 1 + 1) + 2 and abs(auc_raw - 5 / 9) < 1e-12 and auc_sig == u2 / 18
    assert O.binary_auroc(torch.tensor([0.3, 0.6]), torch.tensor([1.0, 1.0]))[0] == 0.0


@pytest.mark.parametrize("name", ["hidden_tiny_bert", "hidden_bert_base"])
def test_hidden_states_match_reference(golden_dir, name):
    """Oracle hidden_states[k] (embedding output, the frozen/trainable boundary, last_hidden_state) vs HF's, taken
    from the reference's own text encoder (real tokens only)."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    cfg = PRESETS[meta["preset"]]
    w = make_plm_weights(cfg, seed=meta["seed"], std=meta["std"])
    keep = torch.from_numpy(z["mask"]).bool()
    for k in meta["layers"]:
        h = O.encode_tokens(z["ids"], z["mask"], w, cfg, layers=k)[keep].numpy()
        assert np.abs(h - z[f"h{k}"]).max() < 2e-5, k


# ------------------------------------------------------------------------------------------------------------------
# Independent pins of the restated (un-importable) pieces — VERDICT r1 item 5.  torchmetrics / torch_geometric are not
# installed; scikit-learn and torch.nn.utils.rnn are, and implement the same definitions independently.

def test_ndcg_against_sklearn():
    """RetrievalNormalizedDCG(top_k=k) restatement vs sklearn.metrics.ndcg_score(k=k) on tie-free scores: same
    definition (linear gains, log2 discounts, IDCG from the sorted targets); an impression without a positive is 0
    in both the restatement (empty_target_action='neg') and here (handled explicitly: sklearn returns 0 too)."""
    from sklearn.metrics import ndcg_score
    g = np.random.Generator(np.random.PCG64(5))
    sizes = [2, 3, 5, 10, 11, 37, 120, 300]
    off = np.concatenate([[0], np.cumsum(sizes)])
    scores = g.permutation(int(off[-1])).astype(np.float32) / 7.0              # distinct values: tie-free
    for graded in (False, True):
        tgt = (g.random(int(off[-1])) < 0.15).astype(np.float32)
        if graded:
            tgt *= g.integers(1, 4, tgt.shape[0]).astype(np.float32)
        tgt[off[2]:off[3]] = 0.0                                               # one impression without a positive
        for k in (5, 10):
            mean, per = O.ndcg_at_k(torch.from_numpy(scores), torch.from_numpy(tgt), off.tolist(), k)
            want = [ndcg_score(tgt[None, a:b], scores[None, a:b], k=k) if tgt[a:b].sum() > 0 else 0.0
                    for a, b in zip(off[:-1], off[1:])]
            assert np.abs(per.numpy() - np.asarray(want)).max() < 1e-12
            assert abs(mean - float(np.mean(want))) < 1e-12


def test_mrr_against_brute_force():
    """RetrievalMRR restatement vs a literal search: rank every candidate by counting the ones that beat it."""
    g = np.random.Generator(np.random.PCG64(6))
    sizes = [1, 2, 3, 8, 40, 300]
    off = np.concatenate([[0], np.cumsum(sizes)])
    scores = g.permutation(int(off[-1])).astype(np.float32)
    tgt = (g.random(int(off[-1])) < 0.1).astype(np.float32)
    tgt[off[3]:off[4]] = 0.0
    _, per = O.mrr(torch.from_numpy(scores), torch.from_numpy(tgt), off.tolist())
    want = []
    for a, b in zip(off[:-1], off[1:]):
        best = None
        for j in range(a, b):
            if tgt[j] > 0:
                rank = 1 + sum(1 for i in range(a, b) if scores[i] > scores[j])
                best = rank if best is None else min(best, rank)
        want.append(0.0 if best is None else 1.0 / best)
    assert np.abs(per.numpy() - np.asarray(want)).max() < 1e-12


def test_to_dense_batch_against_pad_sequence():
    """to_dense_batch restatement vs torch.nn.utils.rnn.pad_sequence over the per-segment slices (both zero-pad to the
    longest segment, keep the original order inside a segment)."""
    from torch.nn.utils.rnn import pad_sequence
    g = np.random.Generator(np.random.PCG64(7))
    sizes = [3, 1, 7, 2, 5]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))      # _make_batch_assignees
    for inner in ((), (4,), (2, 3)):
        x = torch.from_numpy(g.standard_normal((sum(sizes),) + inner).astype(np.float32))
        dense, mask = O.to_dense_batch(x, batch)
        want = pad_sequence(list(torch.split(x, sizes)), batch_first=True)
        assert torch.equal(dense, want)
        assert torch.equal(mask, pad_sequence([torch.ones(s, dtype=torch.bool) for s in sizes], batch_first=True))


def test_zscore_against_numpy():
    """K13 restatement vs a direct numpy evaluation of ensemble_module.py:138-149 on a ragged batch: mean over the
    zero-padded row divided by c_i, unbiased std of the valid entries, the WHOLE padded row normalised."""
    g = np.random.Generator(np.random.PCG64(8))
    sizes = [4, 9, 2, 6]
    batch = torch.repeat_interleave(torch.arange(4), torch.tensor(sizes))
    flat = torch.from_numpy(g.standard_normal(sum(sizes)).astype(np.float32) * 3 + 1)
    dense, mask = O.to_dense_batch(flat, batch)
    z = O.zscore(dense, mask).numpy()
    for i, c in enumerate(sizes):
        row = dense[i].numpy().astype(np.float64)
        mean, std = row.sum() / c, row[:c].std(ddof=1)
        assert np.abs(z[i] - (row - mean) / std).max() < 1e-5
        assert c == max(sizes) or abs(z[i, c] - (-mean / std)) < 1e-5            # padded slot: not zero


def golden_train_case(golden_dir, name):
    """Inputs and reference gradients of a train-mode golden: (cfg, numpy weights, z, meta, {param name: expected grad})."""
    z, meta = _load(golden_dir, name)
    cfg = PRESETS[meta["preset"]]
    w = make_plm_weights(cfg, seed=meta["seed"], std=meta["std"], with_pooler=False)
    expect = {k[len("grad:"):]: z[k] for k in z.files if k.startswith("grad:")}
    return cfg, w, z, meta, expect


def compare_train_grads(grads, z, meta, expect, rel):
    """`grads`: {param name: numpy grad or None}.  Relative to each tensor's own largest reference entry."""
    for k in meta["frozen"]:
        assert grads.get(k) is None, f"{k} is frozen in the reference"
    for k, ref in expect.items():
        g = grads[k]
        assert g is not None, k
        if k == "embeddings.word_embeddings.weight":
            rows = z["word_rows"]
            rest = np.ones(g.shape[0], bool)
            rest[rows] = False
            assert np.abs(g[rest]).sum() == 0.0 == float(z["word_rest_abs_sum"])      # rows of ids that do not occur
            g = g[rows]
        elif g.ndim == 2 and meta["matrix_rows"] is not None and not k.startswith("embeddings."):
            g = g[:meta["matrix_rows"]]
        assert g.shape == ref.shape, k
        # floor: the key-bias gradient is analytically zero (softmax is shift-invariant), i.e. rounding noise of ~1e-8
        assert np.abs(g - ref).max() <= rel * max(np.abs(ref).max(), 1e-3), (k, float(np.abs(g - ref).max()), float(np.abs(ref).max()))


@pytest.mark.parametrize("name", ["train_tiny_bert", "train_tiny_roberta"])
def test_train_mode_gradients_match_reference(golden_dir, name):
    """autograd over the oracle's train-mode forward reproduces the gradients the reference's own
    MannerTextEncoder.train() produced (all dropout probabilities 0) — incl. which tensors `frozen_layers` leaves
    without a gradient while the activation gradient still reaches the embeddings through them."""
    cfg, w, z, meta, expect = golden_train_case(golden_dir, name)
    frozen = set(meta["frozen"])
    wt = {k: torch.from_numpy(v).requires_grad_(k not in frozen) for k, v in w.items()}
    out = O.encode_cls_train(z["ids"], z["mask"], wt, cfg)
    assert np.abs(out.detach().numpy() - z["out"]).max() < 2e-5
    (out * torch.from_numpy(z["R"])).sum().backward()
    compare_train_grads({k: (None if v.grad is None else v.grad.numpy()) for k, v in wt.items()}, z, meta, expect, rel=2e-4)


def test_train_mode_dropout_masks_are_replayable():
    """The oracle's dropout hook: a keep mask of ones with p > 0 is a pure rescale, and masks change the output."""
    cfg = PRESETS["tiny-bert"]
    w = {k: torch.from_numpy(v) for k, v in make_plm_weights(cfg, seed=3, std=0.05).items()}
    from manner_amd.synth import synth_news_tokens
    ids, mask = synth_news_tokens(3, cfg, seed=3, max_len=12)
    base = O.encode_cls_train(ids, mask, w, cfg)
    g = torch.Generator().manual_seed(0)

    def keep(site, kind):
        shape = {"rows": (3, 12, cfg.hidden), "attn": (3, cfg.heads, 12, 12), "cls": (3, cfg.hidden)}[kind]
        return (torch.rand(shape, generator=g) >= 0.1).float()

    dropped = O.encode_cls_train(ids, mask, w, cfg, p_hidden=0.1, p_attn=0.1, p_out=0.1, keep=keep)
    assert (dropped - base).abs().max() > 1e-3
    ones = O.encode_cls_train(ids, mask, w, cfg, p_out=0.5, keep=lambda s, k: torch.ones(3, cfg.hidden))
    assert torch.allclose(ones, base * 2.0, atol=1e-5)


def _baseline_weights(seed, dim, query_dim=200):
    from manner_amd.weights import make_mha_pool_weights
    mw = make_mha_pool_weights(dim, query_dim, seed=seed)
    mha = {k[len("multihead_attention."):]: v for k, v in mw.items() if k.startswith("multihead_attention.")}
    pool = tuple(mw["additive_attention." + k] for k in ("linear.weight", "linear.bias", "query"))
    return mw, mha, pool


def test_baseline_encoders_match_reference(golden_dir):
    """PLMTextEncoder (hidden states AT PADDED POSITIONS flow through the un-masked axis-0 attention and the
    un-masked pooler) and NRMSUserEncoder (attention across the users of the batch) against the reference's own outputs."""
    z, meta = _load(golden_dir, "baselines")
    for tag, (preset, heads) in meta["plm"].items():
        cfg = PRESETS[preset]
        w = make_plm_weights(cfg, seed=meta["seed"], std=meta["std"])
        _, mha, pool = _baseline_weights(meta["seed"], cfg.hidden)
        out = O.plm_text_encoder(z[f"plm_{tag}_ids"], z[f"plm_{tag}_mask"], w, cfg, mha, pool, heads).numpy()
        assert np.abs(out - z[f"plm_{tag}_out"]).max() < 2e-5, tag
    for tag, (dim, heads) in meta["nrms"].items():
        _, mha, pool = _baseline_weights(meta["seed"] + 1, dim)
        out = O.nrms_user_encoder(z[f"nrms_{tag}_x"], mha, pool, heads).numpy()
        assert np.abs(out - z[f"nrms_{tag}_out"]).max() < 1e-5, tag


def golden_train_entities_case(golden_dir):
    from manner_amd.weights import make_entity_weights
    z, meta = _load(golden_dir, "train_entities")
    cfg = PRESETS[meta["preset"]]
    w = make_plm_weights(cfg, seed=meta["seed"], std=meta["std"], with_pooler=False)
    ew = make_entity_weights(meta["n_entities"], dim=100, query_dim=meta["query_dim"], hidden=cfg.hidden, seed=meta["seed"])
    expect = {k[len("grad:"):]: z[k] for k in z.files if k.startswith("grad:")}
    return cfg, w, ew, z, meta, expect


def test_entity_branch_training_gradients_match_reference(golden_dir):
    """The reference's default use_entities=True in train() mode: embedding (padding_idx 0), axis-0 attention, pooler, the
    linear on cat[text, entity] and the text encoder behind it — oracle autograd against the reference's own gradients."""
    cfg, w, ew, z, meta, expect = golden_train_entities_case(golden_dir)
    wt = {k: torch.from_numpy(v).requires_grad_("layer.0." not in k) for k, v in w.items()}
    we = {k: torch.from_numpy(v).requires_grad_(True) for k, v in ew.items()}
    out = O.news_encoder_train(z["ids"], z["mask"], z["entities"], wt, we, cfg, meta["heads"])
    assert np.abs(out.detach().numpy() - z["out"]).max() < 2e-5
    (out * torch.from_numpy(z["R"])).sum().backward()
    for k, ref in expect.items():
        g = (wt[k[len("text_encoder.plm_model."):]] if k.startswith("text_encoder.") else we[k]).grad.numpy()
        assert np.abs(g - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-3), k
    assert np.abs(we["entity_encoder.pretrained_embedding.weight"].grad.numpy()[0]).max() == 0.0      # padding_idx row
