#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference and transformers):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's own leaf modules (manner/models/components/
{news_encoder,attention,user_encoder,click_predictors}.py) and HF BertModel /
RobertaModel, feeds them seeded synthetic inputs and the seeded weights of
``manner_amd.weights``, and stores inputs + reference outputs as small .npz
files.  The reference never travels to the GPU box; these vectors do.

Fixtures whose ``source`` field says "oracle" restate Lightning-level code that
cannot be imported here (lightning / torch_geometric / torchmetrics are absent);
they pin regressions of the restatement, not the reference.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from manner_amd.config import ARCH_BERT, PRESETS  # noqa: E402
from manner_amd.synth import synth_news_tokens  # noqa: E402
from manner_amd.weights import (make_additive_attention_weights, make_entity_weights, make_plm_weights,  # noqa: E402
                                tensor_sha256)

from manner.models.components.attention import AdditiveAttention  # noqa: E402
from manner.models.components.click_predictors import DotProduct  # noqa: E402
from manner.models.components.news_encoder import MannerNewsEncoder  # noqa: E402
from manner.models.components.user_encoder import NAMLUserEncoder  # noqa: E402

torch.set_grad_enabled(False)


def hf_model_dir(cfg, weights, tmp, no_dropout=False):
    from transformers import BertConfig, BertModel, RobertaConfig, RobertaModel
    kw = dict(**({"hidden_dropout_prob": 0.0, "attention_probs_dropout_prob": 0.0} if no_dropout else {}), vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
              num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate,
              max_position_embeddings=cfg.max_pos, type_vocab_size=cfg.type_vocab,
              layer_norm_eps=cfg.ln_eps, pad_token_id=cfg.pad_id, hidden_act="gelu")
    if cfg.naming == "distilbert":
        from transformers import DistilBertConfig, DistilBertModel
        model = DistilBertModel(DistilBertConfig(vocab_size=cfg.vocab, dim=cfg.hidden, n_layers=cfg.layers, n_heads=cfg.heads,
                                                 hidden_dim=cfg.intermediate, max_position_embeddings=cfg.max_pos,
                                                 pad_token_id=cfg.pad_id, activation="gelu"))
    elif cfg.arch == ARCH_BERT:
        model = BertModel(BertConfig(**kw))
    else:
        model = RobertaModel(RobertaConfig(**kw))
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m or "token_type_ids" in m for m in missing), missing
    model.save_pretrained(tmp)
    return tmp


def reference_news_encoder(plm_dir, hidden):
    enc = MannerNewsEncoder(plm_model=plm_dir, frozen_layers=[0], dropout_probability=0.2,
                            use_entities=False, entity_embeddings=None, entity_embedding_dim=100,
                            num_attention_heads=10, query_vector_dim=200, text_embedding_dim=hidden)
    return enc.eval()


def gen_encoder(name, preset, n, lp, seed, std, lengths=None):
    cfg = PRESETS[preset]
    w = make_plm_weights(cfg, seed=seed, std=std)
    ids, mask = synth_news_tokens(n, cfg, seed=seed, max_len=lp, lengths=lengths)
    with tempfile.TemporaryDirectory() as tmp:
        enc = reference_news_encoder(hf_model_dir(cfg, w, tmp), cfg.hidden)
        from transformers import BatchEncoding
        news = {"text": BatchEncoding({"input_ids": torch.from_numpy(ids),
                                       "attention_mask": torch.from_numpy(mask)})}
        out = enc(news).numpy()
        keys = sorted(enc.state_dict().keys())
    np.savez_compressed(
        os.path.join(HERE, f"{name}.npz"), ids=ids, mask=mask, out=out,
        meta=json.dumps({"source": "reference MannerNewsEncoder (news_encoder.py:75-129) over HF "
                                   "transformers " + __import__("transformers").__version__,
                         "preset": preset, "seed": seed, "std": std,
                         "sha256": {k: tensor_sha256(w[k]) for k in
                                    ("embeddings.word_embeddings.weight",
                                     [n for n in w if n.endswith((".query.weight", ".q_lin.weight"))][0],
                                     [n for n in w if n.endswith((".output.dense.bias", ".ffn.lin2.bias"))][-1])}}))
    print(name, out.shape, float(np.abs(out).mean()))
    return keys


def gen_pair(name, preset, n, seed, std, max_length=96):
    """Two text aspects: the reference collate hands the tokenizer [title, abstract] PAIRS with
    return_token_type_ids=False, padding=True, truncation=True (mind_rec_dataset.py:134-137, 161-163; the tokenizer is
    built with model_max_length = tokenizer_max_length = 96, mind_rec_datamodule.py:52-56), so the PLM sees
    [CLS] title [SEP] abstract [SEP] with token type 0 everywhere.  No vocabulary is available offline: a synthetic
    WordPiece vocabulary of the preset's size (whole-word entries w<i>) drives a real BertTokenizerFast through exactly
    that call; the fixture keeps its ids / mask and the reference encoder's output on them."""
    from transformers import BatchEncoding, BertTokenizerFast
    cfg = PRESETS[preset]
    w = make_plm_weights(cfg, seed=seed, std=std)
    g = np.random.Generator(np.random.PCG64(seed))
    with tempfile.TemporaryDirectory() as tmp:
        special = {0: "[PAD]", 100: "[UNK]", 101: "[CLS]", 102: "[SEP]", 103: "[MASK]"} if cfg.vocab > 200 else \
                  {0: "[PAD]", 1: "[UNK]", 2: "[CLS]", 3: "[SEP]", 4: "[MASK]"}
        with open(os.path.join(tmp, "vocab.txt"), "w") as f:
            for i in range(cfg.vocab):
                f.write(special.get(i, f"w{i}") + "\n")
        tok = BertTokenizerFast(vocab_file=os.path.join(tmp, "vocab.txt"), do_lower_case=True, model_max_length=max_length)
        words = [i for i in range(cfg.vocab) if i not in special]
        def text(k):
            return " ".join(f"w{words[j]}" for j in g.integers(0, len(words), k))
        title_len = [3, 9, 14, 20, 11, 7, 30, 16][:n]
        abstr_len = [0, 12, 40, 120, 70, 1, 66, 75][:n]          # 0: empty abstract; 120: truncated at max_length
        pairs = [[text(a), text(b)] for a, b in zip(title_len, abstr_len)]
        enc_in = tok(pairs, return_tensors="pt", return_token_type_ids=False, padding=True, truncation=True)   # _tokenize
        assert "token_type_ids" not in enc_in
        ids, mask = enc_in["input_ids"].numpy(), enc_in["attention_mask"].numpy()
        sep = tok.sep_token_id
        assert ids.shape[1] == max_length and all((row == sep).sum() == 2 for row in ids)
        enc = reference_news_encoder(hf_model_dir(cfg, w, os.path.join(tmp, "plm")), cfg.hidden)
        out = enc({"text": BatchEncoding({"input_ids": enc_in["input_ids"], "attention_mask": enc_in["attention_mask"]})}).numpy()
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), ids=ids, mask=mask, out=out,
                        meta=json.dumps({"source": "reference MannerNewsEncoder on BertTokenizerFast([title, abstract] pairs, "
                                                   "return_token_type_ids=False, padding=True, truncation=True) — "
                                                   "mind_rec_dataset.py:134-137,161-163; synthetic vocabulary",
                                         "preset": preset, "seed": seed, "std": std, "sep_id": int(sep),
                                         "title_words": title_len, "abstract_words": abstr_len}))
    print(name, ids.shape, out.shape, mask.sum(1).tolist())


def gen_hidden(name, preset, n, lp, seed, std, lengths, layers_out):
    """hidden_states[k] of the reference's text encoder (the tensor HF hands from layer k-1 to layer k; k = 8 is the
    frozen / trainable boundary of configs/model/cr_module.yaml:10), real tokens only, packed news after news."""
    cfg = PRESETS[preset]
    w = make_plm_weights(cfg, seed=seed, std=std)
    ids, mask = synth_news_tokens(n, cfg, seed=seed, max_len=lp, lengths=lengths)
    with tempfile.TemporaryDirectory() as tmp:
        enc = reference_news_encoder(hf_model_dir(cfg, w, tmp), cfg.hidden)
        hs = enc.text_encoder.plm_model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                                        output_hidden_states=True).hidden_states
    keep = torch.from_numpy(mask).bool()
    packed = {f"h{k}": hs[k][keep].numpy() for k in layers_out}
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), ids=ids, mask=mask, **packed,
                        meta=json.dumps({"source": "HF hidden_states of the reference MannerTextEncoder.plm_model "
                                                   "(news_encoder.py:20), transformers " + __import__("transformers").__version__,
                                         "preset": preset, "seed": seed, "std": std, "layers": list(layers_out)}))
    print(name, {k: v.shape for k, v in packed.items()})


def gen_components(seed=42):
    g = np.random.Generator(np.random.PCG64(seed))
    d, q = 768, 200
    aw = make_additive_attention_weights(d, q, seed=seed)
    att = AdditiveAttention(d, q).eval()
    att.load_state_dict({k[len("additive_attention."):]: torch.from_numpy(v) for k, v in aw.items()})
    ue = NAMLUserEncoder(news_embedding_dim=d, query_vector_dim=q).eval()
    ue.load_state_dict({k: torch.from_numpy(v) for k, v in aw.items()})
    x = g.standard_normal((4, 30, d), dtype=np.float32)
    x[1, 12:] = 0.0                      # zero-padded history rows (Q2: padding changes the result)
    x[3, 1:] = 0.0
    out = att(torch.from_numpy(x)).numpy()
    out_ue = ue(torch.from_numpy(x)).numpy()
    assert np.array_equal(out, out_ue)
    x1 = g.standard_normal((3, 1, d), dtype=np.float32)   # S = 1
    out1 = att(torch.from_numpy(x1)).numpy()
    np.savez_compressed(os.path.join(HERE, "additive_attention.npz"), x=x, out=out, x1=x1, out1=out1,
                        meta=json.dumps({"source": "reference AdditiveAttention (attention.py:6-29) / "
                                                   "NAMLUserEncoder (user_encoder.py:9-21)", "seed": seed,
                                         "input_dim": d, "query_dim": q}))
    user = g.standard_normal((4, 1, d), dtype=np.float32)
    cand = g.standard_normal((4, d, 37), dtype=np.float32)
    sc = DotProduct()(torch.from_numpy(user), torch.from_numpy(cand)).numpy()
    np.savez_compressed(os.path.join(HERE, "dot_product.npz"), user=user, cand=cand, out=sc,
                        meta=json.dumps({"source": "reference DotProduct (click_predictors.py:5-12)"}))
    print("components", out.shape, sc.shape)
    return sorted(ue.state_dict().keys())


def gen_entities(seed=42):
    """Entity branch (K8) from the reference's own MannerNewsEncoder(use_entities=True) over tiny-bert,
    incl. the Q1 negative: the same news encoded alone differs from its row in the batch."""
    cfg = PRESETS["tiny-bert"]
    w = make_plm_weights(cfg, seed=seed, std=0.05)
    ew = make_entity_weights(60, dim=100, query_dim=200, hidden=cfg.hidden, seed=seed)
    ids, mask = synth_news_tokens(9, cfg, seed=seed, max_len=24)
    g = np.random.Generator(np.random.PCG64(seed + 1))
    ent = g.integers(1, 60, size=(9, 6), dtype=np.int64)
    for r, keep in enumerate((6, 5, 3, 1, 0, 2, 6, 4, 1)):        # right-padded with id 0 (reference collate)
        ent[r, keep:] = 0
    with tempfile.TemporaryDirectory() as tmp:
        enc = MannerNewsEncoder(plm_model=hf_model_dir(cfg, w, tmp), frozen_layers=[], dropout_probability=0.2,
                                use_entities=True, entity_embeddings=ew["entity_encoder.pretrained_embedding.weight"],
                                entity_embedding_dim=100, num_attention_heads=10, query_vector_dim=200,
                                text_embedding_dim=cfg.hidden).eval()
        missing, unexpected = enc.load_state_dict({k: torch.from_numpy(v) for k, v in ew.items()}, strict=False)
        assert not unexpected and all(m.startswith("text_encoder.") for m in missing), (missing, unexpected)
        from transformers import BatchEncoding

        def run(sel):
            return enc({"text": BatchEncoding({"input_ids": torch.from_numpy(ids[sel]), "attention_mask": torch.from_numpy(mask[sel])}),
                        "entities": torch.from_numpy(ent[sel])})

        allrows = np.arange(9)
        out = run(allrows).numpy()
        ent_only = enc.entity_encoder(torch.from_numpy(ent)).numpy()
        single = run(allrows[:1]).numpy()
        keys = sorted(enc.state_dict().keys())
    assert np.abs(single[0] - out[0]).max() > 1e-3            # Q1: batch-coupled
    np.savez_compressed(os.path.join(HERE, "entities.npz"), ids=ids, mask=mask, entities=ent, out=out, entity_vec=ent_only,
                        single0=single,
                        meta=json.dumps({"source": "reference MannerNewsEncoder(use_entities=True) / MannerEntityEncoder "
                                                   "(news_encoder.py:40-129)", "preset": "tiny-bert", "seed": seed, "std": 0.05,
                                         "n_entities": 60, "heads": 10, "query_dim": 200}))
    print("entities", out.shape, float(np.abs(single[0] - out[0]).max()))
    return keys


def gen_pipeline(seed=42):
    """Lightning-level restatement fixtures (source: oracle)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import manner_oracle as O
    from manner_amd.synth import synth_impressions
    g = np.random.Generator(np.random.PCG64(seed))
    nn_, d = 300, 64
    tables = [g.standard_normal((nn_, d), dtype=np.float32) for _ in range(3)]
    imp = synth_impressions(8, nn_, seed=seed, max_cand=40)
    hi, ho, ci, co, lab = imp["hist_idx"], imp["hist_off"], imp["cand_idx"], imp["cand_off"], imp["labels"]
    bh, bc = O.offsets_to_batch(ho.tolist()), O.offsets_to_batch(co.tolist())
    vecs = [(torch.from_numpy(t[hi]), torch.from_numpy(t[ci])) for t in tables]
    lf = O.cr_scores(vecs[0][0], bh, vecs[0][1], bc, late_fusion=True)
    aw = make_additive_attention_weights(d, 20, seed=seed)
    ue = tuple(torch.from_numpy(aw["additive_attention." + k]) for k in ("linear.weight", "linear.bias", "query"))
    ef = O.cr_scores(vecs[0][0], bh, vecs[0][1], bc, late_fusion=False, user_encoder=ue)
    ens = {}
    for wts in ((0.0, 0.0), (-0.3, 0.0), (-0.3, 0.2)):
        ens["ens_%g_%g" % wts] = O.ragged(O.ensemble_scores(vecs, bh, bc, wts), bc).numpy()
    flat = torch.from_numpy(ens["ens_-0.3_0.2"])
    n10, per10 = O.ndcg_at_k(flat, torch.from_numpy(lab), co.tolist(), 10)
    n5, per5 = O.ndcg_at_k(flat, torch.from_numpy(lab), co.tolist(), 5)
    top10 = O.topk_indices(flat, co.tolist(), 10)
    top10_arr = np.full((len(top10), 10), -1, dtype=np.int32)
    for i, t in enumerate(top10):
        top10_arr[i, :len(t)] = t
    np.savez_compressed(
        os.path.join(HERE, "pipeline.npz"), tables=np.stack(tables), hist_idx=hi, hist_off=ho,
        cand_idx=ci, cand_off=co, labels=lab, late=O.ragged(lf, bc).numpy(), early=O.ragged(ef, bc).numpy(),
        late_dense=lf.numpy(), ue_w=ue[0].numpy(), ue_b=ue[1].numpy(), ue_q=ue[2].numpy(),
        ndcg10=np.float64(n10), ndcg5=np.float64(n5), per10=per10.numpy(), per5=per5.numpy(),
        top10=top10_arr, **ens,
        meta=json.dumps({"source": "oracle restatement of cr_module.py:105-131, ensemble_module.py:95-151, "
                                   "RetrievalNormalizedDCG (not importable here)", "seed": seed}))
    print("pipeline", lf.shape, n10, n5)


def gen_train(name, preset, n, lp, seed, std, lengths, frozen_layers, matrix_rows=None):
    """gradients of the reference's MannerTextEncoder in train() mode (news_encoder.py:11-37: HF model,
    "layer.N." parameters of `frozen_layers` frozen, dropout on the [CLS] vector) with every dropout probability 0 (the
    only way to make train mode reproducible): loss = sum(out * R).  Matrices keep `matrix_rows` leading rows (None = all);
    the word-embedding gradient keeps the rows of the ids that occur and the sum of |.| of all other rows (0)."""
    from manner.models.components.news_encoder import MannerTextEncoder
    from transformers import BatchEncoding
    cfg = PRESETS[preset]
    w = make_plm_weights(cfg, seed=seed, std=std)
    ids, mask = synth_news_tokens(n, cfg, seed=seed, max_len=lp, lengths=lengths)
    R = np.random.default_rng(seed).standard_normal((n, cfg.hidden)).astype(np.float32)
    with tempfile.TemporaryDirectory() as tmp, torch.enable_grad():
        enc = MannerTextEncoder(plm_model=hf_model_dir(cfg, w, tmp, no_dropout=True), frozen_layers=list(frozen_layers),
                                dropout_probability=0.0).train()
        out = enc(BatchEncoding({"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}))
        (out * torch.from_numpy(R)).sum().backward()
        grads, frozen = {}, []
        for k, p in enc.plm_model.named_parameters():
            if k.startswith("pooler."):
                continue
            if p.grad is None:
                frozen.append(k)
                continue
            g = p.grad.numpy()
            if k == "embeddings.word_embeddings.weight":
                rows = np.unique(ids[mask > 0])
                rest = np.ones(g.shape[0], bool)
                rest[rows] = False
                grads["word_rows"], grads["word_rest_abs_sum"] = rows, np.float64(np.abs(g[rest]).sum())
                g = g[rows]
            elif g.ndim == 2 and matrix_rows is not None and not k.startswith("embeddings."):
                g = g[:matrix_rows]
            grads["grad:" + k] = np.ascontiguousarray(g)
    np.savez_compressed(
        os.path.join(HERE, f"{name}.npz"), ids=ids, mask=mask, R=R, out=out.detach().numpy(), **grads,
        meta=json.dumps({"source": "reference MannerTextEncoder.train() (news_encoder.py:11-37) over HF transformers "
                                   + __import__("transformers").__version__ + ", all dropout probabilities 0, loss = sum(out * R)",
                         "preset": preset, "seed": seed, "std": std, "frozen_layers": list(frozen_layers), "frozen": frozen,
                         "matrix_rows": matrix_rows}))
    print(name, out.shape, len(grads), "grad tensors,", len(frozen), "frozen")


def gen_train_entities(seed=54):
    """SURVEY §8f-3 with the reference's default use_entities=True: gradients of its own MannerNewsEncoder.train() (all dropout
    probabilities 0; loss = sum(out * R)) for the entity branch (embedding table with padding_idx 0, axis-0 attention, pooler), the
    `linear` on cat[text, entity], and — through it — a sample of the text encoder's tensors."""
    from transformers import BatchEncoding
    cfg = PRESETS["tiny-bert"]
    w = make_plm_weights(cfg, seed=seed, std=0.05)
    ew = make_entity_weights(60, dim=100, query_dim=200, hidden=cfg.hidden, seed=seed)
    ids, mask = synth_news_tokens(9, cfg, seed=seed, max_len=20)
    g = np.random.Generator(np.random.PCG64(seed + 1))
    ent = g.integers(1, 60, size=(9, 6), dtype=np.int64)
    for r, keep in enumerate((6, 5, 3, 1, 0, 2, 6, 4, 1)):
        ent[r, keep:] = 0
    R = np.random.default_rng(seed).standard_normal((9, cfg.hidden)).astype(np.float32)
    with tempfile.TemporaryDirectory() as tmp, torch.enable_grad():
        enc = MannerNewsEncoder(plm_model=hf_model_dir(cfg, w, tmp, no_dropout=True), frozen_layers=[0], dropout_probability=0.0,
                                use_entities=True, entity_embeddings=ew["entity_encoder.pretrained_embedding.weight"],
                                entity_embedding_dim=100, num_attention_heads=10, query_vector_dim=200,
                                text_embedding_dim=cfg.hidden).train()
        missing, unexpected = enc.load_state_dict({k: torch.from_numpy(v) for k, v in ew.items()}, strict=False)
        assert not unexpected and all(m.startswith("text_encoder.") for m in missing), (missing, unexpected)
        out = enc({"text": BatchEncoding({"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}),
                   "entities": torch.from_numpy(ent)})
        (out * torch.from_numpy(R)).sum().backward()
        grads = {}
        for k, p in enc.named_parameters():
            if k.startswith("text_encoder.") and k not in ("text_encoder.plm_model.encoder.layer.1.output.dense.bias",
                                                           "text_encoder.plm_model.encoder.layer.1.attention.self.query.bias",
                                                           "text_encoder.plm_model.embeddings.LayerNorm.weight",
                                                           "text_encoder.plm_model.embeddings.position_embeddings.weight"):
                continue
            assert p.grad is not None, k
            grads["grad:" + k] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "train_entities.npz"), ids=ids, mask=mask, entities=ent, R=R, out=out.detach().numpy(), **grads,
                        meta=json.dumps({"source": "reference MannerNewsEncoder(use_entities=True).train() (news_encoder.py:40-129), all dropout "
                                                   "probabilities 0, loss = sum(out * R), transformers " + __import__("transformers").__version__,
                                         "preset": "tiny-bert", "seed": seed, "std": 0.05, "n_entities": 60, "heads": 10, "query_dim": 200,
                                         "frozen_layers": [0]}))
    print("train_entities", out.shape, len(grads), "grad tensors")


def gen_baselines(seed=53):
    """the reference's PLMTextEncoder (news_encoder.py:132-171) and NRMSUserEncoder (user_encoder.py:24-42),
    eval mode, on padded inputs — both mix padded positions / padded history slots into the result."""
    from manner.models.components.news_encoder import PLMTextEncoder
    from manner.models.components.user_encoder import NRMSUserEncoder
    from manner_amd.weights import make_mha_pool_weights
    from transformers import BatchEncoding
    out = {}
    for tag, preset, heads, lengths in (("bert", "tiny-bert", 4, np.array([2, 5, 9, 14, 20, 20, 7])),
                                        ("roberta", "tiny-roberta", 2, np.array([3, 18, 6, 11, 18]))):
        cfg = PRESETS[preset]
        w = make_plm_weights(cfg, seed=seed, std=0.05)
        ids, mask = synth_news_tokens(len(lengths), cfg, seed=seed, max_len=int(lengths.max()), lengths=lengths)
        mw = make_mha_pool_weights(cfg.hidden, 200, seed=seed)
        with tempfile.TemporaryDirectory() as tmp:
            enc = PLMTextEncoder(plm_model=hf_model_dir(cfg, w, tmp), frozen_layers=[], text_embedding_dim=cfg.hidden,
                                 num_attention_heads=heads, query_vector_dim=200, dropout_probability=0.2).eval()
            missing, unexpected = enc.load_state_dict({k: torch.from_numpy(v) for k, v in mw.items()}, strict=False)
            assert not unexpected and all(m.startswith("plm_model.") for m in missing), (missing, unexpected)
            res = enc(BatchEncoding({"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)})).numpy()
        out.update({f"plm_{tag}_ids": ids, f"plm_{tag}_mask": mask, f"plm_{tag}_out": res})
        print("PLMTextEncoder", preset, res.shape, float(np.abs(res).mean()))
    rng = np.random.default_rng(seed)
    for tag, dim, heads in (("d96", 96, 2), ("d128", 128, 2)):
        mw = make_mha_pool_weights(dim, 200, seed=seed + 1)
        ue = NRMSUserEncoder(news_embedding_dim=dim, num_attention_heads=heads, query_vector_dim=200).eval()
        ue.load_state_dict({k: torch.from_numpy(v) for k, v in mw.items()}, strict=True)
        clicked = rng.standard_normal((6, 9, dim)).astype(np.float32)
        clicked[1, 4:] = 0.0                             # zero-padded history slots, as to_dense_batch leaves them
        clicked[3, 1:] = 0.0
        res = ue(torch.from_numpy(clicked)).numpy()
        out.update({f"nrms_{tag}_x": clicked, f"nrms_{tag}_out": res})
        print("NRMSUserEncoder", dim, res.shape)
    np.savez_compressed(os.path.join(HERE, "baselines.npz"), **out,
                        meta=json.dumps({"source": "reference PLMTextEncoder (news_encoder.py:132-171) / NRMSUserEncoder "
                                                   "(user_encoder.py:24-42), eval(), transformers " + __import__("transformers").__version__,
                                         "seed": seed, "std": 0.05, "plm": {"bert": ["tiny-bert", 4], "roberta": ["tiny-roberta", 2]},
                                         "nrms": {"d96": [96, 2], "d128": [128, 2]}, "query_dim": 200}))


def gen_train_nrms(seed=58):
    """Gradients of the reference's own NRMSUserEncoder (user_encoder.py:24-42) in train() mode — it has no dropout — for the
    parameters and the input, loss = sum(out * R); zero-padded history slots as to_dense_batch leaves them."""
    from manner.models.components.user_encoder import NRMSUserEncoder
    from manner_amd.weights import make_mha_pool_weights
    out = {}
    rng = np.random.default_rng(seed)
    for tag, dim, heads, b, hmax in (("d128", 128, 2, 7, 9), ("d768", 768, 16, 5, 12)):
        mw = make_mha_pool_weights(dim, 200, seed=seed)
        with torch.enable_grad():
            ue = NRMSUserEncoder(news_embedding_dim=dim, num_attention_heads=heads, query_vector_dim=200).train()
            ue.load_state_dict({k: torch.from_numpy(v) for k, v in mw.items()}, strict=True)
            x = (rng.standard_normal((b, hmax, dim)) * 0.5).astype(np.float32)
            x[1, 4:] = 0.0
            x[3, 1:] = 0.0
            R = rng.standard_normal((b, dim)).astype(np.float32)
            xt = torch.from_numpy(x).requires_grad_(True)
            res = ue(xt)
            (res * torch.from_numpy(R)).sum().backward()
            out.update({f"{tag}_x": x, f"{tag}_R": R, f"{tag}_out": res.detach().numpy(), f"{tag}_grad:x": xt.grad.numpy().copy()})
            # large matrices: the first 8 rows and every 37th row after them (keeps the fixture small)
            out.update({f"{tag}_grad:{k}": (p.grad.numpy().copy() if p.grad.numel() <= 40000 else
                                            p.grad.numpy()[np.r_[0:8, 8:p.grad.shape[0]:37]].copy()) for k, p in ue.named_parameters()})
        print("NRMSUserEncoder.train()", tag, res.shape)
    np.savez_compressed(os.path.join(HERE, "train_nrms.npz"), **out,
                        meta=json.dumps({"source": "reference NRMSUserEncoder.train() (user_encoder.py:24-42), loss = sum(out * R), torch " + torch.__version__,
                                         "seed": seed, "query_dim": 200, "cases": {"d128": [128, 2], "d768": [768, 16]}}))


def gen_train_plm(seed=59):
    """Gradients of the reference's own PLMTextEncoder (news_encoder.py:132-171) in train() mode, every dropout probability 0,
    loss = sum(out * R), on RAGGED token lengths: the padded positions take part in the un-masked attention / pooler, so the
    PLM's gradient flows through them (and into the pad token's embedding row)."""
    from manner.models.components.news_encoder import PLMTextEncoder
    from manner_amd.weights import make_mha_pool_weights
    from transformers import BatchEncoding
    out = {}
    for tag, preset, heads, lengths in (("bert", "tiny-bert", 4, np.array([2, 5, 9, 14, 20, 20, 7])),
                                        ("roberta", "tiny-roberta", 2, np.array([3, 18, 6, 11, 18]))):
        cfg = PRESETS[preset]
        w = make_plm_weights(cfg, seed=seed, std=0.05)
        ids, mask = synth_news_tokens(len(lengths), cfg, seed=seed, max_len=int(lengths.max()), lengths=lengths)
        mw = make_mha_pool_weights(cfg.hidden, 200, seed=seed)
        R = np.random.default_rng(seed).standard_normal((len(lengths), cfg.hidden)).astype(np.float32)
        with tempfile.TemporaryDirectory() as tmp, torch.enable_grad():
            enc = PLMTextEncoder(plm_model=hf_model_dir(cfg, w, tmp, no_dropout=True), frozen_layers=[0], text_embedding_dim=cfg.hidden,
                                 num_attention_heads=heads, query_vector_dim=200, dropout_probability=0.0).train()
            missing, unexpected = enc.load_state_dict({k: torch.from_numpy(v) for k, v in mw.items()}, strict=False)
            assert not unexpected and all(m.startswith("plm_model.") for m in missing), (missing, unexpected)
            res = enc(BatchEncoding({"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}))
            (res * torch.from_numpy(R)).sum().backward()
            grads, frozen = {}, []
            for k, p in enc.named_parameters():
                if k.startswith("plm_model.pooler."):
                    continue
                if p.grad is None:
                    frozen.append(k)
                    continue
                g = p.grad.numpy()
                grads[f"{tag}_grad:{k}"] = (g.copy() if g.size <= 20000 else g[np.r_[0:8, 8:g.shape[0]:37]].copy())
        out.update({f"{tag}_ids": ids, f"{tag}_mask": mask, f"{tag}_R": R, f"{tag}_out": res.detach().numpy(), **grads})
        out[f"{tag}_frozen"] = np.array(frozen)
        print("PLMTextEncoder.train()", preset, res.shape, len(grads), "grad tensors,", len(frozen), "frozen")
    np.savez_compressed(os.path.join(HERE, "train_plm.npz"), **out,
                        meta=json.dumps({"source": "reference PLMTextEncoder.train() (news_encoder.py:132-171), all dropout probabilities 0, "
                                                   "loss = sum(out * R), transformers " + __import__("transformers").__version__,
                                         "seed": seed, "std": 0.05, "plm": {"bert": ["tiny-bert", 4], "roberta": ["tiny-roberta", 2]},
                                         "query_dim": 200, "frozen_layers": [0]}))


if __name__ == "__main__":
    if "--train-plm-only" in sys.argv:
        gen_train_plm()
        sys.exit(0)
    if "--train-nrms-only" in sys.argv:
        gen_train_nrms()
        sys.exit(0)
    if "--train-entities-only" in sys.argv:
        gen_train_entities()
        sys.exit(0)
    if "--baselines-only" in sys.argv:
        gen_baselines()
        sys.exit(0)
    if "--train-only" in sys.argv:
        gen_train("train_tiny_bert", "tiny-bert", n=6, lp=24, seed=51, std=0.05, lengths=np.array([3, 7, 12, 16, 23, 24]),
                  frozen_layers=[0])
        gen_train("train_tiny_roberta", "tiny-roberta", n=5, lp=20, seed=52, std=0.05, lengths=np.array([2, 9, 13, 19, 20]),
                  frozen_layers=[], matrix_rows=8)
        sys.exit(0)
    if "--roberta-base-only" in sys.argv:
        # roberta-base is the PLM of the reference's MIND configs (configs/experiment/cr_module_mind_all_scl_lf.yaml:25)
        gen_encoder("enc_roberta_base", "roberta-base", n=16, lp=96, seed=46, std=0.02,
                    lengths=np.array([5, 9, 12, 16, 17, 23, 31, 32, 33, 47, 48, 64, 65, 80, 95, 96]))
        sys.exit(0)
    if "--roberta-large-only" in sys.argv:
        # BASELINE.json configs[4] at its FULL architecture (24 layers, H = 1024, 16 heads, I = 4096; vocab 50265, positions
        # 514, eps 1e-5, pad 1 — SURVEY §8c): 12 news with lengths {2, 33, 96} each four times, in mixed order
        gen_encoder("enc_roberta_large", "roberta-large", n=12, lp=96, seed=56, std=0.02,
                    lengths=np.array([2, 33, 96, 96, 2, 33, 33, 96, 2, 96, 33, 2]))
        sys.exit(0)
    if "--distilbert-only" in sys.argv:
        keys = json.load(open(os.path.join(HERE, "state_dict_keys.json")))
        keys["tiny-distilbert"] = gen_encoder("enc_tiny_distilbert", "tiny-distilbert", n=12, lp=40, seed=45, std=0.05)
        with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
            json.dump(keys, f, indent=0)
        sys.exit(0)
    if "--round2-only" in sys.argv:
        # >= 64 news incl. the extreme lengths 2 and 96, and the two-text-aspect (pair) input
        lens64 = np.concatenate([[2, 2, 3, 96, 96, 95], np.arange(4, 96, 2)[:46], [17, 33, 49, 64, 65, 80, 81, 31, 32, 63, 94, 5]])
        assert lens64.shape == (64,)
        gen_encoder("enc_bert_base_64", "bert-base-uncased", n=64, lp=96, seed=47, std=0.02, lengths=lens64)
        gen_encoder("enc_roberta_base_64", "roberta-base", n=64, lp=96, seed=48, std=0.02, lengths=lens64)
        gen_pair("enc_pair_bert_base", "bert-base-uncased", n=8, seed=49, std=0.02)
        gen_pair("enc_pair_tiny_bert", "tiny-bert", n=8, seed=50, std=0.05, max_length=48)
        sys.exit(0)
    if "--hidden-only" in sys.argv:
        gen_hidden("hidden_tiny_bert", "tiny-bert", n=6, lp=24, seed=42, std=0.05, lengths=np.array([3, 7, 12, 16, 23, 24]),
                   layers_out=(0, 1, 2))
        gen_hidden("hidden_bert_base", "bert-base-uncased", n=4, lp=32, seed=42, std=0.02, lengths=np.array([5, 17, 31, 32]),
                   layers_out=(8, 12))
        sys.exit(0)
    keys = {}
    keys["tiny-bert"] = gen_encoder("enc_tiny_bert", "tiny-bert", n=12, lp=40, seed=42, std=0.05)
    gen_encoder("enc_tiny_roberta", "tiny-roberta", n=12, lp=40, seed=43, std=0.05)
    # N=16 news with lengths spread over 5..96, bert-base architecture
    lens = np.array([5, 9, 12, 16, 17, 23, 31, 32, 33, 47, 48, 64, 65, 80, 95, 96])
    keys["bert-base-uncased"] = gen_encoder("enc_bert_base", "bert-base-uncased", n=16, lp=96, seed=42,
                                            std=0.02, lengths=lens)
    gen_encoder("enc_bert_base_spread", "bert-base-uncased", n=16, lp=96, seed=44, std=0.05, lengths=lens)
    gen_encoder("enc_roberta_base", "roberta-base", n=16, lp=96, seed=46, std=0.02, lengths=lens)
    keys["tiny-distilbert"] = gen_encoder("enc_tiny_distilbert", "tiny-distilbert", n=12, lp=40, seed=45, std=0.05)
    keys["user_encoder"] = gen_components()
    keys["tiny-bert-entities"] = gen_entities()
    gen_pipeline()
    gen_hidden("hidden_tiny_bert", "tiny-bert", n=6, lp=24, seed=42, std=0.05, lengths=np.array([3, 7, 12, 16, 23, 24]),
               layers_out=(0, 1, 2))
    gen_hidden("hidden_bert_base", "bert-base-uncased", n=4, lp=32, seed=42, std=0.02, lengths=np.array([5, 17, 31, 32]),
               layers_out=(8, 12))
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)
