"""Content-addressed news-embedding cache (csrc/cache.hip, hip.NewsEmbeddingCache, MannerTextEncoder.embedding_cache_rows) —
SURVEY.md §8(d) mode T behind the unchanged drop-in call pattern.  Run on the MI355X box: ``pytest -m gpu``.

The reference encodes every occurrence of a news again (manner/models/cr_module.py:107,113 -> news_encoder.py:29-37); the cache
must never change a number: its keys are compared with a Python dictionary of token tuples, its table bookkeeping with a Python
model of the same rules, and the module's cached forward with the uncached forward, bit for bit.
"""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from manner_amd import hip  # noqa: E402
from manner_amd.config import PRESETS  # noqa: E402
from manner_amd.models.components.news_encoder import MannerTextEncoder  # noqa: E402
from manner_amd.synth import synth_news_tokens  # noqa: E402
from manner_amd.weights import make_plm_weights  # noqa: E402

DEV = "cuda:0"


def _rows(n, seed, max_len=40, vocab=50, width=None):
    """n token rows from a SMALL vocabulary and few lengths (so that equal rows and equal prefixes occur), right-padded."""
    g = np.random.default_rng(seed)
    lens = g.integers(1, max_len + 1, n)
    width = width or max_len
    ids = np.zeros((n, width), np.int64)
    mask = np.zeros((n, width), np.int64)
    for i, ln in enumerate(lens):
        ids[i, :ln] = g.integers(1, vocab, ln) if g.random() < 0.7 else np.arange(1, ln + 1)       # the second kind repeats often
        mask[i, :ln] = 1
    return ids, mask


def _tuples(ids, mask):
    return [tuple(int(t) for t, m in zip(r, mr) if m) for r, mr in zip(ids, mask)]


def test_keys_depend_on_the_real_tokens_only():
    """manner_hip_news_key128: equal keys <=> equal real-token sequences — whatever the padded width, whatever sits in the padded
    positions, whatever the row's place in the call; a permutation of the tokens, a prefix, an extra token all change the key."""
    cache = hip.NewsEmbeddingCache(8, 16, DEV)
    ids, mask = _rows(6000, seed=1)
    keys = cache.keys(torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)).cpu().numpy()
    assert (keys[:, 0] != 0).all()
    by_tokens = {}
    for t, k in zip(_tuples(ids, mask), map(tuple, keys)):
        assert by_tokens.setdefault(t, k) == k                     # equal tokens -> equal key
    assert len(set(by_tokens.values())) == len(by_tokens)          # different tokens -> different key
    assert len(by_tokens) < 6000                                   # the draw does contain repeats
    # the same rows at another padded width, with garbage under the zero mask, in another order
    wide_ids = np.full((6000, 57), 12345, np.int64)
    wide_mask = np.zeros((6000, 57), np.int64)
    wide_ids[:, :40][mask == 1] = ids[mask == 1]
    wide_mask[:, :40] = mask
    perm = np.random.default_rng(2).permutation(6000)
    keys2 = cache.keys(torch.from_numpy(wide_ids[perm]).to(DEV), torch.from_numpy(wide_mask[perm]).to(DEV)).cpu().numpy()
    assert np.array_equal(keys2, keys[perm])
    # order matters, length matters
    a = np.array([[5, 6, 7, 0], [7, 6, 5, 0], [5, 6, 0, 0], [5, 6, 7, 7]], np.int64)
    m = (a != 0).astype(np.int64)
    k = cache.keys(torch.from_numpy(a).to(DEV), torch.from_numpy(m).to(DEV)).cpu().numpy()
    assert len({tuple(r) for r in k}) == 4
    # an empty call and an all-padding row are legal
    assert cache.keys(torch.zeros((0, 4), dtype=torch.int64, device=DEV), torch.zeros((0, 4), dtype=torch.int64, device=DEV)).shape == (0, 2)
    z = cache.keys(torch.zeros((2, 4), dtype=torch.int64, device=DEV), torch.zeros((2, 4), dtype=torch.int64, device=DEV)).cpu().numpy()
    assert np.array_equal(z[0], z[1]) and z[0, 0] != 0


def test_lookup_follows_the_table_rules():
    """manner_hip_news_cache_lookup against a Python model: a key seen before -> state 0 and its row; a new key -> state 1 for
    exactly ONE of its occurrences in the call (the others: state 0, same row), rows handed out densely; once `capacity` rows are out a
    new key gets state 2 / row -1 on every occurrence and stays that way; clear() forgets everything."""
    cap = 300
    cache = hip.NewsEmbeddingCache(4, cap, DEV)
    assert cache.n_slots == 1024
    known = {}                   # tokens -> row (None: seen after the table was full)
    handed = 0
    for call in range(12):
        ids, mask = _rows(257, seed=100 + call, max_len=12, vocab=6)
        rows, state = cache.lookup(torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV))
        rows, state = rows.cpu().numpy(), state.cpu().numpy()
        new_in_call = {}
        for t, r, s in zip(_tuples(ids, mask), rows, state):
            if t in known:
                if known[t] is None:
                    assert (s, r) == (2, -1), (call, t, s, r)
                else:
                    assert s == 0 and r == known[t], (call, t, s, r)
            else:
                new_in_call.setdefault(t, []).append((int(s), int(r)))
        for t, occ in new_in_call.items():
            states = sorted(s for s, _ in occ)
            if states[0] == 2:                                      # the table filled up
                assert all(o == (2, -1) for o in occ)
                known[t] = None
            else:
                assert states.count(1) == 1 and states.count(0) == len(occ) - 1, (call, t, occ)
                assert len({r for _, r in occ}) == 1
                known[t] = occ[0][1]
                handed += 1
        got = sorted(r for r in known.values() if r is not None)
        assert got == list(range(len(got))) and len(got) == min(handed, cap)      # dense, unique rows
    assert sum(r is None for r in known.values()) > 0 and len([r for r in known.values() if r is not None]) == cap
    cache.clear()
    ids, mask = _rows(64, seed=100, max_len=12, vocab=6)
    rows, state = cache.lookup(torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV))
    assert int((state == 1).sum()) == len(set(_tuples(ids, mask))) and int(rows.max()) == len(set(_tuples(ids, mask))) - 1


def _text_encoder(seed=5):
    cfg = PRESETS["tiny-bert"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc = MannerTextEncoder("tiny-bert", frozen_layers=[0], dropout_probability=0.2)
    w = make_plm_weights(cfg, seed=seed, std=0.05, with_pooler=True)
    enc.plm_model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    return cfg, enc.to(DEV).eval()


@pytest.mark.parametrize("precision", ["fp32", "f16"])
def test_cached_forward_is_bit_identical_to_encoding_every_occurrence(precision):
    """MannerTextEncoder.forward (eval, no_grad) with embedding_cache_rows > 0 against the same module without: batches that
    share news with earlier batches, list a news several times, arrive at different padded widths — every output row equal to
    the bit; only unseen rows reach the encoder (the cache's counters); a capacity smaller than the number of distinct news
    changes nothing but the hit rate; a weight update or another precision empties the cache."""
    cfg, enc = _text_encoder()
    enc.precision = precision
    pool_ids, pool_mask = synth_news_tokens(400, cfg, seed=9, max_len=48)
    g = np.random.default_rng(3)

    def batch(n, width):
        pick = g.integers(0, 400, n)
        pick[::7] = pick[0]                                        # one news many times in the batch
        lp = max(int(pool_mask[pick].sum(1).max()), width)
        return pick, {"input_ids": torch.from_numpy(pool_ids[pick][:, :lp].copy()).to(DEV),
                      "attention_mask": torch.from_numpy(pool_mask[pick][:, :lp].copy()).to(DEV)}

    batches = [batch(n, w) for n, w in ((90, 0), (150, 48), (33, 40), (260, 0), (90, 48))]
    with torch.no_grad():
        enc.embedding_cache_rows = 0
        plain = [enc(b) for _, b in batches]
        for cap in (1024, 120):                                    # 120 < distinct news of the run: part of the rows is never stored
            enc.embedding_cache_rows = cap
            seen = set()
            for (pick, b), ref in zip(batches, plain):
                before = enc._cache.encoded if enc._cache is not None and enc._cache.capacity == cap else 0
                out = enc(b)
                assert torch.equal(out, ref)
                new = set(pick.tolist()) - seen
                if cap == 1024:
                    assert enc._cache.encoded - before == len(new)          # only the unseen news were encoded, each once
                else:
                    assert enc._cache.encoded - before >= len(new)
                seen |= set(pick.tolist())
            if cap == 1024:
                assert enc._cache.lookups == sum(len(p) for p, _ in batches) and enc._cache.encoded == len(seen)
        # a weight update: the next call must not serve the old embeddings
        enc.embedding_cache_rows = 1024
        first = enc(batches[0][1])
        with torch.no_grad():
            enc.plm_model.get_parameter("encoder.layer.1.output.dense.weight").mul_(1.5)
        enc.embedding_cache_rows = 0
        fresh = enc(batches[0][1])
        enc.embedding_cache_rows = 1024
        again = enc(batches[0][1])
        assert not torch.equal(fresh, first) and torch.equal(again, fresh)
        other = "f16" if precision == "fp32" else "fp32"
        enc.precision = other
        enc.embedding_cache_rows = 0
        ref_other = enc(batches[1][1])
        enc.embedding_cache_rows = 1024
        assert torch.equal(enc(batches[1][1]), ref_other) and not torch.equal(ref_other, plain[1])
    enc.check_inputs()


def test_cache_is_bypassed_when_a_graph_is_recorded():
    """train() mode and eval() with grad mode on run the differentiable engine (dropout, autograd): nothing is cached there."""
    cfg, enc = _text_encoder()
    enc.embedding_cache_rows = 256
    ids, mask = synth_news_tokens(12, cfg, seed=4, max_len=20)
    b = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    enc.train()
    out = enc(b)
    assert out.requires_grad and enc._cache is None
    enc.eval()
    with torch.no_grad():
        enc(b)
    assert enc._cache is not None and enc._cache.encoded == 12


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_prefix_cache_training_is_bit_identical_to_recomputing_the_frozen_prefix(precision):
    """with the embeddings and layer 0 frozen the hidden states after layer 0 are constant across steps;
    `MannerTextEncoder.prefix_cache_rows` keeps them per news in HBM (hip.PrefixCache) instead of running the frozen layers again.
    Same seeds -> the outputs and every gradient of train() forwards/backwards are equal to the bit with the cache on and off, over
    batches that share news, repeat news and arrive at different padded widths (a row stored at a narrow width reads zeros beyond it,
    as encode_hidden writes them); only unseen news reach the prefix engine; changing a FROZEN weight empties the table."""
    cfg, enc = _text_encoder(seed=6)
    enc.train_precision = precision
    for n, p in enc.plm_model.named_parameters():
        if n.startswith("embeddings."):
            p.requires_grad_(False)
    pool_ids, pool_mask = synth_news_tokens(300, cfg, seed=12, max_len=40)
    g = np.random.default_rng(5)

    def batch(n, width):
        pick = g.integers(0, 300, n)
        pick[::5] = pick[1]
        lp = max(int(pool_mask[pick].sum(1).max()), width)
        return pick, {"input_ids": torch.from_numpy(pool_ids[pick][:, :lp].copy()).to(DEV),
                      "attention_mask": torch.from_numpy(pool_mask[pick][:, :lp].copy()).to(DEV)}

    batches = [batch(n, w) for n, w in ((40, 0), (64, 40), (25, 30), (64, 0))]
    R = torch.from_numpy(np.random.default_rng(6).standard_normal((64, cfg.hidden)).astype(np.float32)).to(DEV)
    enc.train()

    def run(rows):
        enc.prefix_cache_rows, enc.prefix_cache_len = rows, 40
        res = []
        for k, (_, b) in enumerate(batches):
            for p in enc.parameters():
                p.grad = None
            torch.manual_seed(100 + k)                              # the dropout seed comes from torch's CPU generator
            out = enc(b)
            (out * R[:out.shape[0]]).sum().backward()
            res.append((out.detach().clone(), {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}))
        return res

    plain = run(0)
    assert getattr(enc, "_prefix_cache", None) is None
    cached = run(512)
    pc = enc._prefix_cache
    seen = set()
    for _, (pick, _) in zip(cached, batches):
        seen |= set(pick.tolist())
    assert pc.lookups == sum(len(p) for p, _ in batches) and pc.encoded == len(seen)
    for (o1, g1), (o2, g2) in zip(plain, cached):
        assert torch.equal(o1, o2)
        assert g1.keys() == g2.keys() and len(g1) > 4
        for n in g1:
            assert torch.equal(g1[n], g2[n]), n
    again = run(512)                                               # every news cached now
    assert pc.encoded == len(seen)
    for (o1, _), (o2, _) in zip(plain, again):
        assert torch.equal(o1, o2)
    # a frozen weight changes: the table must not serve the old hidden states
    with torch.no_grad():
        enc.plm_model.get_parameter("encoder.layer.0.output.dense.weight").mul_(1.3)
    fresh = run(0)
    after = run(512)
    assert not torch.equal(fresh[0][0], plain[0][0])
    for (o1, _), (o2, _) in zip(fresh, after):
        assert torch.equal(o1, o2)


def test_warming_the_cache_with_the_news_pool_makes_every_batch_a_hit():
    """`warm_embedding_cache(pool)` — the table build of mode T in large calls — then batches drawn from the pool reach the encoder
    with nothing to encode, and their outputs are the uncached forward's to the bit."""
    cfg, enc = _text_encoder(seed=8)
    pool_ids, pool_mask = synth_news_tokens(500, cfg, seed=21, max_len=32)
    pool = {"input_ids": torch.from_numpy(pool_ids).to(DEV), "attention_mask": torch.from_numpy(pool_mask).to(DEV)}
    pick = np.random.default_rng(2).integers(0, 500, 96)
    lp = int(pool_mask[pick].sum(1).max())
    b = {"input_ids": pool["input_ids"][pick][:, :lp].contiguous(), "attention_mask": pool["attention_mask"][pick][:, :lp].contiguous()}
    with torch.no_grad():
        ref = enc(b)
    with pytest.raises(RuntimeError):
        enc.warm_embedding_cache(pool)                             # the cache is off
    enc.embedding_cache_rows = 600
    n_unique = len({tuple(r[m == 1]) for r, m in zip(pool_ids, pool_mask)})
    assert enc.warm_embedding_cache(pool, chunk=128) == n_unique
    assert enc.warm_embedding_cache(pool, chunk=128) == 0          # idempotent
    with torch.no_grad():
        before = enc._cache.encoded
        out = enc(b)
    assert enc._cache.encoded == before and torch.equal(out, ref)


def test_cache_created_under_inference_mode_serves_no_grad_calls_too():
    """Lightning's test loop runs under torch.inference_mode(); a later torch.no_grad() call (e.g. `warm_embedding_cache`) must be
    able to update the same table in place — the cache's tensors are ordinary tensors whatever mode created them."""
    cfg, enc = _text_encoder(seed=9)
    enc.embedding_cache_rows = 128
    ids, mask = synth_news_tokens(40, cfg, seed=31, max_len=24)
    b1 = {"input_ids": torch.from_numpy(ids[:20]).to(DEV), "attention_mask": torch.from_numpy(mask[:20]).to(DEV)}
    b2 = {"input_ids": torch.from_numpy(ids).to(DEV), "attention_mask": torch.from_numpy(mask).to(DEV)}
    with torch.inference_mode():
        first = enc(b1).clone()
    with torch.no_grad():
        second = enc(b2)
    with torch.inference_mode():
        third = enc(b2)
    assert torch.equal(second[:20], first) and torch.equal(third, second) and enc._cache.encoded == 40

