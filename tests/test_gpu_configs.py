"""BASELINE.json configs[3] / configs[4] at their real shapes (VERDICT r2 "Next" item 1) — run on the MI355X box: ``pytest -m gpu``.

configs[3]: full MANNeR (CR + category + sentiment A-Modules = three encoders) at the MIND-large shape: three 161 013 x 768
news-embedding tables (495 MB each, beyond the 256 MiB Infinity Cache), 8 192 impressions through
``score_late_fusion -> zscore_fuse -> rank_ndcg`` with an oracle spot-check, and the 8-rank layout of the table exchange —
``equal_news_shards(161013, 8)`` blocks encoded separately and assembled in place — bit-for-bit against the unsharded encode.
Reference: manner/models/ensemble_module.py:95-151 (three ``_submodel_forward`` + z-score + weighted sum),
manner/models/components/news_encoder.py:20,29-37.
configs[4]: the roberta-large architecture at full depth is pinned by tests/golden/enc_roberta_large.npz (test_gpu_parity.py);
here the deferred-LayerNorm 16-bit schedule runs it over several chunks against the fp32 mode.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import manner_oracle as O  # noqa: E402
from manner_amd import distributed as D  # noqa: E402
from manner_amd import hip, hotpath  # noqa: E402
from manner_amd.config import PRESETS  # noqa: E402
from manner_amd.synth import MIND_LARGE, synth_impressions, synth_news_tokens  # noqa: E402
from manner_amd.weights import make_plm_weights  # noqa: E402

DEV = "cuda:0"


def test_config3_mind_large_three_module_pipeline_and_8_rank_table_layout():
    cfg = PRESETS["bert-base-uncased"]
    n_news = MIND_LARGE["n_news"]
    assert n_news == 161013
    fuse_w = (-0.3, 0.2)
    ids_np, mask_np = synth_news_tokens(n_news, cfg, seed=42, max_len=96, profile="title_abstract")
    lens = mask_np.sum(1)
    ids, mask = torch.from_numpy(ids_np).to(DEV), torch.from_numpy(mask_np).to(DEV)
    weight_sets = [make_plm_weights(cfg, seed=42 + 100 * k, std=0.02) for k in range(3)]
    encs = [hip.HipEncoder(cfg, w, precisions=("f16", "fp32"), device=DEV) for w in weight_sets]
    tables = [hotpath.encode_table(e, ids, mask, precision="f16", host_lengths=lens) for e in encs]
    for e in encs:
        e.status()
    assert all(t.shape == (n_news, 768) and bool(torch.isfinite(t).all()) for t in tables)

    # ---- the 8-rank layout of SURVEY §8e: every rank's block encoded on its own, landing in place, == the unsharded table
    shards = D.equal_news_shards(n_news, 8)
    assert shards[0] == (0, 20127) and shards[-1] == (140889, 161013) and all(a[1] == b[0] for a, b in zip(shards, shards[1:]))
    assembled = torch.full((n_news, 768), float("nan"), device=DEV)
    for lo, hi in shards:
        encs[0].encode_cls(ids[lo:hi], mask[lo:hi], precision="f16", host_lengths=lens[lo:hi], out=assembled[lo:hi])
    assert torch.equal(assembled, tables[0])                      # bit for bit: a row does not depend on its launch's composition
    g = D.MeshTableGather(n_news, 768, DEV, pieces=4)             # world size 1 here: the in-place piece layout of the overlapped exchange
    for c in range(4):
        a, b = g.piece_rows(0, c)
        encs[0].encode_cls(ids[a:b], mask[a:b], precision="f16", host_lengths=lens[a:b], out=g.local_out(c))
        g.post(c)
    assert torch.equal(g.wait(), tables[0])
    del assembled, g

    # ---- 24 scattered news against the oracle (CPU restatement of the reference): fp32 mode 1e-4, f16 table rows 2e-2
    pick = np.linspace(0, n_news - 1, 24).astype(np.int64)
    lp = int(lens[pick].max())
    ref_rows = O.encode_cls(ids_np[pick][:, :lp], mask_np[pick][:, :lp], weight_sets[0], cfg)
    rows32 = encs[0].encode_cls(ids[torch.from_numpy(pick).to(DEV)][:, :lp].contiguous(), mask[torch.from_numpy(pick).to(DEV)][:, :lp].contiguous(),
                                precision="fp32").cpu()
    assert float((rows32 - ref_rows).abs().max()) < 1e-4
    assert float((tables[0][torch.from_numpy(pick).to(DEV)].cpu() - ref_rows).abs().max()) < 2e-2

    # ---- 8 192 MIND-large-shaped impressions: fused scorer x3 -> z-score fusion -> rank / nDCG@10
    nb = 8192
    imp = synth_impressions(nb, n_news, seed=43)
    dimp = {k: torch.from_numpy(v).to(DEV) for k, v in imp.items() if k != "labels"}
    labels = torch.from_numpy(imp["labels"]).to(DEV)
    res = hotpath.score_impressions(tables, dimp, weights=fuse_w, labels=labels, k=10)
    hip.check_status(DEV)
    ho, co = imp["hist_off"], imp["cand_off"]
    scores, topk, ndcg = res["scores"].cpu(), res["topk"].cpu(), res["ndcg"].cpu()
    assert scores.shape == (int(co[-1]),) and topk.shape == (nb, 10)
    cpu_tables = [t.cpu() for t in tables]
    checked, worst = 0, 0.0
    for i in range(0, nb, 128):                                    # 64 impressions through the oracle's ensemble restatement
        h = imp["hist_idx"][ho[i]:ho[i + 1]].astype(np.int64)
        c = imp["cand_idx"][co[i]:co[i + 1]].astype(np.int64)
        bh, bc = torch.zeros(len(h), dtype=torch.int64), torch.zeros(len(c), dtype=torch.int64)
        ref = O.ragged(O.ensemble_scores([(t[h], t[c]) for t in cpu_tables], bh, bc, fuse_w), bc)
        got = scores[co[i]:co[i + 1]]
        if len(c) < 2:                                              # torch.std of one value: a NaN row in the reference, here too
            assert bool(torch.isnan(ref).all()) and bool(torch.isnan(got).all())
            continue
        # z-scores are O(1) but divide f32 dot products of magnitude ~|E|^2 by their small spread across the candidates:
        # both sides carry the same few-ulp error of the raw scores amplified by 1 / std
        worst = max(worst, float((got - ref).abs().max()))
        assert float((got - ref).abs().max()) < 2e-2, (i, float((got - ref).abs().max()))
        # ranking and nDCG of the GPU's own scores, recomputed by the oracle: identical indices, nDCG to f32 rounding
        lab = torch.from_numpy(imp["labels"][co[i]:co[i + 1]])
        want_top = O.topk_indices(got, [0, len(c)], 10)[0]
        assert [v for v in topk[i].tolist() if v >= 0] == want_top
        nd, _ = O.ndcg_at_k(got, lab, [0, len(c)], 10)
        assert abs(float(ndcg[i]) - nd) < 1e-6
        checked += 1
    assert checked >= 60
    print(f"configs[3]: fused z-score max-abs difference to the oracle over {checked} impressions: {worst:.3e}")
    # size-independent property at full size: the fusion is invariant to a positive rescaling of any module's table (z-score)
    res2 = hotpath.score_impressions([tables[0], tables[1] * 4.0, tables[2]], dimp, weights=fuse_w, labels=labels, k=10)
    keep = torch.isfinite(res["scores"])
    assert float((res2["scores"][keep] - res["scores"][keep]).abs().max()) < 2e-4
    for e in encs:
        e.close()


def test_config4_roberta_large_full_depth_chunked_16bit_tracks_fp32():
    """roberta-large at full depth (24 layers, H = 1024) on the production 16-bit schedule across several chunks and both
    streams: chunk-size invariant (bit-exact) and within the stated f16 tolerance of the fp32 mode, which the full-depth
    reference golden pins at 1e-4 (test_encoder_fp32_matches_reference[enc_roberta_large])."""
    cfg = PRESETS["roberta-large"]
    assert (cfg.layers, cfg.hidden, cfg.heads, cfg.intermediate) == (24, 1024, 16, 4096)
    w = make_plm_weights(cfg, seed=56, std=0.02)
    enc = hip.HipEncoder(cfg, w, precisions=("f16", "bf16", "fp32"), device=DEV)
    lens = np.array(([2, 33, 96, 64, 17, 95, 5, 80] * 40)[:300])
    ids_np, mask_np = synth_news_tokens(len(lens), cfg, seed=57, lengths=lens)
    ids, mask = torch.from_numpy(ids_np).to(DEV), torch.from_numpy(mask_np).to(DEV)
    ref = enc.encode_cls(ids, mask, precision="fp32", host_lengths=lens)
    out = enc.encode_cls(ids, mask, precision="f16", host_lengths=lens)
    out_small = enc.encode_cls(ids, mask, precision="f16", host_lengths=lens, max_chunk_tokens=4096)
    enc.status()
    assert torch.equal(out, out_small)
    err = float((out - ref).abs().max())
    cos = torch.nn.functional.cosine_similarity(out, ref, dim=1).min().item()
    print(f"roberta-large full depth: f16 vs fp32 max-abs {err:.3e}, min cosine {cos:.7f}")
    assert err < 2e-2 and cos > 0.99999
    bf = float((enc.encode_cls(ids, mask, precision="bf16", host_lengths=lens) - ref).abs().max())
    assert err < bf < 0.15
    enc.close()
