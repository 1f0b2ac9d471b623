#!/usr/bin/env python3
"""Host-side cost of ONE eval forward of the module mirror (a 2-news call: the device work is negligible, what is left is the Python and
launch path every call pays before and after its kernels).  cProfile by own time.  Development aid (round 5)."""
import cProfile, io, os, pstats, sys, time, warnings
sys.path.insert(0, os.getcwd())
import torch
from manner_amd.models.components.news_encoder import MannerNewsEncoder
warnings.simplefilter("ignore")
dev = torch.device("cuda", 0)
enc = MannerNewsEncoder(plm_model="bert-base-uncased", frozen_layers=list(range(8)), dropout_probability=0.2, use_entities=False,
                        entity_embeddings=None, entity_embedding_dim=100, num_attention_heads=10, query_vector_dim=200,
                        text_embedding_dim=768).to(dev).eval()
x = {"text": {"input_ids": torch.randint(5, 3000, (2, 16), device=dev), "attention_mask": torch.ones(2, 16, dtype=torch.int64, device=dev)}}
with torch.no_grad():
    for _ in range(20):
        enc(x)
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for _ in range(n):
        enc(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"enqueue per forward: {(t1 - t0) / n * 1e6:.1f} us (wall incl. drain {(time.perf_counter() - t0) / n * 1e6:.1f})")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        enc(x)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32)
print(s.getvalue()[:9000])
