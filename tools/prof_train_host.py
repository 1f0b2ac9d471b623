"""Host-side (Python) cost of the training step: cProfile of tools/train_probe.py, the lines of this repository by cumulative time, then
the top functions by own time.  Development aid (round 5: found torch.bincount's device wait inside segment_offsets)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.getcwd())
os.environ["TRAIN_PROBE_VARIANT"] = "reference_default_embeddings_trainable"
os.environ["TRAIN_PROBE_STEPS"] = "30"
sys.argv = ["train_probe.py", "bf16"]
pr = cProfile.Profile()
pr.enable()
exec(open("tools/train_probe.py").read())
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("cumulative").print_stats("manner_amd|bench.py", 40)
st.sort_stats("tottime").print_stats(25)
print(s.getvalue()[:14000])
