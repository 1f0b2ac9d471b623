"""Host-side (Python) cost of the drop-in eval forward at the reference's batch size: cProfile of tools/dropin_probe.py 8:eval, the lines of
this repository by cumulative time.  Development aid (round 5)."""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.getcwd())
sys.argv = ["dropin_probe.py", sys.argv[1] if len(sys.argv) > 1 else "8:eval"]
pr = cProfile.Profile()
pr.enable()
exec(open("tools/dropin_probe.py").read())
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("cumulative").print_stats("manner_amd|bench.py", 45)
print(s.getvalue()[:12000])
