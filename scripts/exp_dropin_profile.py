"""cProfile of dropin_regime() inside bench.main(): functions called at least once per iteration, by own time (fast mode after a
20-step headline window, slow mode after a 200-step one: compare the two listings)."""
import cProfile, io, os, pstats, sys, contextlib, json
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
import bench
orig = bench.dropin_regime
rows = []
def wrapped(*a, **k):
    pr = cProfile.Profile()
    pr.enable()
    try:
        return orig(*a, **k)
    finally:
        pr.disable()
        st = pstats.Stats(pr)
        for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
            if nc >= 200:
                rows.append((tt, ct, nc, "%s:%d(%s)" % (fn.replace("/usr/local/lib/python3.10/dist-packages/", "").replace(os.path.abspath(root) + "/", ""), line, name)))
bench.dropin_regime = wrapped
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(["--steps", sys.argv[1], "--warmup", "5", "--no-pmc", "--no-cpu-baseline", "--no-configs"])
j = json.loads(buf.getvalue().strip().splitlines()[-1])
print("steps %s: drop-in %.3f ms per iteration (under cProfile); own ms / cumulative ms / calls" % (sys.argv[1], j["dropin_regime"]["ms_per_step"]))
for tt, ct, nc, name in sorted(rows, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 32]:
    print("  %8.1f %8.1f %7d  %s" % (1e3 * tt, 1e3 * ct, nc, name[:130]))
