"""How much of the drop-in regime's host loop is Python's cyclic collector?  gc.callbacks time every collection during
dropin_regime() (bench.main(), EXP_ITERS timed iterations after a 200-step headline window).  EXP_FREEZE=1: gc.collect() + gc.freeze()
first (everything alive after set-up moves to the permanent generation: a full collection then only walks what the loop created)."""
import gc, io, os, sys, contextlib, json, time
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
import bench
pauses = {0: [], 1: [], 2: []}
t0 = [0.0]
def cb(phase, info):
    if phase == "start":
        t0[0] = time.perf_counter()
    else:
        pauses[info["generation"]].append(time.perf_counter() - t0[0])
orig = bench.dropin_regime
def wrapped(*a, **k):
    if os.environ.get("EXP_FREEZE") == "1":
        gc.collect(); gc.freeze()
    gc.callbacks.append(cb)
    try:
        return orig(*a, **k)
    finally:
        gc.callbacks.remove(cb)
bench.dropin_regime = wrapped
iters = os.environ.get("EXP_ITERS", "600")
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(["--steps", "200", "--warmup", "5", "--no-pmc", "--no-cpu-baseline", "--no-configs", "--dropin-steps", iters])
j = json.loads(buf.getvalue().strip().splitlines()[-1])
n = int(iters) + 107
print("freeze=%s: drop-in %.3f ms per iteration over %s timed iterations; collections during the %d iterations of dropin_regime (warm-up included): %s; objects tracked now: %d, frozen: %d" % (
    os.environ.get("EXP_FREEZE", "0"), j["dropin_regime"]["ms_per_step"], iters, n,
    ", ".join("gen%d: %d x, %.1f ms in all, longest %.1f ms" % (g, len(v), 1e3 * sum(v), 1e3 * max(v or [0])) for g, v in pauses.items()),
    len(gc.get_objects()), gc.get_freeze_count()))
