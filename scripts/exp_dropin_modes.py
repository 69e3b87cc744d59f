"""The drop-in regime's host loop has a fast and a slow mode from one process to the next.  Is the slow one the HOST (Python, launches)
or the DEVICE (the loop's three read-backs waiting longer for the same kernels, e.g. at lower clocks)?  bench.main() with
torch.Tensor.item timed during dropin_regime(); prints ms per iteration and the share of it spent waiting inside item()."""
import io, os, sys, contextlib, json, time
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
import torch
import bench
wait = [0.0, 0]
orig_item = torch.Tensor.item
def timed_item(self):
    t0 = time.perf_counter()
    v = orig_item(self)
    wait[0] += time.perf_counter() - t0
    wait[1] += 1
    return v
orig = bench.dropin_regime
def wrapped(*a, **k):
    pre = os.environ.get("EXP_PRE", "")
    if "gc" in pre:
        import gc
        gc.collect()
    if "empty" in pre:
        torch.cuda.synchronize(); torch.cuda.empty_cache()
    if "stats" in pre:
        st = torch.cuda.memory_stats()
        sys.stderr.write("allocator: %d segments, %d active blocks, %d inactive split blocks, reserved %.2f GB, allocated %.2f GB, %d hipMalloc calls so far\n" % (
            st["segment.all.current"], st["active.all.current"], st["inactive_split.all.current"], st["reserved_bytes.all.current"] / 1e9,
            st["allocated_bytes.all.current"] / 1e9, st["num_device_alloc"]))
    torch.Tensor.item = timed_item
    try:
        return orig(*a, **k)
    finally:
        torch.Tensor.item = orig_item
bench.dropin_regime = wrapped
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(["--steps", sys.argv[1], "--warmup", "5", "--no-pmc", "--no-cpu-baseline", "--no-configs"])
j = json.loads(buf.getvalue().strip().splitlines()[-1])
iters = wait[1] / 4.0 if wait[1] else 1
clk = os.popen("rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk' | head -2").read().strip().replace("\n", " | ")
print("[" + os.environ.get("EXP_PRE", "") + "] steps %s: drop-in %.3f ms per iteration; item(): %d calls, %.3f ms per call; clocks now: %s" % (
    sys.argv[1], j["dropin_regime"]["ms_per_step"], wait[1], 1e3 * wait[0] / max(wait[1], 1), clk))
