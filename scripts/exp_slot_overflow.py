"""How often does a record slot of the binned hash-grid backward fill up (-> the atomic fall-back, then grown again) over a whole
bench run - 300 pre-training steps from the dense octree with a prune every 100, then the timed windows - at the slot headroom
given by WISP_HG_SLOT_HEADROOM?  Checks EVERY launch instead of every 32nd (so the sizes also adapt faster than in production:
the count is an upper bound on what a check would see, per launch)."""
import io, os, sys, contextlib, json
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "kaolin-wisp_amd"))
import wisp._C as C
every = int(os.environ.get("EXP_CHECK_EVERY", "32"))
C._SlotFit.CHECK_EVERY = every
seen = {"checks": 0, "overflowing_checks": 0, "levels": 0, "worst": 0.0}
orig = C._SlotFit._collect
def collect(self):
    had = self.pending is not None
    orig(self)
    if had and self.pending is None and self.last:
        seen["checks"] += 1
        over = [l for l, (f, c, b) in enumerate(zip(self.last["fill"], self.last["cap"], self.last["base"])) if b > 0 and f >= c]
        seen["overflowing_checks"] += 1 if over else 0
        seen["levels"] += len(over)
        for f, c, b in zip(self.last["fill"], self.last["cap"], self.last["base"]):
            if b > 0:
                seen["worst"] = max(seen["worst"], f / c)
C._SlotFit._collect = collect
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(["--steps", "200", "--warmup", "5", "--no-pmc", "--no-cpu-baseline", "--no-configs", "--dropin-steps", "0"])
j = json.loads(buf.getvalue().strip().splitlines()[-1])
print("headroom %s, a check every %d launches: %d checks, %d of them saw a full slot (%d level-checks), fullest slot / capacity at most %.2f; scratch at the end %.3f GB, step %.4f ms" % (
    os.environ.get("WISP_HG_SLOT_HEADROOM", "1.2"), every, seen["checks"], seen["overflowing_checks"], seen["levels"], seen["worst"],
    j["roofline"]["hashgrid_bwd_scratch"]["workspace_bytes"] / 1e9, j["ms_per_step"]))
