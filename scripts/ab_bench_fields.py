"""Runs the bench's headline + 2^18 regimes only and prints the fields an A/B of the hash-grid backward looks at."""
import io, contextlib, json, os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main(["--steps", os.environ.get("AB_STEPS", "100"), "--warmup", "5", "--no-cpu-baseline", "--no-pmc", "--no-configs", "--dropin-steps", "0"])
j = json.loads(buf.getvalue().strip().splitlines()[-1])
r = j["roofline"]; s = r["hashgrid_bwd_scratch"]
print("ms/step %.4f (window %.4f)  bwd pair %.4f ms  frac %.3f  2^18: %.4f ms (%.4f without prunes)  scratch %.3f GB for %.3f GB of records  psnr %s" % (
    j["ms_per_step"], j["timed_window"]["ms_per_step"], r["avg_launch_ms"], r["frac"], j["reference_regime"]["ms_per_step"],
    j["reference_regime"]["ms_per_step_without_its_prunes"], s["workspace_bytes"] / 1e9, s["record_bytes_written"] / 1e9, j.get("psnr", j.get("quality"))))
print("  caps", s["per_level_slot_capacity"]); print("  fullest", s["per_level_fullest_slot"])
