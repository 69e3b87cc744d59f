"""Host-side profile of the VQAD (C5 stand-in) training step."""
import cProfile, pstats, io, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import torch
import bench_configs, bench
args = types.SimpleNamespace(pretrain=20, steps=60, warmup=3, target_samples=2 ** 21, precision="bf16", sdf_batch=512)
dev = torch.device("cuda", 0)
orig = bench_configs._nerf_run
def hooked(a, dev_, pipe, trainer, bank, steps, warmup, label, metric, bytes_fn):
    out = orig(a, dev_, pipe, trainer, bank, steps, warmup, label, metric, bytes_fn)
    import wisp._C as C
    from wisp.core import Rays
    import synlego
    o, d, rgb = bank
    R = out["config"]["rays_per_step_per_gpu"]
    def batch():
        idx = torch.randint(0, o.shape[0], (R,), device=dev_)
        a_, b_, c_ = C.gather_rows(idx, [o, d, rgb])
        return Rays(a_, b_, dist_min=synlego.NEAR, dist_max=synlego.FAR), c_
    pr = cProfile.Profile(); pr.enable()
    r, g = batch()
    for _ in range(50):
        nr, ng = batch(); trainer.step(r, g, prefetch=nr); r, g = nr, ng
    torch.cuda.synchronize(); pr.disable()
    for key in ("tottime", "cumulative"):
        buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(45); print(buf.getvalue()[:7000])
    return out
bench_configs._nerf_run = hooked
r = bench_configs.run_vqad(args, dev)
print(r["ms_per_step"], r["gpu_busy_fraction"])
