"""profiles/r06_time_to_psnr.txt: held-out PSNR against rays consumed and training wall-clock for the reference trainer's batch
(2^18 packed samples per step), the headline's (2^21) and the 8-GPU weak-scaling batch (8 x 2^21, emulated by gradient accumulation),
out to the reference's 100-epoch ray budget - bench_quality.time_to_psnr over several seeds.
usage: python scripts/time_to_psnr.py OUT.txt [seeds=3] [extra "regime:lr_scale[:ray_budget[:prune_every]]" ...]      (seeds < 0: ONLY the extra runs, |seeds| seeds)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import numpy as np
import torch
import bench_quality as bq
import synlego

out_path = sys.argv[1]
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B0 = bq.REFERENCE_RAY_BUDGET
runs = [("ref_2p18", 1.0, B0, 100), ("headline_2p21", 1.0, B0, 100), ("dp8_8x2p21", 1.0, B0, 100)] if seeds > 0 else []
seeds = abs(seeds)
for spec in sys.argv[3:]:
    parts = spec.split(":")
    runs.append((parts[0], float(parts[1]), int(float(parts[2])) if len(parts) > 2 else B0, int(parts[3]) if len(parts) > 3 else 100))
dev = torch.device("cuda", 0)
train_bank = synlego.ray_bank(2 ** 23, seed=1000, device=dev)             # 8.4 M training rays of the 100 training views
eval_bank = synlego.ray_bank(2 ** 16, seed=7, device=dev)                 # held-out views (other cameras)
marks = (2.5e6, 5e6, 1e7, 2e7, 3e7, 4e7)
rows = []
for name, sc, budget, pe in runs:
    for seed in range(seeds):
        mk = marks if budget == B0 else tuple(budget * f for f in (0.0625, 0.125, 0.25, 0.5, 0.75, 1.0 - 1e-9))
        r = bq.time_to_psnr(dev, name, train_bank, eval_bank, ray_budget=budget, checkpoints=mk, seed=seed, lr_scale=sc, prune_every=pe)
        r["budget"] = budget
        rows.append(r)
        print(json.dumps({k: v for k, v in r.items() if k != "curve"}), flush=True)
with open(out_path, "w") as f:
    f.write("Held-out PSNR (dB, 65 536 rays of views the training never saw) of nerf_hash.yaml on SynLego, bf16 step, from the dense level-7\n"
            "octree, prune every `prune` steps, MultiStepLR x0.333 at 50 / 75 / 90 %% of the ray budget; mean over %d seeds (min .. max).\n"
            "rays = training rays consumed; the budget 4.1e7 = 100 epochs x 100 views x 4096 rays (base_trainer.py:198-203).\n"
            "train s = wall-clock of the training steps alone (evaluation excluded), one MI355X.\n\n" % seeds)
    f.write(f"{'regime':<18}{'lr x':>6}{'prune':>6}{'budget':>9}{'steps':>8}{'train s':>9}{'final dB':>22}  " + "".join(f"{m:>20.3g}" for m in marks) + "\n")
    for name, sc, budget, pe in runs:
        mine = [r for r in rows if r["regime"] == name and r["lr_scale"] == sc and r["budget"] == budget and r["prune_every"] == pe]
        cols = []
        for m in marks:
            v = [r["psnr_at_rays"][str(int(m))] for r in mine if str(int(m)) in r["psnr_at_rays"]]
            cols.append(f"{np.mean(v):6.2f} ({min(v):5.2f}..{max(v):5.2f})" if v else "-")
        fin = [r["curve"][-1][3] for r in mine]
        f.write(f"{name:<18}{sc:>6.2f}{pe:>6d}{budget:>9.2g}{int(np.mean([r['optimizer_steps'] for r in mine])):>8d}{np.mean([r['train_seconds'] for r in mine]):>9.2f}"
                f"{np.mean(fin):>8.2f} ({min(fin):5.2f}..{max(fin):5.2f})  " + "".join(f"{c:>20}" for c in cols) + "\n")
    f.write("\nPSNR against training wall-clock (seed 0): regime: (rays, optimizer steps, train s, dB) ...\n")
    for name, sc, budget, pe in runs:
        r = next(r for r in rows if r["regime"] == name and r["lr_scale"] == sc and r["budget"] == budget and r["prune_every"] == pe and r["seed"] == 0)
        f.write(f"{name} x{sc:g} prune {pe} budget {budget:.3g}: " + "  ".join(f"({c[0]:.3g}, {c[1]}, {c[2]:.2f}s, {c[3]:.2f})" for c in r["curve"]) + "\n")
    f.write("\nraw: " + json.dumps([{k: v for k, v in r.items() if k != 'curve'} for r in rows]) + "\n")
print(open(out_path).read())
