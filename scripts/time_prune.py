"""Where does a prune go?  Times NeuralRadianceField.prune's stages on the bench pipeline (level-7, T=2^19)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
import torch
import bench, synlego
from wisp.accelstructs import OctreeAS
from wisp.core import Rays
from wisp.trainers import MultiviewTrainStep

dev = torch.device("cuda", 0)
dense = OctreeAS.make_dense(level=7).points[-(128 ** 3):].to(dev)
pipe = bench.build_pipeline(dev, 64, 2048, dense)
tr = MultiviewTrainStep(pipe, prune_every=-1, target_sample_size=2 ** 21, enable_amp=True)
o, d, rgb = synlego.ray_bank(2 ** 18, seed=1, device=dev)
for it in range(120):
    idx = torch.randint(0, o.shape[0], (2500,), device=dev)
    tr.step(Rays(o[idx], d[idx], dist_min=1.0, dist_max=5.0), rgb[idx])


def t(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, r


nef = pipe.nef
cells = nef.grid.dense_points.shape[0]
ms, _ = t(lambda: tr.prune()); print(f"trainer.prune() total            {ms:8.3f} ms   cells kept {int(nef.grid.blas.pyramid[0, 7])}")
g = tr._prune_gen
ms, (unit, views) = t(lambda: (torch.rand(cells, 3, generator=g, device=g.device),
                               torch.nn.functional.normalize(torch.randn(cells, 3, generator=g, device=g.device), dim=1)))
print(f"  draws on {g.device}                 {ms:8.3f} ms")
pts = nef.grid.dense_points.to(dev)
samples = ((pts.float() + unit) / 128.0) * 2.0 - 1.0
with torch.no_grad():
    ms, dens = t(lambda: nef.forward(coords=samples, ray_d=views, channels="density")); print(f"  density forward (fp32)          {ms:8.3f} ms")
    with torch.autocast('cuda', dtype=torch.bfloat16):
        ms, _ = t(lambda: nef.forward(coords=samples, ray_d=views, channels="density")); print(f"  density forward (bf16 autocast) {ms:8.3f} ms")
keep = nef.grid.occupancy.to(dev) > nef.prune_min_density
ms, b = t(lambda: OctreeAS.from_leaf_mask(keep, 7)); print(f"  OctreeAS.from_leaf_mask         {ms:8.3f} ms")
ms, b2 = t(lambda: OctreeAS.from_quantized_points(pts[keep], 7)); print(f"  from_quantized_points(pts[keep]) {ms:8.3f} ms")
import wisp.ops.spc as S
ms, _ = t(lambda: S.octree_to_spc(b.octree.clone()), n=2); print(f"  generic octree_to_spc (old path) {ms:8.3f} ms")
ms, _ = t(lambda: (b._occ_bits.clear(), b._bitfield(7), b._bitfield(4))); print(f"  bitfields level 7 + 4           {ms:8.3f} ms")
