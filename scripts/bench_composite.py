"""compositing + loss + backward of a training step: three launches against wisp_composite_loss, at the headline shape
(38 K rays x ~53 samples) and at the reference trainer's (2^18 samples)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "kaolin-wisp_amd"))
import wisp._C as C
dev = torch.device("cuda:0")
torch.manual_seed(0)
for R, mean in ((38000, 53), (4800, 53)):
    lens = torch.poisson(torch.full((R,), float(mean))).long().to(dev)
    offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(lens, 0)])
    S = int(offs[-1])
    color, dens, delt = torch.rand(S, 3, device=dev), torch.rand(S, 1, device=dev) * 20, torch.full((S, 1), 0.002, device=dev)
    gts = torch.rand(R, 3, device=dev)
    bg = (0.0, 0.0, 0.0)
    def three():
        rgb = C.composite_fwd(color, dens, delt, None, None, offs, R, bg)[0]
        l, g = C.rgb_loss(rgb, gts, 'huber')
        return C.composite_bwd(g, None, None, color, dens, delt, None, None, offs, bg)
    def one():
        return C.composite_loss(color, dens, delt, offs, R, bg, gts, 'huber')
    for name, fn in (("three launches", three), ("one launch", one)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        print(f"R={R} S={S}: {name:15s} {sorted(ts)[10]:8.1f} us", flush=True)
