"""bench.py with its device-runtime hooks replaced by CPU stand-ins (gloo, CPU tensors, the stub field of
tests/test_distributed_gloo.py): `python tests/bench_cpu_standin.py --gpus 2 ...` is what `python bench.py --gpus 2 ...` does on a
2-GPU node - including bench.py starting its own ranks (it re-executes sys.argv[0], i.e. this wrapper, under
torch.distributed.run).  Test infrastructure: tests/test_bench_multirank.py runs it; nothing else does."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd"), os.path.join(ROOT, "tests")]

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402


def install(bench, visible_devices=2, break_reduce_scatter=False):
    import wisp._C as C
    from test_distributed_gloo import _StubPipeline, _torch_adamw_groups
    rank = int(os.environ.get("RANK", "0"))
    C.adamw_step_groups = _torch_adamw_groups
    bench._device = lambda local: torch.device("cpu")
    bench._device_count = lambda: visible_devices
    bench._init_dist = lambda dev: dist.init_process_group("gloo")
    bench._sync = lambda: None
    bench._gather_rows = lambda idx, tensors: [t.index_select(0, idx) for t in tensors]
    bench._initial_cells = lambda args, dev, true_cells: true_cells
    bench.build_pipeline = lambda dev, hidden, num_steps, cells: _StubPipeline(rows=64)
    # rank-dependent sample yields: common_rays must bring both ranks to the SAME ray count (the smaller one)
    bench._probe_samples = lambda pipe, probe, num_steps: 4096 * (8 + 4 * rank)
    bench._leaf_cells = lambda pipe: 1234
    if break_reduce_scatter:                   # a backend whose reduce-scatter fails softly: the self-test must catch it
        def broken(*a, **k):
            raise RuntimeError("reduce_scatter_tensor is not available (test)")
        dist.reduce_scatter_tensor = broken


if __name__ == "__main__":
    import bench
    install(bench, visible_devices=int(os.environ.get("WISP_STANDIN_DEVICES", "2")))
    bench.main()
