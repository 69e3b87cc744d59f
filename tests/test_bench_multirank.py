"""CPU, world_size 2, gloo: bench.py's own main() driven through its world > 1 control flow - rendezvous from
the torch.distributed.run environment, the collective self-test and its WISP_SHARDED_OPTIM fallback (a reduce-scatter made to
fail softly), common_rays, the barrier / max-over-ranks timing of timed_steps, all_sum,
comm_summary, the reference regime, the prune timing, the PSNR pass and the JSON line of rank 0.  The HIP kernels cannot run
here: the per-ray work is the stand-in field of tests/test_distributed_gloo.py and the few runtime hooks bench.py exposes for
this purpose (device, process group, synchronise, row gather, probe raymarch) are replaced; every other line of main() runs as
the 8-GPU driver will run it."""
import io
import json
import os
import sys
import contextlib

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_distributed_gloo import _StubPipeline, _free_port, _torch_adamw_groups

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_worker(rank, world, port, out, break_reduce_scatter, precision):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.pop("WISP_SHARDED_OPTIM", None)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
    import bench
    import wisp._C as C
    C.adamw_step_groups = _torch_adamw_groups
    bench._device = lambda local: torch.device("cpu")
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
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = bench.main(["--gpus", str(world), "--steps", "4", "--warmup", "1", "--pretrain", "3", "--precision", precision,
                          "--target-samples", "65536", "--ref-target-samples", "16384", "--bank-rays", "8192", "--eval-rays", "512",
                          "--dropin-steps", "0", "--no-pmc", "--no-configs", "--no-cpu-baseline"])
    printed = [l for l in buf.getvalue().splitlines() if l.startswith("{")]
    out[rank] = {"printed": printed, "returned": res is not None, "sharded_env": os.environ.get("WISP_SHARDED_OPTIM")}


import pytest


@pytest.mark.parametrize("break_reduce_scatter,precision", [(False, "fp32"), (True, "fp32"), (False, "bf16")])
def test_bench_main_world2_gloo_prints_one_line_with_common_ray_counts(break_reduce_scatter, precision):
    """fp32: gradient all-reduce; bf16: the sharded optimizer (reduce-scatter, own slice, all-gather of the bf16 shadow) - the
    default of an amp run with more than one rank, i.e. what the driver's 8-GPU bench line takes."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bench_worker, args=(world, _free_port(), out, break_reduce_scatter, precision), nprocs=world, join=True)
    res = dict(out)
    assert len(res[0]["printed"]) == 1 and res[1]["printed"] == [], "exactly one JSON line, from rank 0"
    line = json.loads(res[0]["printed"][0])
    assert line["dtype"] == ("bf16" if precision == "bf16" else "f32")
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "ray-sharded dp2"
    R = line["config"]["rays_per_step_per_gpu"]
    # rank 0 saw 8 samples per probe ray, rank 1 twelve: the common count is rank 1's (the minimum), for both regimes
    assert R == max(256, int(4096 * 65536 / (4096 * 12)))
    assert line["reference_regime"]["rays_per_step_per_gpu"] == max(256, int(4096 * 16384 / (4096 * 12)))
    assert abs(line["value"] - R * 4 * 2 / (line["ms_per_step"] * 4e-3)) <= 1e-6 * line["value"]     # whole-job rays / max-over-ranks time
    comm = line["comm"]
    assert comm is not None and comm["selftest"]["rccl_ranks"] == 2 and comm["selftest"]["allreduce_ok"] is True
    # a reduce-scatter that fails softly: the self-test reports it and main() pins the all-reduce path on BOTH ranks
    assert comm["selftest"]["sharded_path_ok"] is (not break_reduce_scatter)
    want_env = "0" if break_reduce_scatter else None
    assert res[0]["sharded_env"] == want_env and res[1]["sharded_env"] == want_env
    if break_reduce_scatter:
        assert "reduce_scatter_tensor is not available" in comm["selftest"]["sharded_path_error"]
    assert comm["grad_bytes_on_the_wire_per_step"] > 0
    assert comm["optimizer_path"].startswith("sharded" if precision == "bf16" else "all-reduce")
    assert line["psnr_db"] is not None and line["prune"]["ms"] >= 0.0
