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

from test_distributed_gloo import _free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_worker(rank, world, port, out, break_reduce_scatter, precision, extra=()):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.pop("WISP_SHARDED_OPTIM", None)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "kaolin-wisp_amd")]
    import bench
    import bench_cpu_standin
    bench_cpu_standin.install(bench, break_reduce_scatter=break_reduce_scatter)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = bench.main(["--gpus", str(world), "--steps", "4", "--warmup", "1", "--pretrain", "3", "--precision", precision,
                          "--target-samples", "16384", "--large-target-samples", "65536", "--bank-rays", "8192", "--eval-rays", "512",
                          "--dropin-steps", "0", "--no-pmc", "--no-configs", "--no-cpu-baseline"] + list(extra))
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
    assert R == max(256, int(4096 * 16384 / (4096 * 12)))
    assert line["large_batch_regime"]["rays_per_step_per_gpu"] == max(256, int(4096 * 65536 / (4096 * 12)))
    assert abs(line["large_batch_regime"]["lr_scale"] - 2.0) < 1e-9                 # sqrt(65536 / 16384)
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


def test_bench_main_world2_gloo_strong_scaling_splits_the_global_batch():
    """`--scaling strong` keeps the GLOBAL batch at --target-samples and gives every rank 1/N of it (the
    reference trainer's batch stays the reference's as GPUs are added): half the rays per rank of the weak run above, both
    regimes, and the line says so."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bench_worker, args=(world, _free_port(), out, False, "fp32", ("--scaling", "strong")), nprocs=world, join=True)
    line = json.loads(dict(out)[0]["printed"][0])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    cfg = line["config"]
    assert cfg["target_samples_per_step"] == 16384 // 2 and cfg["global_target_samples_per_step"] == 16384
    assert cfg["rays_per_step_per_gpu"] == max(256, int(4096 * (16384 // 2) / (4096 * 12)))
    assert line["large_batch_regime"]["rays_per_step_per_gpu"] == max(256, int(4096 * (65536 // 2) / (4096 * 12)))
    R = cfg["rays_per_step_per_gpu"]
    assert abs(line["value"] - R * 4 * 2 / (line["ms_per_step"] * 4e-3)) <= 1e-6 * line["value"]      # still the whole job's rays / s


STANDIN_ARGS = ["--steps", "3", "--warmup", "1", "--pretrain", "2", "--precision", "fp32", "--target-samples", "16384",
                "--large-target-samples", "65536", "--bank-rays", "8192", "--eval-rays", "512", "--dropin-steps", "0", "--no-pmc",
                "--no-configs", "--no-cpu-baseline"]


def _run_standin(gpus, devices):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(WISP_STANDIN_DEVICES=str(devices), OMP_NUM_THREADS="2")
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_cpu_standin.py"), "--gpus", str(gpus)] + STANDIN_ARGS,
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def test_bench_gpus_2_without_a_launcher_starts_its_own_two_ranks():
    """VERDICT r4 missing-1: a plain `python bench.py --gpus 2` (no torch.distributed.run around it) used to fall through to ONE
    rank and print n_gpus: 1.  It now re-executes itself under torch.distributed.run with 2 ranks: one JSON line, n_gpus 2, and
    the collective self-test saw 2 ranks."""
    r = _run_standin(2, 2)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "ray-sharded dp2"
    assert line["comm"]["selftest"]["rccl_ranks"] == 2 and line["comm"]["selftest"]["allreduce_ok"] is True
    assert line["steps"] == 3 and line["scaling"] == "weak"


def test_bench_gpus_beyond_the_visible_devices_fails_loudly():
    r = _run_standin(4, 2)
    assert r.returncode != 0 and "only 2 GPU(s) visible" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
