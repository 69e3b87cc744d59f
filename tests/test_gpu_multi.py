"""Data-parallel path on real GPUs through RCCL (`-m gpu`): the one-GPU variant (WISP_FORCE_ALLREDUCE=1: the collective,
the side stream and the parameter-ready event all run, with one rank) runs on the single-GPU box; the two-GPU variant is
skipped unless two devices are visible."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _launch(nproc, extra_env):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dp_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("DP_RESULT ")]
    assert r.returncode == 0 and lines, r.stdout[-3000:]
    return json.loads(lines[-1][len("DP_RESULT "):])


@pytest.mark.parametrize("amp", ["1", "0"])
def test_forced_allreduce_side_stream_on_one_gpu(amp):
    """One rank, RCCL all-reduce forced: reduce_and_update() runs on its side stream, the next step's raymarch overlaps it
    and wait_for_parameters() orders the hash-grid forward behind it.  A sum over one rank is the identity, so the result
    must equal the same steps without the collective up to float-atomic order - a missing stream dependency (parameters
    read before their update lands) shows up as a difference of the size of a whole optimizer step."""
    res = _launch(1, {"WISP_FORCE_ALLREDUCE": "1", "DP_AMP": amp})
    assert res["direct"] and res["pruned"] and res["finite"] and res["identical"] and res["same_tree"], res
    assert res["allreduce_numel"] < res["grad_numel"] and res["skipped_tail_zero"], res
    # not bitwise: overflowing gradient slots and the coarse levels' split flush add with float atomics (free order)
    print(res)
    assert res["rel_l2_vs_single"] < 2e-3, res


@pytest.mark.parametrize("amp", ["1", "0"])
def test_forced_sharded_optimizer_on_one_gpu(amp):
    """WISP_SHARDED_OPTIM=1 with one forced rank: RCCL reduce-scatter and all-gather (bf16 shadow with amp, fp32 master
    without), the HIP optimizer launched on a slice of the grid group plus the rows no gradient reaches, the staging copies and
    the side stream all run.  With one rank the slice is the whole window, so the result must equal the plain run like the
    forced all-reduce does; the shadow must be bf16(master) over the whole table afterwards."""
    res = _launch(1, {"WISP_FORCE_ALLREDUCE": "1", "WISP_SHARDED_OPTIM": "1", "DP_AMP": amp})
    print(res)
    assert res["sharded"] and res["plan_direct"] and res["window"] < res["grid_elems"], res      # finest level = untouched tail
    assert res["direct"] and res["pruned"] and res["finite"] and res["identical"] and res["same_tree"], res
    assert res["grads_consumed"] and res["state_only_on_owner"] and not res["stale_before_sync"], res
    if amp == "1":
        assert res["shadow_is_bf16_of_master"], res
    assert res["rel_l2_vs_single"] < 2e-3, res


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (gpurun boxes have one)")
def test_two_gpu_default_path_with_amp_is_the_sharded_one():
    """No switch set: two ranks + amp pick reduce-scatter / sharded optimizer / all-gather by themselves, the decoder group's
    all-reduce goes ahead of the grid's collective (early_reduce_decoder), and the replicas stay in lockstep."""
    res = _launch(2, {"DP_AMP": "1"})
    assert res["world"] == 2 and res["sharded"] and res["stale_before_sync"], res
    assert res["direct"] and res["pruned"] and res["finite"] and res["identical"] and res["same_tree"], res
    assert res["rel_l2_vs_single"] < 2e-2, res


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (gpurun boxes have one)")
def test_two_gpu_sharded_optimizer_stays_in_lockstep():
    """Two ranks, WISP_SHARDED_OPTIM=1: each rank updates half of the live table rows; after sync_master() the replicas hold
    bit-identical master weights and octrees (the prunes inside the run read synced weights), the shadow mirrors the master,
    optimizer state exists only on the owner, and the result agrees with a single-GPU run like the all-reduce path does."""
    res = _launch(2, {"WISP_SHARDED_OPTIM": "1", "DP_AMP": "1"})
    assert res["world"] == 2 and res["sharded"] and res["stale_before_sync"], res
    assert res["direct"] and res["pruned"] and res["finite"] and res["identical"] and res["same_tree"], res
    assert res["grads_consumed"] and res["state_only_on_owner"] and res["shadow_is_bf16_of_master"], res
    assert res["rel_l2_vs_single"] < 2e-2, res


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (gpurun boxes have one)")
def test_two_gpu_ray_sharded_training_stays_in_lockstep():
    """Two ranks over RCCL/xGMI: disjoint ray shards, one all-reduce of the flat gradient per step, identical prune draws -
    after 7 steps (two prunes) the replicas hold bit-identical parameters and octrees, and agree with a single-GPU run over
    the whole batch up to the order of the gradient sum."""
    res = _launch(2, {"DP_AMP": "1", "WISP_SHARDED_OPTIM": "0"})       # (with amp and two ranks the sharded path is the default)
    assert res["world"] == 2 and not res["sharded"] and res["direct"] and res["pruned"] and res["finite"], res
    assert res["identical"] and res["same_tree"], res
    assert res["rel_l2_vs_single"] < 2e-2, res          # bf16 forward: sample-order dependent rounding in the decoder tiles
