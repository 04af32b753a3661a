"""GPU: the RCCL backend on the one GPU this box has.  Every multi-rank test of the suite swaps RCCL for gloo (two ranks
cannot share a device under RCCL), so until this test nothing had ever initialised backend "nccl".  ONE fresh child process
runs `bench.py` as a process group of one rank with the multi-rank code path forced (`PF_BENCH_FORCE_DIST=1` ->
`puflow_amd.dist.force_collectives`): init_process_group("nccl", device_id=...), `broadcast_module`, the training step as
graph A -> eager `all_reduce` of the 806 103-float gradient bucket on the same stream -> graph B, timing barriers, the MAX
all-reduce of the elapsed time, `destroy_process_group` - and must exit 0.  No process that touched the GPU is exec'ed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "MASTER_PORT", "PF_BENCH_BACKEND", "PF_BENCH_SINGLE_DEVICE")}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", PF_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


@pytest.mark.parametrize("graph", ["1", "0"])
def test_training_step_over_rccl_one_rank(graph):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "train", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], cwd=ROOT, env=dict(_env(), PF_BENCH_GRAPH=graph), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "capture failed" not in out.stderr
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    col = rec["config"]["collectives"]
    assert col == {"backend": "nccl", "world_size": 1, "forced_one_rank_group": True}
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["loss"] == rec["loss"] and abs(rec["loss"]) < 1e3
    if graph == "1":
        assert "eager all-reduce" in rec["config"]["launch"]


def test_forced_collectives_do_not_change_the_step():
    """The forced one-rank path (gradient packing -> RCCL all-reduce -> / world -> clip + Adam) takes the same steps as the plain
    single-process path.  Under cfg.deterministic (PF_BENCH_DETERMINISTIC=1: BatchNorm statistics as exact sums, ordered gathers
    instead of float atomics; PF_BENCH_ACTNORM_INITED=1: no data-dependent ActNorm init, which the two paths run through different
    kernels) the step is bit-reproducible, so the comparison is made where it means something again: the
    loss after THREE optimisation steps and the gradient norm of the last one (round 4 had to retreat to the first step's
    gradient norm - without the switch Adam turns the atomics' rounding noise into different trajectories within a few updates)."""
    recs = []
    for force in ("1", "0"):
        env = _env()
        if force == "0":
            for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PF_BENCH_FORCE_DIST"):
                env.pop(k)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--steps", "3", "--warmup", "0",
                              "--no-cpu-baseline", "--no-grad-parity"], cwd=ROOT, env=dict(env, PF_BENCH_GRAPH="0", PF_BENCH_DETERMINISTIC="1", PF_BENCH_ACTNORM_INITED="1"),
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        recs.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1]))
    assert recs[0]["config"]["deterministic"] is True
    g0, g1 = recs[0]["grad_norm_last_step"], recs[1]["grad_norm_last_step"]
    assert g0 > 0 and abs(g0 - g1) <= 1e-6 * g1, (g0, g1)
    assert abs(recs[0]["loss"] - recs[1]["loss"]) <= 1e-6 * abs(recs[1]["loss"]), (recs[0]["loss"], recs[1]["loss"])


def test_inference_bench_over_rccl_one_rank():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2", "--no-cpu-baseline",
                          "--no-reduced"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0
    # one invocation carries both scalings and says what the communicator is
    assert set(rec["strong"]) == {"32", "256"} and all(v["value"] > 0 and v["scaling"] == "strong" for v in rec["strong"].values())
    assert rec["strong"]["32"]["patches_per_rank"] == [32]
    col = rec["config"]["collectives"]
    assert col["backend"] == "nccl" and col["communicator_ranks_seen_by_first_all_reduce"] == 1
    ar = col["gradient_bucket_all_reduce"]
    assert ar["floats"] == 806103 and 0 < ar["median_us"] < 1e5
    assert "rccl" in col and "channel_connections_by_transport" in col["rccl"], col.get("rccl")
    print("RCCL (1 rank): bucket all-reduce", ar["median_us"], "us;", col["rccl"])


@pytest.mark.parametrize("graph", ["1", "0"])
def test_syncbn_training_step_over_rccl_one_rank(graph):
    """cfg.sync_batchnorm on the fused kernels with RCCL: every BatchNorm layer's 257-double all-reduce (csrc/train_fused.hip
    stat_sync -> train_ops._sync_cb_impl -> dist.all_reduce) between two launches - eager, and CAPTURED into the step's hipGraph
    (graph = 1: graphed_train_step no longer refuses SyncBN when the backend is nccl).  One rank, multi-rank path forced."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "train", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-grad-parity"], cwd=ROOT, env=dict(_env(), PF_BENCH_GRAPH=graph, PF_BENCH_SYNCBN="1"),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "capture failed" not in out.stderr, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["config"]["sync_batchnorm"] is True and rec["config"]["collectives"]["backend"] == "nccl"
    assert rec["value"] > 0 and rec["loss"] == rec["loss"] and abs(rec["loss"]) < 1e3
    print("SyncBN step over RCCL (1 rank), graph =", graph, ":", rec["ms_per_step"], "ms")
