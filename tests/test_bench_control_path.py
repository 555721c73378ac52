"""bench.py's N>1 control path (the driver launches it as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py --gpus N …`).

CPU (`--dry-run`): rendezvous, barrier-fenced timed region, MAX over ranks and the single JSON line, no GPU work.
GPU (-m gpu, one GPU shared by two gloo ranks through the LIFTREG_BENCH_BACKEND test hook): the real replicas line and
the z-slab sharded line (`--shard slab`: halo p2p, feature all-gather, NCC-moment all-reduce) at the small C1 shape —
and the sharded NCC equals the unsharded run's."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(nproc, extra, env=None, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + extra
    e = dict(os.environ)
    e.update(env or {})
    e["OMP_NUM_THREADS"] = "2"
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected from rank 0, got {len(lines)}:\n{r.stdout[-2000:]}"
    return json.loads(lines[0])


def test_two_rank_control_path_dry_run():
    d = _launch(2, ["--steps", "4", "--warmup", "1", "--dry-run"])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["dry_run"] is True and d["value"] is None
    assert d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["ms_per_step"] >= 10.0 * 0.95          # rank 1 sleeps 10 ms per step: the line carries the MAX over ranks
    d = _launch(2, ["--steps", "2", "--warmup", "0", "--dry-run", "--shard", "slab"])
    assert d["scaling"] == "strong" and "slab" in d["config"]["parallelism"]


def test_wrong_world_size_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], cwd=ROOT,
                       capture_output=True, text=True, timeout=120, env={**os.environ, "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "needs torch.distributed.run" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_replicas_and_slab():
    env = {"LIFTREG_BENCH_BACKEND": "gloo"}
    one = _launch(1, ["--config", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-drr"], env)
    rep = _launch(2, ["--config", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-drr"], env)
    assert rep["n_gpus"] == 2 and rep["scaling"] == "weak" and rep["config"]["global_batch"] == 2 * one["config"]["global_batch"]
    assert rep["value"] > 0 and "replicas x2" in rep["config"]["parallelism"]
    slab = _launch(2, ["--config", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-drr", "--shard", "slab"], env)
    assert slab["n_gpus"] == 2 and slab["scaling"] == "strong" and slab["config"]["global_batch"] == one["config"]["global_batch"]
    assert "z-slab x2" in slab["config"]["parallelism"] and slab["value"] > 0
    # the same batch (seed 2021), sharded over two ranks: the NCC from all-reduced slab moments = the unsharded NCC
    assert abs(slab["ncc_loss"] - one["ncc_loss"]) < 1e-6, (slab["ncc_loss"], one["ncc_loss"])
