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


def _launch(nproc, extra, env=None, timeout=600, script="bench.py", full=False):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, script), "--gpus", str(nproc)] + extra
    e = dict(os.environ)
    e.update(env or {})
    e["OMP_NUM_THREADS"] = "2"
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, f"exactly one JSON line expected from rank 0, got {len(lines)}:\n{r.stdout[-2000:]}"
    if script == "bench.py":     # the final line is the LAST stdout line and fits the driver's tail with room to spare
        assert r.stdout.rstrip("\n").splitlines()[-1] == lines[0] and len(lines[0]) < 4096, len(lines[0])
    if full:                     # the verbose record of the same run: the first `#full` line
        fl = [l[len("#full "):] for l in r.stdout.splitlines() if l.startswith("#full ")]
        assert fl, r.stdout[-2000:]
        return json.loads(fl[0])
    return json.loads(lines[0])


def test_two_rank_control_path_dry_run():
    d = _launch(2, ["--steps", "4", "--warmup", "1", "--dry-run"])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["dry_run"] is True and d["value"] is None
    assert d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["ms_per_step"] >= 10.0 * 0.95          # rank 1 sleeps 10 ms per step: the line carries the MAX over ranks
    d = _launch(2, ["--steps", "2", "--warmup", "0", "--dry-run", "--shard", "slab"])
    assert d["scaling"] == "strong" and "slab" in d["config"]["parallelism"]


def test_training_two_rank_control_path_dry_run():
    """tools/train_bench.py as the driver would launch C5's data-parallel training step: rendezvous, the flat-bucket
    gradient all-reduce fired from the autograd hooks (parallel.GradientAllReduce), fences, MAX over ranks, one line.
    With the SAME batch on both ranks the averaged gradient is one rank's gradient: the Adam trajectory must equal the
    single-process one; with different batches it must not."""
    tb = os.path.join("tools", "train_bench.py")
    one = _launch(1, ["--dry-run", "--config", "c2", "--steps", "3", "--warmup", "1"], script=tb)
    same = _launch(2, ["--dry-run", "--config", "c2", "--steps", "3", "--warmup", "1", "--same-data"], script=tb)
    diff = _launch(2, ["--dry-run", "--config", "c2", "--steps", "3", "--warmup", "1"], script=tb)
    assert same["n_gpus"] == 2 and same["global_batch"] == 2 * one["global_batch"] and same["scaling"] == "weak"
    assert "data parallel x2" in same["parallelism"] and "2 flat buckets" in same["parallelism"]
    assert same["losses"] == pytest.approx(one["losses"], rel=1e-6)
    assert diff["losses"][-1] != pytest.approx(one["losses"][-1], rel=1e-6)


def test_final_line_fits_the_drivers_stdout_tail():
    """the driver keeps an 8 KB tail of stdout and parses the last line.  A recorded full record
    (tests/golden/bench_full_stub.json: round 5's 20.8 KB line + two more training children = six extras) goes through
    bench.py's own emit(): the verbose records land on `#full` lines, the LAST line is < 4 KB, parses, and still carries the
    contract's keys, `roofline`, `roofline_backproject`, `cpu_baseline`, `parity_vs_cpu` and one short record per extra."""
    stub = os.path.join(ROOT, "tests", "golden", "bench_full_stub.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--stub-full", stub, "--steps", "2", "--warmup", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.rstrip("\n").splitlines()
    last = out[-1]
    assert len(last) < 4096, len(last)
    d = json.loads(last)
    with open(stub) as fh:
        src = json.load(fh)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_backproject", "cpu_baseline", "parity_vs_cpu", "extra_lines"):
        assert k in d, k
    assert d["value"] == pytest.approx(src["value"], rel=1e-4) and d["dtype"] == "f32" and "256^3" in d["config"]["workload"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"] and k in d["roofline_backproject"], k
    assert d["roofline"]["frac"] == pytest.approx(src["roofline"]["frac"], rel=1e-4) and len(d["roofline"].get("note", "")) <= 120
    assert set(d["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert set(d["extra_lines"]) == set(src["extra_lines"]) and len(d["extra_lines"]) == 6
    for name, e in d["extra_lines"].items():
        assert e["value"] == pytest.approx(src["extra_lines"][name]["value"], rel=1e-4) and e["ms_per_step"] > 0 and e["steps"] > 0
    assert d["extra_lines"]["c4_bf16"]["vs_fp32_reference"]["max_abs_disp"] < 1e-4
    # every measured line's verbose record is on its own earlier line
    names = [json.loads(l[len("#full "):])["name"] for l in out[:-1] if l.startswith("#full ")]
    assert names == ["dry_run"] + list(src["extra_lines"])
    # an oversized record degrades by dropping optional keys, never the contract's
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    big = dict(src)
    big["extra_lines"] = {f"line{i}": v for i, v in enumerate(list(src["extra_lines"].values()) * 2)}
    line = bench.compact_line(big)
    assert len(line) <= bench.LINE_BUDGET and json.loads(line)["roofline"]["frac"] > 0 and json.loads(line)["cpu_baseline"]["cores"] == 256


def test_wrong_world_size_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], cwd=ROOT,
                       capture_output=True, text=True, timeout=120, env={**os.environ, "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "needs torch.distributed.run" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_replicas_and_slab():
    env = {"LIFTREG_BENCH_BACKEND": "gloo"}
    one = _launch(1, ["--config", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-drr"], env, full=True)
    rep = _launch(2, ["--config", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-drr"], env, full=True)
    assert rep["n_gpus"] == 2 and rep["scaling"] == "weak" and rep["config"]["global_batch"] == 2 * one["config"]["global_batch"]
    assert rep["value"] > 0 and "replicas x2" in rep["config"]["parallelism"]
    slab = _launch(2, ["--config", "c1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-drr", "--shard", "slab"], env, full=True)
    assert slab["n_gpus"] == 2 and slab["scaling"] == "strong" and slab["config"]["global_batch"] == one["config"]["global_batch"]
    assert "z-slab x2" in slab["config"]["parallelism"] and slab["value"] > 0
    # the same batch (seed 2021), sharded over two ranks: the NCC from all-reduced slab moments = the unsharded NCC
    assert abs(slab["ncc_loss"] - one["ncc_loss"]) < 1e-6, (slab["ncc_loss"], one["ncc_loss"])


@pytest.mark.gpu
def test_slab_mode_reports_the_sharded_projector_leg():
    """`--shard slab` without --no-drr: the partial DRRs of the two ranks' slabs, summed by the all-reduce, equal the
    single-GPU projector's images (north star: "all-reduce of slab-boundary partial sums")."""
    env = {"LIFTREG_BENCH_BACKEND": "gloo"}
    slab = _launch(2, ["--config", "c1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--shard", "slab", "--ramp-seconds", "0"], env, full=True)
    leg = slab["drr_forward_sharded"]
    assert leg["max_rel_vs_unsharded"] < 1e-5 and leg["volumes_per_s"] > 0 and leg["allreduce_bytes"] == 4 * 1 * 2 * 64 * 64
    assert slab["ramp_seconds"] == 0 and slab["ramp_steps"] == 0


@pytest.mark.gpu
def test_training_two_ranks_on_one_gpu_same_data_equals_single_process():
    """The data-parallel training step on the HIP kernels (C1 shape, two gloo ranks sharing the GPU): same batch on both
    ranks -> bucketed, averaged gradients == one rank's -> the loss trajectory of the single-process run."""
    env = {"LIFTREG_BENCH_BACKEND": "gloo"}
    tb = os.path.join("tools", "train_bench.py")
    args = ["--config", "c1", "--steps", "3", "--warmup", "1", "--no-kernel-table"]
    one = _launch(1, args, env, script=tb)
    two = _launch(2, args + ["--same-data"], env, script=tb)
    assert two["n_gpus"] == 2 and two["global_batch"] == 2 and "data parallel x2" in two["parallelism"]
    assert two["losses"] == pytest.approx(one["losses"], rel=2e-4, abs=1e-6), (one["losses"], two["losses"])


def test_sclk_sampler_parses_sysfs_and_degrades_to_none(tmp_path, monkeypatch):
    """bench.SclkSampler: reads the starred level of a pp_dpm_sclk file; without the file (this container, or a driver that
    does not expose it) the roofline's sclk_mhz is None and nothing else changes."""
    import importlib.util
    import os
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    s = bench.SclkSampler(0)
    if s.path is None:                       # no GPU here
        with s:
            pass
        assert s.median() is None
    f = tmp_path / "pp_dpm_sclk"
    f.write_text("0: 500Mhz\n1: 2245Mhz *\n2: 2400Mhz\n")
    s = bench.SclkSampler(0)
    s.path = str(f)
    with s:
        time.sleep(0.2)
    assert s.median() == 2245.0
