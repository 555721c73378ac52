#!/usr/bin/env python3
"""Cost of the z-slab decomposition itself: the slab-sharded forward with N virtual ranks on ONE GPU (LocalComm: the
halo exchange is a device copy) against the unsharded forward of the same batch.  Development aid.

  python tools/shard_bench.py [--n 256] [--world 4] [--batch 8] [--views 2] [--conv-dtype fp32|bf16] [--graph]
  python tools/shard_bench.py --procs 4 [--n 256] [--batch 8] …       # N PROCESSES sharing cuda:0 through parallel.DistComm

`--graph`: both forwards are also captured into HIP graphs and replayed — the GPU-side time of the sum of slabs without the
one-process host cost of issuing every virtual rank's launches (what the wall time of the eager proxy is bound by at 8 ranks).
`--procs N`: N fresh child processes (started BEFORE anything touches the GPU here), one rank each, all on cuda:0, joined by
torch.distributed: `nccl` when RCCL accepts several ranks on one device (it refuses duplicates: probed with a short timeout),
else `gloo` with DistComm's host staging.  Every rank asserts its slab == the rows of its own unsharded forward, bit for bit,
and reports its wall time between barriers and its event-bracketed kernel time (both INFLATED by the other ranks' kernels
sharing the device: the numbers that transfer to N GPUs are the equality and the per-rank kernel list, not the wall time)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from liftreg_amd import parallel as par  # noqa: E402
from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model  # noqa: E402
from liftreg_amd.utils.sdct_projection_utils import scan_poses  # noqa: E402


def _build(a, dev):
    n, P, B = a.n, a.views, a.batch
    torch.manual_seed(1)
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": 56, "pca_path": "synthetic:1", "conv_dtype": a.conv_dtype}).to(dev).eval()
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    poses = scan_poses(30, P, n).astype(np.float32)
    inp = {"source": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
           "target": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
           "target_proj": torch.rand((B, P, n, n), generator=g, device=dev) * 2 - 1,
           "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}
    return net, inp


def launch_procs(a):
    """Parent of --procs N: never touches the GPU; starts N workers (RANK / WORLD_SIZE / MASTER_* in their environment), `nccl`
    first under a short timeout when the backend is `auto`, then `gloo`; prints the workers' lines and one summary line."""
    import socket
    import subprocess

    def port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]

    def run(backend, timeout):
        env0 = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port()), WORLD_SIZE=str(a.procs), SHARD_BACKEND=backend,
                    OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        argv = [sys.executable, os.path.abspath(__file__), "--worker", "--procs", str(a.procs), "--n", str(a.n), "--batch", str(a.batch),
                "--views", str(a.views), "--conv-dtype", a.conv_dtype, "--iters", str(a.iters)] + (["--exchange"] if a.exchange else [])
        ps = [subprocess.Popen(argv, env=dict(env0, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
              for r in range(a.procs)]
        outs, ok = [], True
        t_end = time.time() + timeout
        for p in ps:
            try:
                o, e = p.communicate(timeout=max(1.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                ok = False
                o, e = "", "timeout"
            outs.append((p.returncode, o, e))
        for p in ps:               # exactly the children started here
            if p.poll() is None:
                p.kill()
                p.wait()
        return ok and all(rc == 0 for rc, _, _ in outs), outs

    import time
    tried = []
    for backend in (("nccl", "gloo") if a.backend == "auto" else (a.backend,)):
        ok, outs = run(backend, 90 if (backend == "nccl" and a.backend == "auto") else 900)
        tried.append(backend)
        if ok:
            break
        print(json.dumps({"backend": backend, "ok": False, "why": (outs[0][2] or "")[-300:].replace("\n", " ")}), flush=True)
    if not ok:
        sys.exit("every backend failed: " + ", ".join(tried))
    rows = [json.loads(ln) for _, o, _ in outs for ln in o.splitlines() if ln.startswith("{")]
    for r in rows:
        print(json.dumps(r))
    print(json.dumps({"procs": a.procs, "backend": backend, "n": a.n, "batch": a.batch, "views": a.views, "conv_dtype": a.conv_dtype,
                      "slabs_equal_unsharded": all(r["slab_equals_unsharded"] for r in rows),
                      "unsharded_ms_alone": rows[0]["unsharded_ms_rank0_alone"],
                      "sharded_wall_ms_max": max(r["sharded_wall_ms"] for r in rows),
                      "kernel_ms_per_rank": [r["kernel_ms"] for r in rows],
                      "note": "N processes share ONE GPU: wall and kernel times include the other ranks' kernels on the same device"}))


def worker(a):
    """One rank of --procs N: cuda:0, DistComm over the group, slab == rows of this process's own unsharded forward."""
    import time
    import torch.distributed as dist
    from liftreg_amd import ops
    from liftreg_amd.layers.losses import NCCLoss
    rank, world_size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("SHARD_BACKEND", "gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    net, inp = _build(a, dev)
    sim = NCCLoss(check_nan=False)
    sh = par.SlabShardedRegistration(net, par.DistComm())
    if a.exchange:
        sh.halo_free01 = False
    n = a.n
    d0, d1 = par.slab_bounds(n, world_size, rank)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        o = sh.forward([inp])[0]          # (first: collectives need every rank at the same point)
        ref = net(inp)
        ref_loss = float(sim(ref["warped"], ref["target"]))
        same = bool(torch.equal(o["pca_coefs"], ref["pca_coefs"]) and all(torch.equal(o[k], ref[k][:, :, d0:d1]) for k in ("params", "phi", "warped"))
                    and abs(float(o["sim_loss"]) - ref_loss) < 2e-7)
        del ref, o
        # the unsharded step, rank 0 ALONE on the device (the others wait at the barrier)
        t_un = None
        fence()
        if rank == 0:
            for _ in range(2):
                out = net(inp); sim(out["warped"], out["target"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                out = net(inp); sim(out["warped"], out["target"])
            torch.cuda.synchronize()
            t_un = (time.perf_counter() - t0) / a.iters * 1e3
            del out
        fence()
        for _ in range(2):
            sh.forward([inp])
        fence()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            sh.forward([inp])
        fence()
        wall = (time.perf_counter() - t0) / a.iters * 1e3
        with ops.kernel_timer() as kt:
            sh.forward([inp])
            torch.cuda.synchronize()
        fence()
    tab = {}
    for name, rec in kt.summary().items():
        key = name.split("_s")[0] if name.startswith("conv3d") else name
        tab[key] = round(tab.get(key, 0.0) + float(np.sum(rec["ms"])), 3)
    t = torch.tensor([t_un if t_un is not None else 0.0], dtype=torch.float64)
    dist.all_reduce(t)
    print(json.dumps({"rank": rank, "world": world_size, "backend": backend, "rows": [d0, d1], "slab_equals_unsharded": same,
                      "unsharded_ms_rank0_alone": round(float(t.item()), 3), "sharded_wall_ms": round(wall, 3),
                      "kernel_ms": round(sum(tab.values()), 3), "kernels": tab}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not same:
        sys.exit(f"rank {rank}: the slab differs from the unsharded forward")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--conv-dtype", default="fp32")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--exchange", action="store_true", help="blocks 0/1 of the general / bf16 path with the round-3 halo exchange (halo_free01 = False)")
    ap.add_argument("--max-overhead", type=float, default=None, help="exit non-zero when sum_of_slabs / unsharded - 1 exceeds this")
    ap.add_argument("--graph", action="store_true", help="also time HIP-graph replays of both forwards (GPU-side time, no host)")
    ap.add_argument("--procs", type=int, default=0, help="N processes sharing cuda:0 through DistComm (see the docstring)")
    ap.add_argument("--backend", default="auto", choices=("auto", "nccl", "gloo"), help="--procs: process-group backend")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.procs and not a.worker:
        return launch_procs(a)
    if a.worker:
        return worker(a)
    dev = torch.device("cuda:0")
    n, P, B = a.n, a.views, a.batch
    net, inp = _build(a, dev)
    sh = par.SlabShardedRegistration(net, par.LocalComm(a.world))
    if a.exchange:
        sh.halo_free01 = False

    def timeit(fn):
        with torch.no_grad():
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iters):
                fn()
            e.record()
            torch.cuda.synchronize()
        return s.elapsed_time(e) / a.iters

    from liftreg_amd import ops

    def table(fn):
        with torch.no_grad(), ops.kernel_timer() as kt:
            fn()
            torch.cuda.synchronize()
        out = {}
        for name, rec in kt.summary().items():
            key = name.split("_s")[0] if name.startswith("conv3d") else name
            out[key] = round(out.get(key, 0.0) + float(np.sum(rec["ms"])), 3)
        return out

    from liftreg_amd.layers.losses import NCCLoss
    sim = NCCLoss(check_nan=False)

    def full():          # the unsharded step INCLUDING the similarity (the sharded forward computes it from all-reduced moments)
        out = net(inp)
        return sim(out["warped"], out["target"])

    # the decomposition must not change a bit: every rank's slab of the outputs equals the rows of the unsharded forward
    with torch.no_grad():
        ref = net(inp)
        ref_loss = float(sim(ref["warped"], ref["target"]))
        for r, o in enumerate(sh.forward([inp] * a.world)):
            d0, d1 = par.slab_bounds(n, a.world, r)
            assert torch.equal(o["pca_coefs"], ref["pca_coefs"]), f"rank {r}: coefficients differ"
            for k in ("params", "phi", "warped"):
                assert torch.equal(o[k], ref[k][:, :, d0:d1]), f"rank {r}: {k} differs"
            assert abs(float(o["sim_loss"]) - ref_loss) < 2e-7, f"rank {r}: loss differs"
        del ref
    if os.environ.get("SHARD_BENCH_TABLE"):
        print("unsharded", json.dumps(table(full)))
        print("sharded  ", json.dumps(table(lambda: sh.forward([inp] * a.world))))
    t_full = timeit(full)
    t_shard = timeit(lambda: sh.forward([inp] * a.world))
    g_full = g_shard = None
    if a.graph:
        def graphed(fn):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                fn(); fn()
            torch.cuda.current_stream().wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(gr):
                keep = fn()
            gr._keep = keep
            return gr
        gf = graphed(full)
        g_full = timeit(gf.replay)
        del gf
        gs = graphed(lambda: sh.forward([inp] * a.world))
        g_shard = timeit(gs.replay)
        del gs
    # the library kernels' own time (event-bracketed one by one): what the decomposition costs the GPUs, without the host — one
    # Python process issues every virtual rank's launches here, and at 8 ranks that, not the GPU, sets the wall time above
    k_full = sum(table(full).values())
    k_shard = sum(table(lambda: sh.forward([inp] * a.world)).values())
    print(json.dumps({"n": n, "views": P, "batch": B, "world": a.world, "conv_dtype": a.conv_dtype,
                      "unsharded_ms": round(t_full, 3), "sum_of_slabs_ms": round(t_shard, 3),
                      "decomposition_overhead": round(t_shard / t_full - 1, 3),
                      "kernels_unsharded_ms": round(k_full, 3), "kernels_slabs_ms": round(k_shard, 3),
                      "kernel_overhead": round(k_shard / k_full - 1, 3),
                      **({"graph_unsharded_ms": round(g_full, 3), "graph_slabs_ms": round(g_shard, 3),
                          "graph_overhead": round(g_shard / g_full - 1, 3)} if g_full else {}),
                      "slabs_equal_unsharded": True}))
    if a.max_overhead is not None and t_shard / t_full - 1 > a.max_overhead:
        sys.exit(f"decomposition overhead {t_shard / t_full - 1:.3f} above --max-overhead {a.max_overhead}")


if __name__ == "__main__":
    main()
