#!/usr/bin/env python3
"""Training-step timing at a BASELINE configuration — single GPU, or data-parallel over N GPUs of one node
(BASELINE configs[4] = C5: 384^3 CT, 2x512^2 DRR, batch 32 over 8 GPUs, bf16 convs + fp32 warp, NCC loss backward).

  python tools/train_bench.py [--config c3] [--steps 5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         tools/train_bench.py --gpus N --config c5 --conv-dtype bf16 --grad-dtype bf16

One step = model(input) -> SubspaceLoss -> backward -> [gradient all-reduce] -> Adam.step (the reference's single-GPU
`step()`, RegistrationNet.py:389-406 / main.py:108-111, with the one collective data parallelism needs: SURVEY 8e).
N > 1: one process per GPU, B registrations per rank (weak scaling), gradients averaged by parallel.GradientAllReduce
(flat buckets; the 52 MB FC-head bucket goes out from an autograd hook underneath the conv backward kernels).
Rank 0 prints ONE JSON line {"metric": "training samples/s", "value": world*B*steps/time, ...} (time = MAX over ranks
between barrier+synchronize fences) followed by the per-kernel table of its own step (HIP events, one line per kernel).
`--dry-run`: the same control path on CPU over gloo with a stand-in module (no HIP kernel runs): rendezvous, bucket
all-reduce from the hooks, fences, max-over-ranks, the single line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {"c1": dict(n=64, P=2, R=64, B=1, L=56), "c2": dict(n=128, P=2, R=128, B=4, L=56),
           "c3": dict(n=256, P=2, R=256, B=8, L=56),
           "c5": dict(n=384, P=2, R=512, B=4, L=56),    # C5 per GPU: batch 32 over 8 GPUs
           # the reference's OWN shipped training configuration (/root/reference/cur_task_setting.json:30,56-57: batch_size 30,
           # drr_feature_num 4, latent_dim 56; 160^3 hard-coded at models/LiftRegDeformSubspaceBackproj.py:36; detector int(1.5*160))
           # consumed by main.py -> RegistrationNet.step (networks/RegistrationNet.py:389-406)
           "native160": dict(n=160, P=4, R=240, B=30, L=56)}


def vs_fp32(net, inp, how, c, dev):
    """Sample 0's forward of the (bf16) model `net` against fp32 arithmetic with the same weights and basis.  Checker only,
    after the timed region.  how = "cpu": oracle/ref_ops.model_forward (the reference's ATen op sequence on the host);
    "hip": this library's fp32 forward (conv_dtype fp32)."""
    one = {k: v[:1].contiguous() for k, v in inp.items()}
    with torch.no_grad():
        net.eval()
        out = net(one)
        g = {k: out[k].detach().float() for k in ("params", "pca_coefs", "warped")}
        del out
        if how == "cpu":
            from oracle import ref_ops as ro
            sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
            ref = ro.model_forward(sd, {k: v.cpu() for k, v in one.items()}, net.pca_vectors_LxM.float().cpu(),
                                   net.pca_mean.float().cpu(), conv_dtype="fp32")
            g = {k: v.cpu() for k, v in g.items()}
            against = "oracle/ref_ops.model_forward(conv_dtype='fp32') on the host — the reference's fp32 arithmetic"
        else:
            from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
            n = c["n"]
            pca_dtype = getattr(net, "pca_dtype", "fp32")
            ref_net = model([n, n, n], {"drr_feature_num": c["P"], "latent_dim": c["L"], "pca_path": "synthetic:2021",
                                        "conv_dtype": "fp32", "pca_dtype": pca_dtype}).to(dev).eval()
            ref_net.load_state_dict(net.state_dict(), strict=True)
            ref = ref_net(one)
            against = ("this library's fp32 HIP forward with the same weights and basis (the fp32 path is within 1e-9 of the "
                       "CPU oracle: bench.py parity_vs_cpu); the CPU-oracle form of this record: --vs-fp32 cpu")
        net.train()
        dd = (g["params"] - ref["params"].float()).abs()
        cs = float(ref["pca_coefs"].abs().max())
        return {"against": against, "max_abs_disp": float(dd.max()), "mean_abs_disp": float(dd.mean()),
                "disp_scale": float(ref["params"].abs().max()),
                "max_rel_disp": float(dd.max()) / max(float(ref["params"].abs().max()), 1e-30),
                "max_rel_coefs": float((g["pca_coefs"] - ref["pca_coefs"].float()).abs().max()) / max(cs, 1e-30),
                "max_abs_warped": float((g["warped"] - ref["warped"].float()).abs().max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pca-dtype", default="fp32", choices=("fp32", "bf16"),
                    help="bf16: the PCA basis stored as bfloat16 in HBM (opt-in, not the headline configuration)")
    ap.add_argument("--conv-dtype", default="fp32", choices=("fp32", "bf16"),
                    help="bf16: bf16 forward of the conv blocks, fp32 gradients of that arithmetic (config C5)")
    ap.add_argument("--grad-dtype", default="fp32", choices=("fp32", "bf16"),
                    help="bf16 (with --conv-dtype bf16): pre-activation gradients between the blocks stored as bf16")
    ap.add_argument("--ddp", action="store_true", help="use the flat gradient buckets at world size 1 too (A/B aid)")
    ap.add_argument("--same-data", action="store_true",
                    help="every rank trains on the SAME batch (test hook: the averaged gradient then equals one rank's, "
                         "so the loss trajectory must equal the single-process run's)")
    ap.add_argument("--no-kernel-table", action="store_true")
    ap.add_argument("--model-opt", action="append", default=[], metavar="KEY=BOOL",
                    help="a boolean model option, e.g. reg_in_coef_space=false ncc_grad_via_moments=false (A/B aid)")
    ap.add_argument("--adam-foreach", action="store_true", help="torch's default (multi-pass) Adam instead of fused=True (A/B aid)")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo control-path self-test with a stand-in module")
    ap.add_argument("--batch", type=int, default=None, help="registrations per rank (default: the configuration's)")
    ap.add_argument("--ramp-seconds", type=float, default=1.0, help="untimed seconds of the same step before the warm-up (clock ramp)")
    ap.add_argument("--vs-fp32", nargs="?", const="hip", default=None, choices=("hip", "cpu"),
                    help="bf16 lines, after the timed region: sample 0's forward against the fp32 arithmetic of the reference — "
                         "`cpu`: oracle/ref_ops.model_forward on the host (the checker; ~40 s at C5), `hip` (default): this "
                         "library's fp32 path with the same weights and basis (itself within 1e-9 of the CPU oracle at C3: "
                         "bench.py parity_vs_cpu) — the record says which")
    a = ap.parse_args()
    c = CONFIGS[a.config]
    n, P, R, B, L = c["n"], c["P"], c["R"], c["B"], c["L"]
    if a.batch:
        B = a.batch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        sys.exit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    backend = "gloo" if a.dry_run else os.environ.get("LIFTREG_BENCH_BACKEND", "nccl")   # nccl == RCCL on ROCm
    dist = None
    if a.dry_run:
        dev = torch.device("cpu")
    else:
        if backend != "nccl":       # test hook: several gloo ranks share one GPU on a 1-GPU box
            local %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from liftreg_amd.parallel import GradientAllReduce

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    torch.manual_seed(2021)                     # identical initial weights on every rank (as DDP's broadcast would give)
    if a.dry_run:
        net = torch.nn.Sequential()
        net.add_module("encoders", torch.nn.ModuleList([torch.nn.Linear(16, 16) for _ in range(6)] +
                                                       [torch.nn.Sequential(torch.nn.Linear(16, 4))]))
        g = torch.Generator().manual_seed(2021 if a.same_data else 2021 + rank)
        xin = torch.randn(B, 16, generator=g)

        def loss_of(ep):
            h = xin
            for m in net.encoders:
                h = torch.tanh(m(h))
            return (h ** 2).mean()
    else:
        from liftreg_amd import ops
        from liftreg_amd.losses.SubspaceLoss import loss as SubspaceLoss
        from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
        from liftreg_amd.utils.sdct_projection_utils import scan_poses
        net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:2021",
                                "conv_dtype": a.conv_dtype, "grad_dtype": a.grad_dtype,
                                "pca_dtype": a.pca_dtype,
                                **{k: (v.lower() in ("1", "true", "yes")) for k, v in (kv.split("=", 1) for kv in a.model_opt)}}).to(dev).train()
        crit = SubspaceLoss({"initial_reg_factor": 0.01, "min_reg_factor": 0.01, "reg_factor_decay_from": 2})
        crit.sim.check_nan = False
        g = torch.Generator(device=dev)
        g.manual_seed(2021 if a.same_data else 2021 + rank)
        rnd = lambda *s: torch.rand(*s, generator=g, device=dev) * 2 - 1
        poses = scan_poses(30, P, n).astype(np.float32)
        if a.vs_fp32:      # the accuracy record needs a REGISTRATION (bench.py's phantom pair and its DRRs): on uniform random
            import bench   # volumes the coefficients are ~0 and every relative figure is meaningless; timings do not depend on the data
            inp = bench.synth_inputs(dict(n=n, P=P, R=R, B=B, L=L), dev, seed=2021 if a.same_data else 2021 + rank)
        else:
            inp = {"source": rnd(B, 1, n, n, n), "target": rnd(B, 1, n, n, n), "target_proj": rnd(B, P, R, R),
                   "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}

        def loss_of(ep):
            out = net(inp)
            out["epoch"] = ep
            return crit(out)["total_loss"]

    # RegistrationNet.py:245 (Adam, eps 1e-5).  fused=True is torch's own single-kernel implementation of the same update
    # (one pass over p, g, m, v instead of the default's ~8 elementwise passes); --adam-foreach selects the default
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, eps=1e-5, **({} if (a.adam_foreach or a.dry_run) else {"fused": True}))
    ddp = GradientAllReduce(net) if (world > 1 or a.ddp) else None

    def step(ep):
        if ddp is not None:
            ddp.zero_grad()                      # one memset per bucket; .grad stays a view into its bucket
        else:
            opt.zero_grad(set_to_none=True)
        l = loss_of(ep)
        l.backward()
        if ddp is not None:
            ddp.finish()                         # wait for the bucket all-reduces (launched from hooks), average
        opt.step()
        return l.detach()

    vs = None
    if a.vs_fp32 and rank == 0 and not a.dry_run:      # at the INITIAL weights (seed 2021), before any optimizer step: a reproducible record
        vs = vs_fp32(net, inp, a.vs_fp32, dict(n=n, P=P, L=L), dev)
        torch.cuda.empty_cache()
    losses = []
    if not a.dry_run and a.ramp_seconds > 0:      # untimed clock ramp (as bench.py): a GPU that was idle starts at reduced clocks
        t_r = time.perf_counter()
        while True:
            step(0)
            torch.cuda.synchronize()
            go = torch.tensor([1.0 if time.perf_counter() - t_r < a.ramp_seconds else 0.0], device=dev)
            if dist is not None:
                dist.all_reduce(go, op=dist.ReduceOp.MIN)
            if float(go.item()) == 0.0:
                break
    for i in range(a.warmup):
        losses.append(step(i))
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        losses.append(step(a.warmup + i))
    fence()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms = elapsed / a.steps * 1e3
    losses = [float(l) for l in losses]
    assert all(np.isfinite(losses)), losses

    rows = []
    if not a.dry_run and not a.no_kernel_table:
        with ops.kernel_timer() as kt:
            step(a.warmup + a.steps)
            torch.cuda.synchronize()
        for name, rec in kt.summary().items():
            avg = float(np.mean(rec["ms"]))
            info = rec.get("info", {})
            row = {"kernel": name, "ms": round(avg, 4), "launches": len(rec["ms"])}
            if info.get("flops"):
                row["TFLOP/s"] = round(info["flops"] / avg / 1e9, 1)
            if info.get("bytes"):
                row["GB/s"] = round(info["bytes"] / avg / 1e6, 0)
            rows.append(row)
        rows.sort(key=lambda r: -r["ms"] * r["launches"])
    if rank == 0:
        print(json.dumps({
            **({"vs_fp32_reference": vs} if vs is not None else {}),
            "metric": "training samples/s", "value": round(world * B / ms * 1e3, 2), "unit": "samples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_train_step": round(ms, 3), "samples_per_s": round(world * B / ms * 1e3, 1),
            "higher_is_better": True, "scaling": "weak", "dry_run": bool(a.dry_run),
            "config": a.config, "conv_dtype": a.conv_dtype, "grad_dtype": a.grad_dtype, "pca_dtype": a.pca_dtype,
            "global_batch": world * B,
            "parallelism": (f"data parallel x{world}: {B} registrations per rank, gradient all-reduce of "
                            f"{ddp.nbytes() / 2**20:.1f} MiB in {len(ddp.buckets)} flat buckets ({backend})" if ddp is not None
                            else "single process"),
            "losses": [round(l, 6) for l in losses],
            "peak_mem_GB": None if a.dry_run else round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
        for r in rows:
            print(json.dumps(r))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
