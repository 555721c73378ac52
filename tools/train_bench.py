#!/usr/bin/env python3
"""Training-step timing at a BASELINE configuration (development aid; bench.py is the inference contract).

  python tools/train_bench.py [--config c3] [--steps 5]
One step = model(input) → SubspaceLoss → backward → Adam.step (RegistrationNet.py:389-406).
Prints ms/step and the per-kernel table (HIP events around every launch, forward and backward).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from liftreg_amd import ops  # noqa: E402
from liftreg_amd.losses.SubspaceLoss import loss as SubspaceLoss  # noqa: E402
from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model  # noqa: E402
from liftreg_amd.utils.sdct_projection_utils import scan_poses  # noqa: E402

CONFIGS = {"c1": dict(n=64, P=2, R=64, B=1, L=56), "c2": dict(n=128, P=2, R=128, B=4, L=56),
           "c3": dict(n=256, P=2, R=256, B=8, L=56),
           "c5": dict(n=384, P=2, R=512, B=4, L=56)}   # C5 per GPU: batch 32 over 8 GPUs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pca-dtype", default="fp32", choices=("fp32", "bf16"),
                    help="bf16: the PCA basis stored as bfloat16 in HBM (opt-in, not the headline configuration)")
    ap.add_argument("--conv-dtype", default="fp32", choices=("fp32", "bf16"),
                    help="bf16: bf16 forward of the conv blocks, fp32 gradients of that arithmetic (config C5)")
    ap.add_argument("--grad-dtype", default="fp32", choices=("fp32", "bf16"),
                    help="bf16 (with --conv-dtype bf16): pre-activation gradients between the blocks stored as bf16")
    a = ap.parse_args()
    c = CONFIGS[a.config]
    n, P, R, B, L = c["n"], c["P"], c["R"], c["B"], c["L"]
    dev = torch.device("cuda:0")
    torch.manual_seed(2021)
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:2021",
                            "conv_dtype": a.conv_dtype, "grad_dtype": a.grad_dtype,
                            "pca_dtype": a.pca_dtype}).to(dev).train()
    crit = SubspaceLoss({"initial_reg_factor": 0.01, "min_reg_factor": 0.01, "reg_factor_decay_from": 2})
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, eps=1e-5)
    g = torch.Generator(device=dev)
    g.manual_seed(2021)
    rnd = lambda *s: torch.rand(*s, generator=g, device=dev) * 2 - 1
    poses = scan_poses(30, P, n).astype(np.float32)
    inp = {"source": rnd(B, 1, n, n, n), "target": rnd(B, 1, n, n, n), "target_proj": rnd(B, P, R, R),
           "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}

    def step(ep):
        opt.zero_grad(set_to_none=True)
        out = net(inp)
        out["epoch"] = ep
        crit.sim.check_nan = False
        l = crit(out)["total_loss"]
        l.backward()
        opt.step()
        return l

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(a.steps):
        step(i)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / a.steps
    with ops.kernel_timer() as t:
        step(0)
        torch.cuda.synchronize()
    rows = []
    for name, rec in t.summary().items():
        avg = float(np.mean(rec["ms"]))
        info = rec.get("info", {})
        row = {"kernel": name, "ms": round(avg, 4), "launches": len(rec["ms"])}
        if info.get("flops"):
            row["TFLOP/s"] = round(info["flops"] / avg / 1e9, 1)
        if info.get("bytes"):
            row["GB/s"] = round(info["bytes"] / avg / 1e6, 0)
        rows.append(row)
    rows.sort(key=lambda r: -r["ms"] * r["launches"])
    print(json.dumps({"config": a.config, "conv_dtype": a.conv_dtype, "grad_dtype": a.grad_dtype, "pca_dtype": a.pca_dtype, "ms_per_train_step": round(ms, 3), "samples_per_s": round(B / ms * 1e3, 1),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
