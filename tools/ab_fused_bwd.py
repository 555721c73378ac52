#!/usr/bin/env python3
"""A/B of lr_conv3d_dgrad_wgrad0_split_f32's tile depth (LIFTREG_FUSED_BWD_NZ: 8 quotient planes per tile, or 4 with two waves per
plane) in one process, at C3 (256^3, 3 channels, B = 8) and the reference's shipped shape (160^3, 5 channels, B = 30: always 4);
the 4-plane form's gw0 / gb0 against the 8-plane form (relative to the scale).  (Round 6 also measured the weight-gradient half on
exact bf16 splits here: 12.2 vs 6.9 ms at C3 — commit 4ae0fef, profiles/NOTES_r06.md.)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import _hip, ops, ops_bwd
dev = torch.device("cuda:0")


def run(n, cin0, B, env, ref=None, reps=8):
    for k in ("LIFTREG_FUSED_BWD_NZ",):
        os.environ.pop(k, None)
    os.environ.update(env); _hip.reload_switches()
    g = torch.Generator(device=dev); g.manual_seed(1)
    x0 = torch.rand((B, cin0, n, n, n), generator=g, device=dev) * 2 - 1
    w1 = torch.randn((32, 16, 3, 3, 3), generator=g, device=dev) * 0.05
    mask = torch.randint(0, 256, (B, n, n, n, 4), generator=g, device=dev, dtype=torch.uint8)
    gpre1 = torch.randn((B, n // 2, n // 2, n // 2, 32), generator=g, device=dev) * 1e-3
    for _ in range(2):
        gw, gb = ops_bwd.conv3d_dgrad_wgrad0(gpre1, w1, mask, 0.2, x0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); ops_bwd.conv3d_dgrad_wgrad0(gpre1, w1, mask, 0.2, x0); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    err = None
    if ref is not None:
        err = (float((gw - ref[0]).abs().max() / ref[0].abs().max()), float((gb - ref[1]).abs().max() / ref[1].abs().max()))
    return float(np.median(ts)), (gw, gb), err


for name, (n, c, B) in {"c3": (256, 3, 8), "native160": (160, 5, 30)}.items():
    ref = None
    for env in ({}, {"LIFTREG_FUSED_BWD_NZ": "4"}, {}, {"LIFTREG_FUSED_BWD_NZ": "4"}):
        ms, out, err = run(n, c, B, env, ref)
        if ref is None:
            ref = out
        print(f"{name:10s} {str(env):70s} {ms:.3f} ms   vs first: {err}", flush=True)
