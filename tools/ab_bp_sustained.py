#!/usr/bin/env python3
"""Is the back-to-back timing of lr_backproject_f32 (tools/ab_backproject.py: record, launch, record, SYNCHRONIZE per repetition) the
sustained rate?  The same kernel in one uninterrupted queue of 40 launches (events recorded around each, one synchronize at the end),
and with a host sleep in front of every launch."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops
from liftreg_amd.utils.sdct_projection_utils import scan_poses
dev = torch.device("cuda:0")
for name, (n, P, R, B) in {"native160": (160, 4, 240, 30), "c3": (256, 2, 256, 8)}.items():
    proj = torch.rand(B, P, R, R, device=dev); poses = scan_poses(30, P, n).astype(np.float32)
    out = torch.empty(B, P, n, n, n, device=dev)
    gb = 4 * B * P * (n ** 3 + R * R) / 1e9
    for _ in range(3):
        ops.backproject(proj, poses, (n, n, n), out=out)
    torch.cuda.synchronize()
    for mode in ("sync each", "one queue", "sleep 20 ms each"):
        evs = []
        for _ in range(40):
            if mode == "sleep 20 ms each":
                torch.cuda.synchronize(); time.sleep(0.02)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.backproject(proj, poses, (n, n, n), out=out); e1.record()
            if mode == "sync each":
                torch.cuda.synchronize()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in evs]
        print(f"{name:10s} {mode:18s} median {np.median(ts):.4f} ms ({gb / np.median(ts) / 8:.3f} of 8 TB/s)  first 5: {[round(t, 3) for t in ts[:5]]}  last 5: {[round(t, 3) for t in ts[-5:]]}", flush=True)
