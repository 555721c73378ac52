#!/usr/bin/env python3
"""The batched projector (C3: 8 volumes of 256^3, 2 views, HU input folded) back to back, after another kernel's 2 GB stream (the
in-sequence situation of simulate + register), and after that stream + one pass over the volumes — is its in-sequence time (0.118 ms
per volume in the bench line against 0.104 back to back) the volumes' cache residency, as it was for the backprojection's views?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops
from liftreg_amd.utils.sdct_projection_utils import scan_poses
dev = torch.device("cuda:0")
n, P, R, B = 256, 2, 256, 8
g = torch.Generator(device=dev).manual_seed(n)
vols = torch.rand(B, n, n, n, device=dev, generator=g) * 2000 - 1000
p32 = scan_poses(30, P, n).astype(np.float32)
big = torch.empty(2 * 1024 ** 3 // 4, device=dev), torch.empty(2 * 1024 ** 3 // 4, device=dev)
def project():
    return ops.drr_forward_batch(vols, p32, (R, R), hu_input=True, flip_w=True)
for _ in range(20):
    project()
torch.cuda.synchronize()
for name, pre in (("back to back", None), ("after a 2 GB copy", lambda: big[1].copy_(big[0])),
                  ("after a 2 GB copy + volumes.sum()", lambda: (big[1].copy_(big[0]), vols.sum())), ("back to back", None)):
    ts = []
    for _ in range(15):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); project(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"{name:36s} {np.median(ts) / B:.4f} ms per volume", flush=True)
