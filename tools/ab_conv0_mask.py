#!/usr/bin/env python3
"""First block at the C3 shape with and without the LeakyReLU sign mask output (the training forward writes it), interleaved
in one process; also the two-buffer input form the inference path uses."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from liftreg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, n = 8, 256
g = torch.Generator(device=dev)
g.manual_seed(0)
x = torch.empty((B, 3, n, n, n), device=dev).uniform_(-1, 1, generator=g)
w = torch.empty((16, 3, 3, 3, 3), device=dev).normal_(0, 0.1, generator=g)
b = torch.zeros(16, device=dev)
packed = ops.conv3d_pack_weights(w, ops.LAYOUT_NCDHW)
out = torch.empty((B, n, n, n, 16), device=dev)
mask = torch.empty((B, n, n, n, 4), dtype=torch.uint8, device=dev)
mv, tv = x[:, :1].contiguous(), x[:, 1:].contiguous()
cases = {
    "plain": lambda: ops.conv3d_k3_lrelu(x, w, b, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=ops.LAYOUT_NDHWC_HPS, packed=packed, out=out),
    "with mask": lambda: ops.conv3d_k3_lrelu(x, w, b, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=ops.LAYOUT_NDHWC_HPS, packed=packed, out=out, mask_out=mask),
    "two input buffers": lambda: ops.conv3d_first_split(mv, tv, w, b, out_layout=ops.LAYOUT_NDHWC_HPS, packed=packed, out=out),
}
for _ in range(60):
    for f in cases.values():
        f()
torch.cuda.synchronize()
t = {k: [] for k in cases}
for r in range(7):
    for k, f in cases.items():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            f()
        e.record()
        torch.cuda.synchronize()
        t[k].append(s.elapsed_time(e) / 10)
for k, v in t.items():
    print(f"{k:20s} median {np.median(v):.4f} ms  min {np.min(v):.4f}")
