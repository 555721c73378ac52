#!/usr/bin/env python3
"""The many-channel first block alone at the C4 shape, a few launches (profiling target; env LIFTREG_C0CL_ABL etc. apply)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops
dev = torch.device("cuda:0")
B, C, n = 4, 12, 256
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.rand((B, C, n, n, n), generator=g, device=dev) * 2 - 1
w = torch.randn((16, C, 3, 3, 3), generator=g, device=dev) * 0.05
b = torch.randn((16,), generator=g, device=dev) * 0.1
pk = ops.conv3d_pack_weights_bf16_planar(w)
out = torch.empty((B, n, n, n, 16), dtype=torch.bfloat16, device=dev)
for _ in range(5):
    ops.conv3d_first_bf16(x, w, b, out_layout=ops.LAYOUT_BF16_NDHWC_HPS, packed=pk, out=out)
torch.cuda.synchronize()
