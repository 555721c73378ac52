#!/usr/bin/env python3
"""Timing-only ablations of the many-channel first block (conv0_cl_bf16.hip) at the C4 shape, interleaved in one process.
The LIFTREG_C0CL_ABL switches (skip loads / stores / sweep / LDS writes: wrong results) exist only in a diagnostic build:
    make -C liftreg_amd/csrc -B -j8 EXTRA=-DLR_C0CL_ABLATIONS
(the shape / chunk / channel-pass comparisons work with the product build)."""
import os, sys, json
import numpy as np, torch
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
def run(env):
    for k in ("LIFTREG_C0CL_ABL", "LIFTREG_CONV0_BF16_PASSES", "LIFTREG_CONV0_CL_BLOCKS", "LIFTREG_C0CL_SHAPE", "LIFTREG_C0CL_CHUNKS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(3):
        ops.conv3d_first_bf16(x, w, b, out_layout=ops.LAYOUT_BF16_NDHWC_HPS, packed=pk, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.conv3d_first_bf16(x, w, b, out_layout=ops.LAYOUT_BF16_NDHWC_HPS, packed=pk, out=out)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10
cases = [("all", {}), ("no loads", {"LIFTREG_C0CL_ABL": "1"}), ("no stores", {"LIFTREG_C0CL_ABL": "2"}), ("no sweep", {"LIFTREG_C0CL_ABL": "4"}),
         ("no lds writes", {"LIFTREG_C0CL_ABL": "8"}), ("no loads+stores", {"LIFTREG_C0CL_ABL": "3"}), ("sweep only", {"LIFTREG_C0CL_ABL": "11"}),
         ("loads only", {"LIFTREG_C0CL_ABL": "14"}), ("loads+lds writes", {"LIFTREG_C0CL_ABL": "6"}), ("512 blocks", {"LIFTREG_CONV0_CL_BLOCKS": "512"}),
         ("shape 2x8", {"LIFTREG_C0CL_SHAPE": "28"}), ("shape 2x8 loads only", {"LIFTREG_C0CL_SHAPE": "28", "LIFTREG_C0CL_ABL": "14"}),
         ("shape 4x4", {"LIFTREG_C0CL_SHAPE": "44"}), ("shape 4x4 loads only", {"LIFTREG_C0CL_SHAPE": "44", "LIFTREG_C0CL_ABL": "14"}),
         ("chunks 1", {"LIFTREG_C0CL_CHUNKS": "1"}), ("chunks 4", {"LIFTREG_C0CL_CHUNKS": "4"}), ("chunks 8", {"LIFTREG_C0CL_CHUNKS": "8"}),
         ("2x8 chunks 4", {"LIFTREG_C0CL_SHAPE": "28", "LIFTREG_C0CL_CHUNKS": "4"}),
         ("old passes", {"LIFTREG_CONV0_BF16_PASSES": "1"})]
res = {}
for rnd in range(2):
    for name, env in cases:
        res.setdefault(name, []).append(run(env))
for name, v in res.items():
    print(f"{name:20s} {min(v):.3f} ms")
