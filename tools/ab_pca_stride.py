#!/usr/bin/env python3
"""Does the one-pass decode care about the power-of-two distance between its streams?  At the C3 shape the basis rows are
192 MiB apart and the three component thirds 64 MiB apart: 168 read streams whose addresses differ only above bit 26.
Times lr_pca_warp_slab_f32 with the row stride (ldb) and the component stride padded by a few KiB (one process,
interleaved)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from liftreg_amd import _hip, ops  # noqa: E402
from liftreg_amd.utils.net_utils import identity_axis_tables  # noqa: E402

dev = torch.device("cuda:0")
n, B, L = 256, 8, 56
V = n ** 3
g = torch.Generator(device=dev)
g.manual_seed(1)
img = torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1
coefs = torch.randn((B, L), generator=g, device=dev)
ids = [torch.from_numpy(t).to(dev) for t in identity_axis_tables((n, n, n))]
disp = torch.empty((B, 3, n, n, n), device=dev)
phi = torch.empty_like(disp)
warped = torch.empty((B, 1, n, n, n), device=dev)
lib = _hip.lib()
st = torch.cuda.current_stream().cuda_stream


def make(pad_c, pad_l):
    bcs = V + pad_c
    ldb = 3 * bcs + pad_l
    store = torch.full((L * ldb,), 0.0004, device=dev)      # a smooth (constant) field: the gather stays local, as in the model
    mean = torch.zeros(3 * bcs, device=dev)

    def run():
        _hip.check(lib.lr_pca_warp_slab_f32(coefs.data_ptr(), store.data_ptr(), 0, mean.data_ptr(), img.data_ptr(),
                                            ids[0].data_ptr(), ids[1].data_ptr(), ids[2].data_ptr(), disp.data_ptr(),
                                            phi.data_ptr(), warped.data_ptr(), B, L, 1, n, n, n, 0, n, ldb, bcs,
                                            _hip.WARP_USING_SCALE, None, None, None, st), "pca_warp")
    return run, store


def timeit(f, it=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


cases = [(0, 0), (0, 1024), (1024, 0), (1024, 4096), (4096 + 64, 64 * 37)]
runs = [make(*c) for c in cases]
for _ in range(150):                       # ~2 s: let the clocks ramp before anything is timed
    for r, _ in runs:
        r()
torch.cuda.synchronize()
t = [[] for _ in cases]
for rep in range(7):
    for i, (r, _) in enumerate(runs):
        t[i].append(timeit(r))
for c, ti in zip(cases, t):
    print(f"component pad {c[0]:5d} floats, row pad {c[1]:5d} floats: median {np.median(ti):.4f} ms  min {np.min(ti):.4f}")
