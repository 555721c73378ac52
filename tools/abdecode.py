#!/usr/bin/env python3
"""Interleaved A/B (one process) of the decode half at the C3 shape: one-pass decode + NCC-moments kernel versus the
one-pass decode with the moments in its epilogue."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from liftreg_amd import ops  # noqa: E402
from liftreg_amd.utils.net_utils import identity_axis_tables  # noqa: E402

dev = torch.device("cuda:0")
n, B, L = 256, 8, 56
g = torch.Generator(device=dev)
g.manual_seed(1)
V = n ** 3
basis = torch.empty((L, 3 * V), device=dev).normal_(0, 0.02 / np.sqrt(L), generator=g)
mean = torch.zeros(3 * V, device=dev)
img = torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1
tgt = torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1
coefs = torch.randn((B, L), generator=g, device=dev)
ids = [torch.from_numpy(t).to(dev) for t in identity_axis_tables((n, n, n))]


def two():
    d, p, w = ops.pca_warp(coefs, basis, mean, ids, img)
    return ops.ncc_moments(w, tgt, B)


def fused():
    return ops.pca_warp(coefs, basis, mean, ids, img, target=tgt)[3]


def timeit(f, it=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


two(); fused(); torch.cuda.synchronize()
ta, tb = [], []
for r in range(9):
    ta.append(timeit(two))
    tb.append(timeit(fused))
print(f"decode + ncc_moments kernel : median {np.median(ta):.4f} ms  min {np.min(ta):.4f}")
print(f"decode with NCC epilogue    : median {np.median(tb):.4f} ms  min {np.min(tb):.4f}")
