#!/usr/bin/env python3
"""lr_pca_warp_slab_f32 at C3 (256^3, B = 8, L = 56) for slabs of Dn planes: ms, TB/s of the slab's algorithmic bytes, and ms per plane —
does a slab launch of 1/8 cost 1/8?  (Sharding: eight launches of 32 planes took 3.57 ms against 2.76 for the whole volume.)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops
from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
dev = torch.device("cuda:0")
n, B, L = 256, 8, 56
net = model([n, n, n], {"drr_feature_num": 2, "latent_dim": L, "pca_path": "synthetic:1"}).to(dev).eval()
g = torch.Generator(device=dev); g.manual_seed(1)
moving = torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1
coefs = torch.randn((B, L), generator=g, device=dev)
net._ensure_pca(dev)
for Dn in (16, 24, 32, 36, 40, 48, 64, 96, 128, 256):
    d0, d1 = 64 if Dn <= 128 else 0, (64 if Dn <= 128 else 0) + Dn
    basis_s, mean_s = net.pca_slab(d0, d1, dev) if Dn < n else (net.pca_vectors_LxM, net.pca_mean)
    ids = (net._id0[d0:d1].contiguous(), net._id1, net._id2)
    fn = lambda: ops.pca_warp(coefs, basis_s, mean_s, ids, moving, d0=d0, d1=d1)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    Vs = Dn * n * n
    gb = 4 * (L * 3 * Vs + 3 * Vs + B * 3 * Vs * 2 + B * Vs * 2) / 1e9
    print(f"Dn {Dn:4d}  blocks {64 * Dn:6d}  {ms:.4f} ms  {gb / ms:.2f} TB/s  {ms / Dn * 1e3:.2f} us per plane", flush=True)
