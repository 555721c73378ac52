#!/usr/bin/env python3
"""PCA-gradient launch sweep (development aid): ms per call of ops_bwd.pca_bwd_coef at C3 for several grid sizes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops_bwd  # noqa: E402

dev = torch.device("cuda:0")
L, M, B = 56, 3 * 256 ** 3, 8
basis = torch.empty((L, M), device=dev).normal_(0, 0.01)
gd = torch.rand(B, M, device=dev)
for dt in (torch.float32, torch.bfloat16):
    bs = basis.to(dt)
    for nblk in (256, 512, 1024, 2048, 4096):
        f = lambda: ops_bwd.pca_bwd_coef(gd, bs, nblk=nblk)
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            f()
        e.record()
        torch.cuda.synchronize()
        print(dt, nblk, round(s.elapsed_time(e) / 10, 4), "ms", flush=True)
    del bs
