#!/usr/bin/env python3
"""Per-kernel micro-benchmark at BASELINE config shapes (development aid; bench.py is the contract).

  python tools/kbench.py [--config c3] [--only backproject,conv0,...] [--iters 10]
Prints one line per kernel: avg ms, achieved GB/s or TFLOP/s, fraction of the HBM / fp32-MFMA peak.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from liftreg_amd import ops  # noqa: E402
from liftreg_amd.utils.net_utils import identity_axis_tables  # noqa: E402
from liftreg_amd.utils.sdct_projection_utils import scan_poses  # noqa: E402

CONFIGS = {"c1": dict(n=64, P=2, R=64, B=1, L=56), "c2": dict(n=128, P=2, R=128, B=4, L=56),
           "c3": dict(n=256, P=2, R=256, B=8, L=56), "c4": dict(n=256, P=11, R=256, B=4, L=56)}


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--bf16", action="store_true", help="conv1..5 in the bf16-storage variant (model layouts)")
    ap.add_argument("--lib", default="", help="load this build of libliftreg_hip.so instead (kernel experiments)")
    a = ap.parse_args()
    if a.lib:
        from liftreg_amd import _hip
        _hip.LIB_PATH = os.path.abspath(a.lib)
    c = CONFIGS[a.config]
    n, P, R, B, L = c["n"], c["P"], c["R"], c["B"], c["L"]
    only = set(a.only.split(",")) if a.only else None
    dev = torch.device("cuda:0")
    V = n ** 3
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    rnd = lambda *s: torch.rand(*s, generator=g, device=dev) * 2 - 1
    poses = scan_poses(30, P, n).astype(np.float32)

    def report(name, ms, nbytes=None, flops=None):
        if flops is not None:
            tf = flops / ms / 1e9
            print(f"{name:28s} {ms:9.4f} ms  {tf:8.2f} TFLOP/s  {tf / 157.3:6.1%} of fp32 MFMA", flush=True)
        else:
            gb = nbytes / ms / 1e6
            print(f"{name:28s} {ms:9.4f} ms  {gb:8.1f} GB/s     {gb / 8000:6.1%} of HBM", flush=True)

    def want(k):
        return only is None or k in only

    if want("backproject"):
        proj = rnd(B, P, R, R)
        buf = torch.empty((B, P + 1, n, n, n), device=dev)
        f = lambda: ops.backproject(proj, poses, (n, n, n), out=buf[:, 1:], out_batch_stride=(P + 1) * V)
        report("backproject", timeit(f, a.iters), nbytes=4 * (B * P * V + B * P * R * R))
        del buf
    if want("drr"):
        vol = rnd(n, n, n).abs()
        for nseg in (0, 1, 4, 8):
            f = lambda: ops.drr_forward(vol, poses, (R, R), (2.2, 2.2, 2.2), nseg=nseg)
            report(f"drr_forward nseg={nseg}", timeit(f, a.iters), nbytes=4 * (V + P * R * R))
    chans = [(P + 1, 16, 1), (16, 32, 2), (32, 32, 2), (32, 32, 2), (32, 32, 2), (32, 32, 2)]
    size = n
    for i, (ci, co, s) in enumerate(chans):
        if want(f"conv{i}") and size >= 8:
            if i == 0:
                x = rnd(B, ci, size, size, size)
                lay = ops.LAYOUT_NCDHW
            else:   # the model's layouts: rows parity-split along H whenever H is even
                x = rnd(B, size, size, size, ci)
                lay = ops.LAYOUT_NDHWC_HPS if size % 2 == 0 else ops.LAYOUT_NDHWC
            w = rnd(co, ci, 3, 3, 3) / (27 * ci) ** 0.5
            bb = rnd(co) * 0.1
            if a.bf16 and i > 0:
                so = (size - 1) // s + 1
                xb = x.to(torch.bfloat16)
                li = ops.LAYOUT_BF16_NDHWC_HPS if size % 2 == 0 else ops.LAYOUT_BF16_NDHWC
                lo = ops.LAYOUT_NCDHW if i == 5 else (ops.LAYOUT_BF16_NDHWC_HPS if so % 2 == 0 else ops.LAYOUT_BF16_NDHWC)
                pkb = ops.conv3d_pack_weights_bf16(w)
                f = lambda: ops.conv3d_k3_lrelu_bf16(xb, w, bb, s, in_layout=li, out_layout=lo, packed=pkb)
                report(f"conv{i}_bf16 {ci}->{co} s{s} @{size}", timeit(f, a.iters), nbytes=2 * B * (ci * size ** 3 + co * so ** 3))
                del x, xb
                size = so
                continue
            pk = ops.conv3d_pack_weights(w, lay)
            so = (size - 1) // s + 1
            lay_out = ops.LAYOUT_NCDHW if i == 5 else (ops.LAYOUT_NDHWC_HPS if so % 2 == 0 else ops.LAYOUT_NDHWC)
            f = lambda: ops.conv3d_k3_lrelu(x, w, bb, s, in_layout=lay, out_layout=lay_out, packed=pk)
            report(f"conv{i} {ci}->{co} s{s} @{size}", timeit(f, a.iters), flops=2.0 * 27 * ci * co * B * so ** 3)
            del x
        size = (size - 1) // s + 1
    if want("pca") or want("pca_bwd"):
        basis = torch.empty((L, 3 * V), device=dev).normal_(0, 0.01, generator=g)
        mean = torch.zeros(3 * V, device=dev)
        coefs = rnd(B, L)
        f = lambda: ops.pca_reconstruct(coefs, basis, mean)
        report("pca_reconstruct", timeit(f, a.iters), nbytes=4 * (L * 3 * V + 3 * V + B * 3 * V))
        if want("pca_bwd"):
            from liftreg_amd import ops_bwd
            gd = rnd(B, 3 * V)
            f = lambda: ops_bwd.pca_bwd_coef(gd, basis)
            report("pca_bwd_coef", timeit(f, a.iters), nbytes=4 * (L * 3 * V + B * 3 * V))
            del gd
        del basis
    if want("warp"):
        img = rnd(B, 1, n, n, n)
        disp = rnd(B, 3, n, n, n) * 0.02
        ids = [torch.from_numpy(t).to(dev) for t in identity_axis_tables((n, n, n))]
        f = lambda: ops.warp(img, disp, ids, None)
        report("warp_trilinear", timeit(f, a.iters), nbytes=4 * B * V * 8)
    if want("ncc"):
        x, y = rnd(B, 1, n, n, n), rnd(B, 1, n, n, n)
        f = lambda: ops.ncc_loss(x, y)
        report("ncc", timeit(f, a.iters), nbytes=8 * B * V)
    if want("fc"):
        K = 32 * (n // 32) ** 3
        x, w, bb = rnd(B, K), rnd(800, K), rnd(800)
        f = lambda: ops.linear_lrelu(x, w, bb, 0.2)
        report(f"linear {K}x800", timeit(f, a.iters), nbytes=4 * (800 * K + B * K))


if __name__ == "__main__":
    main()
