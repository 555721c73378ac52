"""A/B on one box: encoder blocks 0 + 1 as two fp32-MFMA kernels (default) vs the fused split-operand pair kernel
(csrc/conv01_fused.hip), C3 shapes (256^3, 3 channels, B = 8).  Interleaved, HIP-event timed; also the distance between
the two results.  Usage: python tools/ab_pair01.py [--n 256] [--batch 8] [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    B, N = a.batch, a.n
    x0 = torch.rand(B, 1, N, N, N, device=dev, generator=g)
    rest = torch.randn(B, 2, N, N, N, device=dev, generator=g)
    w0 = torch.randn(16, 3, 3, 3, 3, device=dev, generator=g) * (2.0 / 81) ** 0.5
    b0 = torch.randn(16, device=dev, generator=g) * 0.1
    w1 = torch.randn(32, 16, 3, 3, 3, device=dev, generator=g) * (2.0 / 432) ** 0.5
    b1 = torch.randn(32, device=dev, generator=g) * 0.1
    pk0 = ops.conv3d_pack_weights(w0, ops.LAYOUT_NCDHW)
    pk1 = ops.conv3d_pack_weights(w1, ops.LAYOUT_NDHWC_HPS)
    pkp = ops.conv3d_pair01_pack(w0, w1)
    hps = ops.LAYOUT_NDHWC_HPS

    def two():
        y = ops.conv3d_first_split(x0, rest, w0, b0, out_layout=hps, packed=pk0)
        return ops.conv3d_k3_lrelu(y, w1, b1, 2, in_layout=hps, out_layout=hps, packed=pk1)

    def pair():
        return ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, out_layout=hps, packed=pkp)

    ya, yb = two(), pair()
    torch.cuda.synchronize()
    diff = float((ya - yb).abs().max())
    print(f"max |two - pair| = {diff:.3e}, scale {float(ya.abs().max()):.3f}, rms diff {float((ya - yb).pow(2).mean().sqrt()):.3e}")
    del ya, yb
    for rep in range(a.reps):
        for name, fn in (("two kernels", two), ("pair kernel", pair)):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"rep {rep}: {name}: {e0.elapsed_time(e1) / 5:.3f} ms")


if __name__ == "__main__":
    main()
