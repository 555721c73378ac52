"""A/B on one box: the fused pair kernel (csrc/conv01_fused.hip) with block 0's K packed densely (17 MFMAs per 16-voxel tile,
default for three input channels) vs padded (24; LIFTREG_PAIR01_DENSE=0), C3 shapes (256^3, 3 channels, B = 8).  Interleaved,
HIP-event timed; also both results' distance from an fp64 convolution on a small case.
Usage: python tools/ab_pair01_dense.py [--n 256] [--batch 8] [--reps 5]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import _hip, ops  # noqa: E402


def _mode(dense):
    os.environ["LIFTREG_PAIR01_DENSE"] = "1" if dense else "0"
    _hip.lib().lr_reload_switches()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    w0 = torch.randn(16, 3, 3, 3, 3, device=dev, generator=g) * (2.0 / 81) ** 0.5
    b0 = torch.randn(16, device=dev, generator=g) * 0.1
    w1 = torch.randn(32, 16, 3, 3, 3, device=dev, generator=g) * (2.0 / 432) ** 0.5
    b1 = torch.randn(32, device=dev, generator=g) * 0.1
    pkp = ops.conv3d_pair01_pack(w0, w1)
    hps = ops.LAYOUT_NDHWC_HPS
    # accuracy on a small case against fp64
    xs = torch.randn(1, 3, 12, 40, 48, device=dev, generator=g)
    y = F.leaky_relu(F.conv3d(xs.double().cpu(), w0.double().cpu(), b0.double().cpu(), padding=1), 0.2)
    ref = F.leaky_relu(F.conv3d(y, w1.double().cpu(), b1.double().cpu(), stride=2, padding=1), 0.2)
    for dense in (False, True):
        _mode(dense)
        got = ops.conv3d_pair01(xs[:, 0:1].contiguous(), xs[:, 1:].contiguous(), w0, b0, w1, b1, out_layout=ops.LAYOUT_NDHWC, packed=pkp)
        got = got.permute(0, 4, 1, 2, 3).double().cpu()
        print(f"dense={int(dense)}: vs fp64: max {float((got - ref).abs().max()):.3e} rms {float((got - ref).pow(2).mean().sqrt()):.3e} "
              f"(scale {float(ref.abs().max()):.3f})")
    B, N = a.batch, a.n
    x0 = torch.rand(B, 1, N, N, N, device=dev, generator=g)
    rest = torch.randn(B, 2, N, N, N, device=dev, generator=g)
    outs = {}
    for rep in range(a.reps):
        for dense in (False, True):
            _mode(dense)
            for _ in range(2):
                yy = ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, out_layout=hps, packed=pkp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                yy = ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, out_layout=hps, packed=pkp)
            e1.record()
            torch.cuda.synchronize()
            outs[dense] = yy
            print(f"rep {rep}: dense={int(dense)}: {e0.elapsed_time(e1) / 5:.3f} ms")
    d = (outs[True] - outs[False]).abs()
    print(f"max |dense - padded| = {float(d.max()):.3e} (scale {float(outs[False].abs().max()):.3f})")


if __name__ == "__main__":
    main()
