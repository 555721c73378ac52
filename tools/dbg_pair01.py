"""Debug probes for csrc/conv01_fused.hip: structured weights locate an addressing error."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops

dev = torch.device("cuda:0")


def ref64(x, w0, b0, w1, b1):
    y = F.leaky_relu(F.conv3d(x.double(), w0.double(), b0.double(), stride=1, padding=1), 0.2)
    return F.leaky_relu(F.conv3d(y, w1.double(), b1.double(), stride=2, padding=1), 0.2)


def run(x, w0, b0, w1, b1):
    xd = x.to(dev)
    y = ops.conv3d_pair01(xd[:, 0:1].contiguous(), xd[:, 1:].contiguous(), w0.to(dev), b0.to(dev), w1.to(dev), b1.to(dev),
                          out_layout=ops.LAYOUT_NDHWC)
    torch.cuda.synchronize()
    return y.permute(0, 4, 1, 2, 3).cpu().double()


def report(name, got, ref):
    err = (got - ref).abs()
    bad = err > 1e-4 * max(1.0, float(ref.abs().max()))
    print(f"== {name}: max err {float(err.max()):.3e}, bad {int(bad.sum())}/{bad.numel()}")
    if bad.any():
        B, C, D, W, H = bad.shape
        print("   bad by channel:", bad.sum(dim=(0, 2, 3, 4)).tolist())
        print("   bad by oz     :", bad.sum(dim=(0, 1, 3, 4)).tolist())
        print("   bad by oy     :", bad.sum(dim=(0, 1, 2, 4)).tolist())
        print("   bad by ox     :", bad.sum(dim=(0, 1, 2, 3)).tolist())
        idx = bad.nonzero()[:6]
        for i in idx:
            t = tuple(int(v) for v in i)
            print("   e.g.", t, "got", float(got[t]), "ref", float(ref[t]))


def main():
    g = torch.Generator().manual_seed(3)
    B, Cin, D, W, H = 1, 3, 8, 32, 32
    x = torch.randn(B, Cin, D, W, H, generator=g)
    z16, z32 = torch.zeros(16), torch.zeros(32)
    # probe 1: block 0 = copy channel c%3 of the centre voxel into channel c (positive inputs), block 1 = centre tap identity on 16 channels
    xp = x.abs() + 0.5
    w0 = torch.zeros(16, 3, 3, 3, 3)
    for c in range(16):
        w0[c, c % 3, 1, 1, 1] = 1.0 + c / 16.0
    w1 = torch.zeros(32, 16, 3, 3, 3)
    for c in range(32):
        w1[c, c % 16, 1, 1, 1] = 1.0
    report("copy/copy", run(xp, w0, z16, w1, z32), ref64(xp, w0, z16, w1, z32))
    # probe 2: block 1 picks one tap at a time
    for tap in (0, 2, 6, 8, 13, 14, 18, 26):
        w1 = torch.zeros(32, 16, 3, 3, 3)
        for c in range(32):
            w1[c, c % 16].view(-1)[tap] = 1.0
        report(f"copy/tap{tap}", run(xp, w0, z16, w1, z32), ref64(xp, w0, z16, w1, z32))
    # probe 3: block 0 picks one tap at a time, block 1 centre
    w1 = torch.zeros(32, 16, 3, 3, 3)
    for c in range(32):
        w1[c, c % 16, 1, 1, 1] = 1.0
    for tap in (0, 1, 2, 5, 8, 9, 13, 17, 20, 23, 26):
        w0 = torch.zeros(16, 3, 3, 3, 3)
        for c in range(16):
            w0[c, c % 3].view(-1)[tap] = 1.0
        report(f"tap{tap}/copy", run(xp, w0, z16, w1, z32), ref64(xp, w0, z16, w1, z32))
    # probe 4: random everything
    w0 = torch.randn(16, 3, 3, 3, 3, generator=g) * 0.15
    w1 = torch.randn(32, 16, 3, 3, 3, generator=g) * 0.07
    b0, b1 = torch.randn(16, generator=g) * 0.1, torch.randn(32, generator=g) * 0.1
    report("random", run(x, w0, b0, w1, b1), ref64(x, w0, b0, w1, b1))


main()
