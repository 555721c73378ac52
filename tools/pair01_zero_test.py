"""Diagnostic: the fused pair kernel on random data vs on all-zero data (same instruction stream: a large gap means the chip
holds its clock / issue rate down under the power of the dense bf16 MFMA stream, MI355X_MICROARCH.md 'DVFS give-back')."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops
dev = torch.device("cuda:0")
B, n = 8, 256
g = torch.Generator(device=dev).manual_seed(1)
def make(zero):
    x0 = torch.rand(B, 1, n, n, n, device=dev, generator=g)
    rest = torch.randn(B, 2, n, n, n, device=dev, generator=g)
    w0 = torch.randn(16, 3, 3, 3, 3, device=dev, generator=g) / 9
    w1 = torch.randn(32, 16, 3, 3, 3, device=dev, generator=g) / 20
    b0 = torch.randn(16, device=dev, generator=g) * 0.1
    b1 = torch.randn(32, device=dev, generator=g) * 0.1
    if zero:
        for t in (x0, rest, w0, w1, b0, b1):
            t.zero_()
    return x0, rest, w0, b0, w1, b1
sets = {"random": make(False), "zeros": make(True)}
packs = {k: ops.conv3d_pair01_pack(v[2], v[4]) for k, v in sets.items()}
for rep in range(3):
    for k, (x0, rest, w0, b0, w1, b1) in sets.items():
        for _ in range(3):
            ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, packed=packs[k])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, packed=packs[k])
        e1.record()
        torch.cuda.synchronize()
        print(f"rep {rep} {k:7s} {e0.elapsed_time(e1) / 8:7.3f} ms")
