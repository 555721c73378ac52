import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
B, D, W, H = 2, 32, 32, 32
x0 = torch.rand(B, 1, D, W, H, device=dev, generator=g)
rest = torch.randn(B, 2, D, W, H, device=dev, generator=g)
w0 = torch.randn(16, 3, 3, 3, 3, device=dev, generator=g) / 9
b0 = torch.randn(16, device=dev, generator=g) * 0.1
w1 = torch.randn(32, 16, 3, 3, 3, device=dev, generator=g) / 20
b1 = torch.randn(32, device=dev, generator=g) * 0.1
full = ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, out_layout=ops.LAYOUT_NDHWC)
for world in (2, 4):
    for r in range(world):
        d0, d1 = r * D // world, (r + 1) * D // world
        lo2, hi2 = max(d0 - 2, 0), min(d1 + 1, D)
        mvs = x0[:, :, lo2:hi2]
        tv = rest[:, :, lo2:hi2].contiguous()
        rows = (d1 - d0) // 2
        buf = torch.zeros(B, rows + 3, W // 2, H // 2, 32, device=dev)
        y = ops.conv3d_pair01(mvs, tv, w0, b0, w1, b1, out_layout=ops.LAYOUT_NDHWC, out=buf[:, 2:2 + rows], slab=(D, lo2, d0 // 2, rows))
        torch.cuda.synchronize()
        ref = full[:, d0 // 2:d1 // 2]
        diff = (y - ref).abs()
        print(world, r, "equal", torch.equal(y, ref), "max diff", float(diff.max()), "bad planes", (diff.amax(dim=(0, 2, 3, 4)) > 0).nonzero().flatten().tolist())
