#!/usr/bin/env python3
"""DRR leg with HU input: HU->mu as a one-pass prologue (lr_hu_to_mu_f32) vs folded into the fast projector's taps; bits + time."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops
from liftreg_amd.utils.sdct_projection_utils import scan_poses
dev = torch.device("cuda:0")
n, P, B = 256, 2, 8
g = torch.Generator(device=dev); g.manual_seed(3)
vols = [(torch.rand((n, n, n), generator=g, device=dev) * 2200 - 1100) for _ in range(B)]
p32 = scan_poses(30, P, n).astype(np.float32)
def run(fold):
    return [ops.drr_forward(v, p32, (n, n), (2.2, 2.2, 2.2), hu_input=True, flip_w=True, fold_hu=fold) for v in vols]
a, b = run(False), run(True)
print("bit-identical:", all(torch.equal(x, y) for x, y in zip(a, b)))
# the conversion itself on hostile values
x = torch.cat([torch.rand(1 << 22, generator=g, device=dev) * 7000 - 1500, torch.tensor([-1000.0, -999.99994, -1000.0001, 0.0, -0.0, 1e-30, 3e38, -3e38], device=dev)])
vol = x[: (x.numel() // 64) * 64].reshape(-1, 8, 8).contiguous()
d1 = ops.drr_forward(vol, p32[:1], (8, 8), (1, 1, 1), hu_input=True, fold_hu=False)
d2 = ops.drr_forward(vol, p32[:1], (8, 8), (1, 1, 1), hu_input=True, fold_hu=True)
print("hostile volume identical:", torch.equal(d1, d2))
for fold in (False, True, False, True):
    run(fold); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run(fold)
    torch.cuda.synchronize(); print("fold" if fold else "prologue", f"{(time.perf_counter() - t0) / 10 / B * 1e3:.4f} ms per volume")
