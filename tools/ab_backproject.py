"""The backprojection at C3 shapes (8 x 2 views of 256^2 -> 256^3) beside torch.fill_ of the same 1.07 GB in the same process;
an argument names a variant build (make -C liftreg_amd/csrc variantf FILE=backproject NAME=<x> VFLAGS=...)."""
import os, sys
os.environ.setdefault("LIFTREG_SWITCH_AUTOSYNC", "1")   # this tool flips library switches between calls
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import _hip
if len(sys.argv) > 1:
    _hip.LIB_PATH = os.path.join(_hip.CSRC, f"libliftreg_hip_{sys.argv[1]}.so")
from liftreg_amd import ops
from liftreg_amd.utils.sdct_projection_utils import scan_poses
dev = torch.device("cuda:0")
B, P, n = 8, 2, 256
g = torch.Generator(device=dev).manual_seed(1)
proj = torch.rand((B, P, n, n), generator=g, device=dev) * 2 - 1
poses = scan_poses(30, P, n).astype(np.float32)
out = torch.empty((B, P, n, n, n), device=dev)
gb = out.numel() * 4 / 1e9


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rep in range(3):
    t4 = timeit(lambda: ops.backproject(proj, poses, (n, n, n), out=out, out_batch_stride=P * n ** 3))
    tf = timeit(lambda: out.fill_(1.0))
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'default':8s} rep {rep}: backproject {t4:.4f} ms = {gb / t4:.2f} TB/s | fill_ {tf:.4f} ms = {gb / tf:.2f} TB/s")
