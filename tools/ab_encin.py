#!/usr/bin/env python3
"""C4 first block: fp32 feature volume path (backproject + cat + conv0) vs the bf16 channels-last encoder input path; checks bits."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops
from liftreg_amd.utils.sdct_projection_utils import scan_poses
dev = torch.device("cuda:0")
B, P, n = 4, 11, 256
g = torch.Generator(device=dev); g.manual_seed(1)
mv = torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1
proj = torch.rand((B, P, n, n), generator=g, device=dev) * 2 - 1
w = torch.randn((16, P + 1, 3, 3, 3), generator=g, device=dev) * 0.05
b = torch.randn((16,), generator=g, device=dev) * 0.1
poses = scan_poses(30, P, n).astype(np.float32)
pk = ops.conv3d_pack_weights_bf16_planar(w)
lay = ops.LAYOUT_BF16_NDHWC_HPS
def old():
    x = torch.empty((B, P + 1, n, n, n), dtype=torch.float32, device=dev)
    x[:, 0:1].copy_(mv)
    ops.backproject(proj, poses, (n, n, n), out=x[:, 1:], out_batch_stride=(P + 1) * n ** 3)
    return ops.conv3d_first_bf16(x, w, b, out_layout=lay, packed=pk)
def new():
    e = ops.backproject_encoder_input_bf16(mv, proj, poses)
    return ops.conv3d_first_clin_bf16(e, w, b, out_layout=lay, packed=pk)
def t(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 5
yo, yn = old(), new()
print("identical:", torch.equal(yo, yn), "mismatch frac", float((yo != yn).float().mean()))
# the encoder input itself: channels vs the fp32 volume rounded
e = ops.backproject_encoder_input_bf16(mv, proj, poses)
tv = ops.backproject(proj, poses, (n, n, n))
print("encin ch0 == bf16(moving):", torch.equal(e[..., 0], mv[:, 0].to(torch.bfloat16)),
      " views == bf16(backproject):", torch.equal(e[..., 1:P + 1], tv.permute(0, 2, 3, 4, 1).to(torch.bfloat16)), " pad zero:", bool((e[..., P + 1:] == 0).all()))
with ops.kernel_timer() as kt:
    old(); new(); torch.cuda.synchronize()
print({k: round(float(np.mean(v["ms"])), 3) for k, v in kt.summary().items()})
print(f"old path {t(old):.3f} ms (incl. alloc + copy of moving), new path {t(new):.3f} ms")
