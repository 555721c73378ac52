#!/usr/bin/env python3
"""Who is closer to fp64 at the reference's shipped shape (160^3, 5 input channels, batch 2): the fused first-blocks backward
(lr_conv3d_dgrad_wgrad0_split_f32), the two generic kernels it replaces, or ATen's fp32 CPU autograd?  Experiment behind the
tolerance of tests/test_gpu_training.py::test_native160_training_step_takes_the_fused_path."""
import os, sys, time
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from liftreg_amd import ops, ops_bwd

dev = torch.device("cuda:0")
n, P, B = int(os.environ.get("N", "160")), 4, 2
d = bench.synth_inputs(dict(n=n, P=P, R=n * 3 // 2, B=B, L=8), dev, seed=11)
from liftreg_amd.utils.sdct_projection_utils import scan_poses
tv = ops.backproject(d["target_proj"], scan_poses(30, P, n).astype(np.float32), (n, n, n))
x0 = torch.cat([d["source"], tv], 1).contiguous()
torch.manual_seed(3)
w0 = (torch.randn(16, P + 1, 3, 3, 3) * 0.1).to(dev); b0 = (torch.randn(16) * 0.1).to(dev)
w1 = (torch.randn(32, 16, 3, 3, 3) * 0.05).to(dev)
lay = ops.LAYOUT_NDHWC_HPS
y1, y0, mask = ops.conv3d_pair01_train(x0, w0, b0, w1, None, mid_layout=lay, out_layout=ops.LAYOUT_NDHWC)
g = torch.Generator(device=dev); g.manual_seed(5)
# a smooth + noisy upstream gradient, as the loss produces (coherent sums)
gpre1 = (torch.randn(y1.shape, generator=g, device=dev) * 0.1 + 1.0) * 1e-3
gw0, gb0 = ops_bwd.conv3d_dgrad_wgrad0(gpre1, w1, mask, 0.2, x0)
gpre0, _, _ = ops_bwd.conv3d_bwd(y0, lay, w1, y1, ops.LAYOUT_NDHWC, gpre1, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True, mask_input_slope=0.2, x_sign4=mask)
_, gw_gen, gb_gen = ops_bwd.conv3d_bwd(x0, ops.LAYOUT_NCDHW, w0, y0, lay, gpre0, ops.LAYOUT_NDHWC, 1, need_gx=False, gy_is_gpre=True)
res = {}
for name, dt in (("aten_fp32", torch.float32), ("aten_fp64", torch.float64)):
    t0 = time.time()
    xd = x0.cpu().to(dt)
    w0d, b0d = w0.cpu().to(dt).requires_grad_(True), b0.cpu().to(dt).requires_grad_(True)
    y0d = F.leaky_relu(F.conv3d(xd, w0d, b0d, padding=1), 0.2)
    y1d = F.conv3d(y0d, w1.cpu().to(dt), None, stride=2, padding=1)
    (y1d * gpre1.permute(0, 4, 1, 2, 3).cpu().to(dt)).sum().backward()
    res[name] = (w0d.grad.double(), b0d.grad.double())
    print(name, f"{time.time() - t0:.1f} s", flush=True)
ref_w, ref_b = res["aten_fp64"]
sw, sb = float(ref_w.abs().max()), float(ref_b.abs().max())
for name, (gw, gb) in (("fused", (gw0, gb0)), ("generic", (gw_gen, gb_gen)), ("aten_fp32", res["aten_fp32"])):
    print(f"{name:10s} gw max|err|/scale {float((gw.cpu().double() - ref_w).abs().max()) / sw:.2e}   gb {float((gb.cpu().double() - ref_b).abs().max()) / sb:.2e}")
