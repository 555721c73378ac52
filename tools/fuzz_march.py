"""Development aid (GPU box): 60 random shapes through the z-marching first-block kernels (conv0_split_f32.hip) — the bf16
contract against the channel-pass kernel (>= 99.9 % identical, one bf16 ulp) and the split-operand fp32 block against the
default fp32-MFMA kernel (2e-6 of the scale).  Ragged W / H, 1..4 channels, 1..18 planes, both output layouts."""
import os
os.environ.setdefault("LIFTREG_SWITCH_AUTOSYNC", "1")   # this tool flips library switches between calls, sys, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops
dev = torch.device("cuda:0")
random.seed(5)
bad = 0
for it in range(60):
    B, Cin = random.randint(1, 3), random.randint(1, 4)
    D, W, H = random.randint(1, 18), random.randint(128, 210), 4 * random.randint(32, 55)
    if W * H < 128 * 128: W = 160
    hps = (H % 2 == 0) and random.random() < 0.5
    g = torch.Generator().manual_seed(it)
    x = torch.randn(B, Cin, D, W, H, generator=g).to(dev)
    w = (torch.randn(16, Cin, 3, 3, 3, generator=g) * 0.2).to(dev)
    b = (torch.randn(16, generator=g) * 0.1).to(dev)
    # bf16 contract: march vs channel-pass kernel
    lay = ops.LAYOUT_BF16_NDHWC_HPS if hps else ops.LAYOUT_BF16_NDHWC
    if Cin <= 3:
        y = ops.conv3d_first_bf16(x, w, b, out_layout=lay)
        os.environ["LIFTREG_CONV0_BF16_PASSES"] = "1"
        y0 = ops.conv3d_first_bf16(x, w, b, out_layout=lay)
        del os.environ["LIFTREG_CONV0_BF16_PASSES"]
        same = (y == y0).float().mean().item()
        d = (y.float() - y0.float()).abs().max().item() / max(y0.float().abs().max().item(), 1e-9)
        if same < 0.999 or d > 2.0 ** -7: bad += 1; print("BF16 MISMATCH", B, Cin, D, W, H, hps, same, d)
    # fp32 split vs native
    layf = ops.LAYOUT_NDHWC_HPS if hps else ops.LAYOUT_NDHWC
    os.environ["LIFTREG_CONV0_SPLIT"] = "1"
    z = ops.conv3d_k3_lrelu(x, w, b, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=layf)
    del os.environ["LIFTREG_CONV0_SPLIT"]
    z0 = ops.conv3d_k3_lrelu(x, w, b, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=layf)
    e = (z - z0).abs().max().item() / max(z0.abs().max().item(), 1e-9)
    if not (e < 2e-6): bad += 1; print("F32 MISMATCH", B, Cin, D, W, H, hps, e)
print("fuzz done, mismatches:", bad)
