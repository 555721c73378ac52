"""A/B on one box: the batched projector (lr_drr_forward_batch_f32, HU input, flip folded) at C3 / native sizes — kernel ms per
volume (HIP events), also with the general kernel (LIFTREG_DRR_GENERAL=1), and (round 6) the two-pass form: one HU -> mu pass
over the volumes (lr_hu_to_mu_f32, one conversion per VOXEL) + the mu-input projector (no conversion per TAP).
Usage: python tools/ab_drr.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import _hip, ops  # noqa: E402
from liftreg_amd.utils.sdct_projection_utils import scan_poses  # noqa: E402

dev = torch.device("cuda:0")
for n, P, R, B in ((256, 2, 256, 8), (160, 4, 240, 30)):
    g = torch.Generator(device=dev).manual_seed(n)
    vols = torch.rand(B, n, n, n, device=dev, generator=g) * 2000 - 1000
    p32 = scan_poses(30, P, n).astype(np.float32)
    for gen in (0, 1):
        if gen:
            os.environ["LIFTREG_DRR_GENERAL"] = "1"
        else:
            os.environ.pop("LIFTREG_DRR_GENERAL", None)
        _hip.lib().lr_reload_switches()
        for rep in range(3):
            for _ in range(2):
                ops.drr_forward_batch(vols, p32, (R, R), hu_input=True, flip_w=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.drr_forward_batch(vols, p32, (R, R), hu_input=True, flip_w=True)
            e1.record()
            torch.cuda.synchronize()
            print(f"{n}^3 P={P} R={R} B={B} {'general' if gen else 'fast'}: {e0.elapsed_time(e1) / 5 / B:.4f} ms per volume")
    os.environ.pop("LIFTREG_DRR_GENERAL", None)
    _hip.lib().lr_reload_switches()

    def two_pass():
        return ops.drr_forward_batch(ops.hu_to_mu(vols), p32, (R, R), hu_input=False, flip_w=True)

    for rep in range(3):
        for _ in range(2):
            two_pass()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(5):
            two_pass()
        e1.record()
        mu = ops.hu_to_mu(vols)
        for _ in range(5):
            ops.drr_forward_batch(mu, p32, (R, R), hu_input=False, flip_w=True)
        e2.record()
        torch.cuda.synchronize()
        print(f"{n}^3 P={P} R={R} B={B} two-pass (hu_to_mu + mu projector): {e0.elapsed_time(e1) / 5 / B:.4f} ms per volume; "
              f"mu projector alone {e1.elapsed_time(e2) / 5 / B:.4f}")
