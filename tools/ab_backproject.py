#!/usr/bin/env python3
"""A/B of lr_backproject_f32's block shapes in one process (same box, same clocks): batch chunk per block (LIFTREG_BP_CHUNK) and
planes side by side (LIFTREG_BP_JP) at the BASELINE shapes and the reference's shipped one.  Prints ms, TB/s of the algorithmic
bytes (4 P V + 4 P Pw Ph per sample) and the fraction of 8 TB/s."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import _hip, ops
from liftreg_amd.utils.sdct_projection_utils import scan_poses

dev = torch.device("cuda:0")


POLLUTE = "--pollute" in sys.argv     # a 2 GB device copy in front of every timed launch: the in-step situation (other kernels'
# buffers have gone through the caches / TLBs since this kernel last ran), not the back-to-back loop
# --pollute-mode copy2g (default) | copy256m | fill2g | read2g | altbuf (no other kernel: the output alternates between two buffers)
MODE = sys.argv[sys.argv.index("--pollute-mode") + 1] if "--pollute-mode" in sys.argv else "copy2g"
_big = None


def run(n, P, R, B, env, reps=20):
    global _big
    if POLLUTE and _big is None:
        _big = torch.empty(2 * 1024 ** 3 // 4, device=dev), torch.empty(2 * 1024 ** 3 // 4, device=dev)
    for k in ("LIFTREG_BP_CHUNK", "LIFTREG_BP_JP", "LIFTREG_BP_TOUCH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    _hip.reload_switches()
    proj = torch.rand(B, P, R, R, device=dev)
    poses = scan_poses(30, P, n).astype(np.float32)
    out = torch.empty(B, P, n, n, n, device=dev)
    for _ in range(3):
        ops.backproject(proj, poses, (n, n, n), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    out2 = torch.empty_like(out) if (POLLUTE and MODE == "altbuf") else None
    for it in range(reps):
        if POLLUTE:
            if MODE == "copy2g":
                _big[1].copy_(_big[0])
            elif MODE == "copy256m":
                _big[1][:64 * 1024 ** 2].copy_(_big[0][:64 * 1024 ** 2])
            elif MODE == "fill2g":
                _big[1].fill_(1.0)
            elif MODE == "read2g":
                _big[0].sum()
            elif MODE == "altbuf" and it % 2:
                out, out2 = out2, out
            elif MODE == "copy2g_then_touch_views":      # the polluter, then one pass over the views: are they what the kernel misses?
                _big[1].copy_(_big[0])
                proj.sum()
        e0.record()
        ops.backproject(proj, poses, (n, n, n), out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    return ms, 4 * B * P * (n ** 3 + R * R) / 1e9 / ms


SHAPES = {"native160": (160, 4, 240, 30), "c3": (256, 2, 256, 8), "c2": (128, 2, 128, 4), "c4": (256, 11, 256, 4), "c5": (384, 2, 512, 4)}
ENVS = ({"LIFTREG_BP_CHUNK": "0", "LIFTREG_BP_JP": "1"}, {"LIFTREG_BP_CHUNK": "0"}, {"LIFTREG_BP_JP": "1"}, {}, {"LIFTREG_BP_TOUCH": "0"},
        {"LIFTREG_BP_CHUNK": "4"}, {"LIFTREG_BP_CHUNK": "2"}, {"LIFTREG_BP_CHUNK": "1"})
for name, (n, P, R, B) in SHAPES.items():
    for env in ENVS:
        ms, tb = run(n, P, R, B, env)
        print(f"{name:10s} {str(env):58s} {ms:.4f} ms  {tb:.2f} TB/s  frac {tb / 8:.3f}", flush=True)
