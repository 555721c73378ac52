#!/usr/bin/env python3
"""Interleaved A/B timing of the weight-gradient kernel of an encoder block in ONE process (like tools/abconv.py):
  python tools/abwgrad.py --block 1 --libs liftreg_amd/csrc/libliftreg_hip.so,liftreg_amd/csrc/libx_A.so [--env-b NAME=VALUE]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from liftreg_amd import _hip  # noqa: E402


def load(path):
    h = C.CDLL(os.path.abspath(path))
    for name in ("lr_conv3d_wgrad_partial_floats", "lr_conv3d_wgrad_f32"):
        res, args = _hip.SIGNATURES[name]
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True)
    ap.add_argument("--block", type=int, default=1, help="encoder block 1..2 at the C3 shape")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--env-b", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = 8
    ci, co, n = {1: (16, 32, 256), 2: (32, 32, 128)}[a.block]
    x = torch.rand(B, n, n, n, ci, device=dev) * 2 - 1
    g = torch.rand(B, n // 2, n // 2, n // 2, co, device=dev) * 2 - 1
    stream = torch.cuda.current_stream().cuda_stream
    libs = [(os.path.basename(p), load(p)) for p in a.libs.split(",")]
    envb = a.env_b.split("=", 1) if a.env_b else None
    if envb:
        libs.append((libs[0][0] + " " + a.env_b, libs[0][1]))
    nblk = 1024
    outs = []
    for name, h in libs:
        npart = h.lr_conv3d_wgrad_partial_floats(ci, co, _hip.LAYOUT_NDHWC_HPS, nblk)
        outs.append((torch.empty(npart, device=dev), torch.empty(co, ci, 3, 3, 3, device=dev), torch.empty(co, device=dev)))

    def run(i):
        name, h = libs[i]
        if envb:
            if i == len(libs) - 1:
                os.environ[envb[0]] = envb[1]
            else:
                os.environ.pop(envb[0], None)
        p, gw, gb = outs[i]
        rc = h.lr_conv3d_wgrad_f32(x.data_ptr(), _hip.LAYOUT_NDHWC_HPS, g.data_ptr(), p.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, ci, co,
                                   n, n, n, 2, nblk, stream)
        assert rc == 0, (name, rc)

    for i in range(len(libs)):
        run(i)
    torch.cuda.synchronize()
    for i in range(1, len(libs)):
        dw = float((outs[0][1] - outs[i][1]).abs().max() / outs[0][1].abs().max())
        db = float((outs[0][2] - outs[i][2]).abs().max() / outs[0][2].abs().max())
        print(f"{libs[i][0]}: max rel diff vs {libs[0][0]}: gw {dw:.2e}  gb {db:.2e}")
    times = [[] for _ in libs]
    for r in range(a.rounds):
        for i in range(len(libs)):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iters):
                run(i)
            e.record()
            torch.cuda.synchronize()
            times[i].append(s.elapsed_time(e) / a.iters)
    flops = 2.0 * 27 * ci * co * B * (n // 2) ** 3
    for (name, _), t in zip(libs, times):
        med = float(np.median(t))
        print(f"{name:40s} median {med:7.4f} ms  min {min(t):7.4f} ms  {flops / med / 1e9:7.1f} TFLOP/s  {flops / med / 1e9 / 157.3:6.1%}")


if __name__ == "__main__":
    main()
