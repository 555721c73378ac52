#!/usr/bin/env python3
"""Interleaved A/B timing of conv-kernel builds in ONE process (cdna_hip_programming.md §5.4 rule 24): every library in
--libs runs the same launch in turn, several rounds; prints median / min per library and checks the outputs agree.

  python tools/abconv.py --block 0 --libs liftreg_amd/csrc/libliftreg_hip.so,liftreg_amd/csrc/libx_A.so [--rounds 7]
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
    for name in ("lr_conv3d_packed_floats", "lr_conv3d_pack_weights_f32", "lr_conv3d_k3_lrelu_f32", "lr_conv3d_first_split_f32"):
        res, args = _hip.SIGNATURES[name]
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", required=True)
    ap.add_argument("--block", type=int, default=0, help="encoder block 0..2 at the C3 shape")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--env-b", default="", help="NAME=VALUE: time the FIRST library a second time with this variable set (kernels read their A/B switches per launch)")
    ap.add_argument("--split", action="store_true", help="block 0 through lr_conv3d_first_split_f32 (the model's inference path)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    rnd = lambda *s: torch.rand(*s, generator=g, device=dev) * 2 - 1
    B, n = a.B, a.n
    chans = [(3, 16, 1, n), (16, 32, 2, n), (32, 32, 2, n // 2)]
    ci, co, st, size = chans[a.block]
    so = (size - 1) // st + 1
    if a.block == 0:
        x = rnd(B, ci, size, size, size)
        lin = _hip.LAYOUT_NCDHW
    else:
        x = rnd(B, size, size, size, ci)
        lin = _hip.LAYOUT_NDHWC_HPS
    lout = _hip.LAYOUT_NDHWC_HPS
    w = rnd(co, ci, 3, 3, 3) / (27 * ci) ** 0.5
    bias = rnd(co) * 0.1
    stream = torch.cuda.current_stream().cuda_stream
    libs = [(os.path.basename(p), load(p)) for p in a.libs.split(",")]
    envb = None
    if a.env_b:
        envb = a.env_b.split("=", 1)
        libs.append((libs[0][0] + " " + a.env_b, libs[0][1]))
    outs, packs = [], []
    for name, h in libs:
        npk = h.lr_conv3d_packed_floats(ci, co, lin)
        pk = torch.empty((npk,), device=dev)
        assert h.lr_conv3d_pack_weights_f32(w.data_ptr(), pk.data_ptr(), ci, co, lin, stream) == 0
        packs.append(pk)
        outs.append(torch.empty((B, so, so, so, co), device=dev))
    x0 = x[:, :1].contiguous() if a.split else None
    xr = x[:, 1:].contiguous() if a.split else None

    def run(i):
        name, h = libs[i]
        if envb:
            if i == len(libs) - 1:
                os.environ[envb[0]] = envb[1]
            else:
                os.environ.pop(envb[0], None)
        if a.split:
            rc = h.lr_conv3d_first_split_f32(x0.data_ptr(), xr.data_ptr(), packs[i].data_ptr(), bias.data_ptr(), outs[i].data_ptr(),
                                             B, ci, co, size, size, size, lout, 0.2, stream)
        else:
            rc = h.lr_conv3d_k3_lrelu_f32(x.data_ptr(), packs[i].data_ptr(), bias.data_ptr(), outs[i].data_ptr(), B, ci, co, size,
                                          size, size, st, lin, lout, 0.2, stream)
        assert rc == 0, (name, rc)

    for i in range(len(libs)):
        run(i)
    torch.cuda.synchronize()
    for i in range(1, len(libs)):
        same = torch.equal(outs[0], outs[i])
        md = float((outs[0] - outs[i]).abs().max())
        print(f"{libs[i][0]}: output {'identical to' if same else 'DIFFERS from'} {libs[0][0]} (max abs diff {md:.3e})")
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
    flops = 2.0 * 27 * ci * co * B * so ** 3
    for (name, _), t in zip(libs, times):
        med, mn = float(np.median(t)), float(np.min(t))
        print(f"{name:36s} median {med:7.4f} ms  min {mn:7.4f} ms  {flops / med / 1e9:7.1f} TFLOP/s  {flops / med / 1e9 / 157.3:6.1%}")


if __name__ == "__main__":
    main()
