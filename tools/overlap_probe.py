"""Probe (development aid): does an HBM-streaming kernel on a second HIP stream run INSIDE the fused pair kernel's
shadow?  The pair kernel holds every CU with 8 waves x ~208 VGPRs and 151 KB of LDS but moves < 1 TB/s; a light
streaming kernel (no LDS, few registers) could in principle co-reside as a third wave per SIMD.
Prints: pair alone, stream alone, both concurrently (wall)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    B, N = 8, 256
    x0 = torch.rand(B, 1, N, N, N, device=dev, generator=g)
    rest = torch.randn(B, 2, N, N, N, device=dev, generator=g)
    w0 = torch.randn(16, 3, 3, 3, 3, device=dev, generator=g) * (2.0 / 81) ** 0.5
    b0 = torch.randn(16, device=dev, generator=g) * 0.1
    w1 = torch.randn(32, 16, 3, 3, 3, device=dev, generator=g) * (2.0 / 432) ** 0.5
    b1 = torch.randn(32, device=dev, generator=g) * 0.1
    pkp = ops.conv3d_pair01_pack(w0, w1)
    hps = ops.LAYOUT_NDHWC_HPS
    out = ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, out_layout=hps, packed=pkp)
    src = torch.randn(1 << 30, device=dev)   # 4 GB
    dst = torch.empty_like(src)
    nfill_box = [3]

    def pair():
        return ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, out_layout=hps, packed=pkp)

    def fill():
        for _ in range(nfill_box[0]):
            torch.add(src, 1.0, out=dst)

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fa, fb):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        cur = torch.cuda.current_stream()
        ev[0].record(cur)
        s1.wait_stream(cur); s2.wait_stream(cur)
        if fa:
            with torch.cuda.stream(s1):
                ev[2].record(s1); fa(); ev[3].record(s1)
        if fb:
            with torch.cuda.stream(s2):
                ev[4].record(s2); fb(); ev[5].record(s2)
        cur.wait_stream(s1); cur.wait_stream(s2)
        ev[1].record(cur)
        torch.cuda.synchronize()
        return (ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3]) if fa else 0.0, ev[4].elapsed_time(ev[5]) if fb else 0.0)

    # the plain backprojection kernel (28 VGPRs: LIFTREG_BP_LIGHT) and the NCC moments in the pair kernel's shadow
    import numpy as np
    from liftreg_amd import _hip
    proj = torch.rand(B, 2, 256, 256, device=dev, generator=g)
    poses = np.array([[0.0, -160.0, 0.0], [-160.0, 0.0, 0.0]], np.float32)
    tv = torch.empty(B, 2, N, N, N, device=dev)
    wv = torch.rand(B, 1, N, N, N, device=dev, generator=g)

    light_box = [False]

    def bp():
        ops.backproject(proj, poses, (N, N, N), out=tv, out_batch_stride=2 * N ** 3, light=light_box[0])

    def ncc():
        ops.ncc_moments(wv, x0, B)

    for light in ("0", "1"):
        light_box[0] = light == "1"
        for name, fn in (("backproject", bp), ("ncc_moments", ncc)):
            for _ in range(2):
                timed(pair, fn)
            for rep in range(2):
                a = timed(pair, None)
                b = timed(None, fn)
                c = timed(pair, fn)
                print(f"BP_LIGHT={light} {name}: pair alone {a[0]:.3f} | {name} alone {b[0]:.3f} | both wall {c[0]:.3f} (pair {c[1]:.3f}, {name} {c[2]:.3f}) | sum {a[0] + b[0]:.3f}")

    for nf in (2,):
        nfill_box[0] = nf
        for _ in range(2):
            timed(pair, fill)
        for rep in range(2):
            a = timed(pair, None)
            b = timed(None, fill)
            c = timed(pair, fill)
            print(f"nfill {nf} ({nf * 8} GB): pair alone {a[0]:.3f} | stream alone {b[0]:.3f} | both wall {c[0]:.3f} (pair {c[1]:.3f}, stream {c[2]:.3f}) | sum {a[0] + b[0]:.3f}")


if __name__ == "__main__":
    main()
