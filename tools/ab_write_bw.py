#!/usr/bin/env python3
"""What streaming-WRITE rate does this part sustain, with and without the Infinity Cache's help?  A 2 GiB fill (torch's
vectorized fill kernel: 16-byte stores, nothing read) repeated back to back (the buffer's lines written by the launch before are
partly still in the 256 MB memory-side cache and are overwritten there) and with another kernel's 2 GiB stream in between (the
in-step situation of lr_backproject_f32: tools/ab_backproject.py --pollute).  Also the library's own non-temporal stores (lr_backproject_f32
with its taps) for comparison are in ab_backproject.py."""
import numpy as np
import torch
dev = torch.device("cuda:0")
N = 2 * 1024 ** 3 // 4
a, b, c = (torch.empty(N, device=dev) for _ in range(3))
def timed(fn, pre=None, reps=20):
    ts = []
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    for _ in range(reps):
        if pre is not None:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
gib = N * 4 / 1e9
for name, fn, pre, nbytes in (
        ("fill 2 GiB, back to back", lambda: a.fill_(1.0), None, gib),
        ("fill 2 GiB after a 2 GiB read of another buffer", lambda: a.fill_(1.0), lambda: b.sum(), gib),
        ("fill 2 GiB after a 2 GiB fill of another buffer", lambda: a.fill_(1.0), lambda: b.fill_(2.0), gib),
        ("copy 2 GiB -> 2 GiB, back to back", lambda: a.copy_(b), None, 2 * gib),
        ("copy 2 GiB -> 2 GiB after a fill of a third buffer", lambda: a.copy_(b), lambda: c.fill_(1.0), 2 * gib),
        ("read (sum) 2 GiB, back to back", lambda: b.sum(), None, gib),
        ("read (sum) 2 GiB after a fill of another buffer", lambda: b.sum(), lambda: c.fill_(1.0), gib)):
    ms = timed(fn, pre)
    print(f"{name:55s} {ms:.4f} ms  {nbytes / ms:.2f} TB/s", flush=True)
