"""Is the fp32 encoder bound by the chip's power budget?  On the GPU box: block 1 (16 -> 32, stride 2, the Winograd rows
kernel) timed with HIP events (a) back to back with itself, (b) behind the default first block, (c) behind the split
first block (LIFTREG_CONV0_SPLIT=1: 0.3 ms faster), (d) behind a low-power spin of the same length as the first block,
(e) behind a 6 ms spin.  If (c) > (b) and (d),(e) < (b), a faster neighbour is paid back by a lower clock."""
import os
os.environ.setdefault("LIFTREG_SWITCH_AUTOSYNC", "1")   # this tool flips library switches between calls, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops

dev = torch.device("cuda:0")
B, n = 8, 256
g = torch.Generator().manual_seed(1)
x0 = torch.rand(B, 1, n, n, n, generator=g).to(dev)
rest = torch.randn(B, 2, n, n, n, generator=g).to(dev)
w0 = (torch.randn(16, 3, 3, 3, 3, generator=g) * 0.15).to(dev)
b0 = (torch.randn(16, generator=g) * 0.1).to(dev)
w1 = (torch.randn(32, 16, 3, 3, 3, generator=g) * 0.07).to(dev)
b1 = (torch.randn(32, generator=g) * 0.1).to(dev)
y0 = torch.empty(B, n, n, n, 16, device=dev)
y1 = torch.empty(B, n // 2, n // 2, n // 2, 32, device=dev)
p0 = ops.conv3d_pack_weights(w0, ops.LAYOUT_NCDHW)
p1 = ops.conv3d_pack_weights(w1, ops.LAYOUT_NDHWC_HPS)


def blk0():
    ops.conv3d_first_split(x0, rest, w0, b0, out_layout=ops.LAYOUT_NDHWC_HPS, packed=p0, out=y0)


def blk1():
    ops.conv3d_k3_lrelu(y0, w1, b1, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=ops.LAYOUT_NDHWC_HPS, packed=p1, out=y1)


def spin(ms):
    torch.cuda._sleep(int(ms * 2.0e6))   # cycles of a ~2 GHz clock: a one-wave spin, next to no power


def run(pre, reps=40):
    for _ in range(8):
        pre(); blk1()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b, c in ev:
        a.record(); pre(); b.record(); blk1(); c.record()
    torch.cuda.synchronize()
    t0 = sorted(a.elapsed_time(b) for a, b, c in ev)[reps // 2]
    t1 = sorted(b.elapsed_time(c) for a, b, c in ev)[reps // 2]
    return t0, t1


blk0()
os.environ.pop("LIFTREG_CONV0_SPLIT", None)
for name, pre, env in (("block 1 back to back", lambda: None, None), ("behind the default first block", blk0, None),
                       ("behind the split first block", blk0, "1"), ("behind a 3 ms spin", lambda: spin(3.0), None),
                       ("behind a 6 ms spin", lambda: spin(6.0), None), ("behind the default first block (again)", blk0, None)):
    if env:
        os.environ["LIFTREG_CONV0_SPLIT"] = env
    else:
        os.environ.pop("LIFTREG_CONV0_SPLIT", None)
    t0, t1 = run(pre)
    print(f"{name:42s} predecessor {t0:6.3f} ms   block 1 {t1:6.3f} ms   pair {t0 + t1:6.3f} ms", flush=True)
