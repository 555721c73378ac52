"""Diagnostic: per-phase cycle shares of the fused blocks-0+1 kernel (stamped build: `make -C liftreg_amd/csrc stamps`)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import _hip
_hip.LIB_PATH = os.path.join(_hip.CSRC, "libliftreg_hip_stamps.so")
from liftreg_amd import ops
dev = torch.device("cuda:0")
B, n = 8, 256
g = torch.Generator(device=dev).manual_seed(1)
x0 = torch.rand(B, 1, n, n, n, device=dev, generator=g)
rest = torch.randn(B, 2, n, n, n, device=dev, generator=g)
w0 = torch.randn(16, 3, 3, 3, 3, device=dev, generator=g) / 9
b0 = torch.randn(16, device=dev, generator=g) * 0.1
w1 = torch.randn(32, 16, 3, 3, 3, device=dev, generator=g) / 20
b1 = torch.randn(32, device=dev, generator=g) * 0.1
pk = ops.conv3d_pair01_pack(w0, w1)
lib = _hip.lib()
lib.lr_debug_read_c01_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 64)()
for _ in range(3):
    ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, packed=pk)
torch.cuda.synchronize(); lib.lr_debug_read_c01_stamps(buf, 1)
ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, packed=pk)
torch.cuda.synchronize(); lib.lr_debug_read_c01_stamps(buf, 0)
names = ["0 step top", "1 A: first fragments | B: staging", "2 A tile 0", "3 A tile 1 | B: MFMAs", "4 A tiles 3, 4 | B: partial", "5 A tail", "6 barrier", "7 A tile 2"]
steps = (256 * 8 * 2) * 129 / 256   # per block
for w in range(8):
    tot = sum(buf[w * 8 + i] for i in range(8))
    print(f"wave {w}: {tot / 256 / steps:8.0f} cycles per step")
    for i, nme in enumerate(names):
        print(f"   {nme:36s} {buf[w * 8 + i] / max(tot, 1):6.1%}   {buf[w * 8 + i] / 256 / steps:8.0f}")
