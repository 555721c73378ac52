"""Diagnostic: per-phase cycle shares of the persistent conv0 kernel (stamped build, see `make stamps`)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import _hip
_hip.LIB_PATH = os.path.join(_hip.CSRC, "libliftreg_hip_stamps.so")
from liftreg_amd import ops
dev = torch.device("cuda:0")
B, n = 8, 256
x = torch.rand((B, 3, n, n, n), device=dev) * 2 - 1
w = torch.randn((16, 3, 3, 3, 3), device=dev) / 9
b = torch.randn(16, device=dev) * 0.1
pk = ops.conv3d_pack_weights(w, ops.LAYOUT_NCDHW)
lib = _hip.lib()
lib.lr_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
for _ in range(2):
    ops.conv3d_k3_lrelu(x, w, b, 1, out_layout=ops.LAYOUT_NDHWC_HPS, packed=pk)
torch.cuda.synchronize(); lib.lr_debug_read_stamps(buf, 1)
ops.conv3d_k3_lrelu(x, w, b, 1, out_layout=ops.LAYOUT_NDHWC_HPS, packed=pk)
torch.cuda.synchronize(); lib.lr_debug_read_stamps(buf, 0)
names = ["sweep", "barrier A", "stage(+wait prefetch)", "stores", "setup", "barrier B", "prefetch issue"]
tot = sum(buf[i] for i in range(7))
for i, nme in enumerate(names):
    print(f"{nme:24s} {buf[i] / tot:6.1%}   {buf[i] / 512 / 256:10.0f} cycles per brick (avg over blocks)")
