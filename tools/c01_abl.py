"""Diagnostic: timing-only ablations of the fused blocks-0+1 kernel (stamped build: `make -C liftreg_amd/csrc stamps`;
WRONG results by design).  Prints ms per launch for each LIFTREG_C01_ABL value."""
import os, sys
import subprocess
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
names = {0: "everything", 1: "B: no MFMAs", 2: "A idle", 3: "A idle, B no MFMAs (staging + barriers only)", 8: "B: no fragment reads",
         128: "B: no split work in the staging", 129: "B: no MFMAs, no split work",
         # A alone = 137 (B: no MFMAs, no fragment reads, no split work); its parts (the chains stay alive: one store per tile /
         # opaque fragments): NOTE the skeleton alone (157) is bound by the HBM latency of the staging loads - shorter steps than that
         # cannot be read off these builds
         137: "A alone", 141: "A alone: no epilogue", 169: "A alone: no fragment reads", 173: "A alone: MFMAs only",
         153: "A alone: no MFMAs", 157: "skeleton (barriers, staging loads, B's exchange)"}
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    from liftreg_amd import _hip
    v = sys.argv[2]
    if v != "0":
        _hip.LIB_PATH = os.path.join(_hip.CSRC, f"libliftreg_hip_abl{v}.so" if v.isdigit() else f"libliftreg_hip_{v}.so")
    v = int(v) if v.isdigit() else v
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
    for rep in range(2):
        for _ in range(3):
            ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, packed=pk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, packed=pk)
        e1.record()
        torch.cuda.synchronize()
        print(f"abl {str(v):>8s} {names.get(v, ''):48s} {e0.elapsed_time(e1) / 5:7.3f} ms", flush=True)
    sys.exit(0)
vals = sys.argv[1:] or [0] + sorted(int(f.split("abl")[1].split(".")[0]) for f in os.listdir(os.path.join(ROOT, "liftreg_amd", "csrc")) if f.startswith("libliftreg_hip_abl"))
for v in vals:
    subprocess.call([sys.executable, os.path.abspath(__file__), "--one", str(v)])
