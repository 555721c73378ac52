"""A/B on the GPU box: the first block at C3 (B=8, 3 x 256^3 -> 16 channels, fp32 HPS output) — default fp32-MFMA Winograd kernel
against conv0_split_f32.hip (LIFTREG_CONV0_SPLIT=1); errors of both against an fp64 convolution on a crop."""
import os
os.environ.setdefault("LIFTREG_SWITCH_AUTOSYNC", "1")   # this tool flips library switches between calls, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import ops

dev = torch.device("cuda:0")
B, n = int(os.environ.get("AB_B", 8)), int(os.environ.get("AB_N", 256))
g = torch.Generator().manual_seed(1)
x0 = torch.rand(B, 1, n, n, n, generator=g).to(dev)
rest = torch.randn(B, 2, n, n, n, generator=g).to(dev)
w = (torch.randn(16, 3, 3, 3, 3, generator=g) * 0.15).to(dev)
b = (torch.randn(16, generator=g) * 0.1).to(dev)
out = torch.empty(B, n, n, n, 16, device=dev)
packed = ops.conv3d_pack_weights(w, ops.LAYOUT_NCDHW)


def run(native, reps=10):
    if native:
        os.environ.pop("LIFTREG_CONV0_SPLIT", None)
    else:
        os.environ["LIFTREG_CONV0_SPLIT"] = "1"
    for _ in range(3):
        ops.conv3d_first_split(x0, rest, w, b, out_layout=ops.LAYOUT_NDHWC_HPS, packed=packed, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv3d_first_split(x0, rest, w, b, out_layout=ops.LAYOUT_NDHWC_HPS, packed=packed, out=out)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ABL = [{"LIFTREG_C0S_ABL": str(a)} for a in (1, 2, 4, 8, 3, 12, 6, 9, 14, 7)] if os.environ.get("AB_ABL") else []
for env in [{}, {"LIFTREG_CONV0_SPLIT_BLOCKS": "512"}, {"LIFTREG_CONV0_SPLIT_CHUNKS": "1"}, {"LIFTREG_CONV0_SPLIT_CHUNKS": "2"}] + ABL:
    os.environ.update(env)
    print(f"split {env}: {run(False):.3f} ms", flush=True)
    for k in env:
        del os.environ[k]
print(f"fp32 MFMA (Winograd) kernel: {run(True):.3f} ms", flush=True)

# accuracy on a crop (whole planes of a thin slab, so the kernels run their real tiles)
xs0, xsr = x0[:1, :, 100:112].contiguous(), rest[:1, :, 100:112].contiguous()
ref = F.leaky_relu(F.conv3d(torch.cat([xs0, xsr], 1).double().cpu(), w.double().cpu(), b.double().cpu(), padding=1), 0.2)
for native in (False, True):
    if native:
        os.environ.pop("LIFTREG_CONV0_SPLIT", None)
    else:
        os.environ["LIFTREG_CONV0_SPLIT"] = "1"
    y = ops.hps_to_ndhwc(ops.conv3d_first_split(xs0, xsr, w, b, out_layout=ops.LAYOUT_NDHWC_HPS)).permute(0, 4, 1, 2, 3).double().cpu()
    err = (y - ref).abs()
    print(f"{'fp32 MFMA' if native else 'split    '}: max |err| {float(err.max()):.3e}  rms {float(err.pow(2).mean().sqrt()):.3e}  (scale {float(ref.abs().max()):.2f})")
