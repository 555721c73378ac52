"""Development aid: time the training form of the fused pair kernel (lr_conv3d_pair01_train_f32) beside the inference form."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liftreg_amd import _hip  # noqa: E402
if os.environ.get("LR_VARIANT"):      # a `make variant NAME=…` build of conv01_fused.hip (timing builds: LR_C01_SAVE_PARTS=0|1|2)
    _hip.LIB_PATH = os.path.join(_hip.CSRC, f"libliftreg_hip_{os.environ['LR_VARIANT']}.so")
from liftreg_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    B, N = 8, 256
    x = torch.randn(B, 3, N, N, N, device=dev, generator=g)
    x[:, 0] = x[:, 0].abs() * 0.3
    w0 = torch.randn(16, 3, 3, 3, 3, device=dev, generator=g) * (2.0 / 81) ** 0.5
    b0 = torch.randn(16, device=dev, generator=g) * 0.1
    w1 = torch.randn(32, 16, 3, 3, 3, device=dev, generator=g) * (2.0 / 432) ** 0.5
    b1 = torch.randn(32, device=dev, generator=g) * 0.1
    pk = ops.conv3d_pair01_pack(w0, w1)
    x0, rest = x[:, 0:1].contiguous(), x[:, 1:].contiguous()

    def inf():
        return ops.conv3d_pair01(x0, rest, w0, b0, w1, b1, packed=pk)

    def trn():
        return ops.conv3d_pair01_train(x, w0, b0, w1, b1, packed=pk)

    for rep in range(3):
        for name, fn in (("inference", inf), ("training form", trn)):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"rep {rep}: {name}: {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
