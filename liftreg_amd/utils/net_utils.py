"""Spatial transformer and identity maps with the reference's interface
(src/liftreg/utils/net_utils.py:9-125), backed by the HIP warp kernel."""
import numpy as np
import torch
from torch.nn import Module

from .. import ops

dim = 3


def identity_axis_tables(sz, dtype=np.float32):
    """Per-axis 1-D tables whose outer broadcast equals identity_map(sz).

    Same numpy arithmetic as the reference (net_utils.py:80-85): float32 index times the
    float64 spacing 1/(sz-1) (rounded to float32), then *2-1 in float32.
    """
    spacing = 1. / (np.array(sz) - 1)
    tabs = []
    for d in range(len(sz)):
        t = np.arange(sz[d]).astype(dtype)
        t *= spacing[d]
        t = t * 2 - 1
        tabs.append(np.ascontiguousarray(t.astype(np.float32)))
    return tabs


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("liftreg_amd needs a GPU (the reference hard-codes .cuda() here too, net_utils.py:87)")
    return torch.device("cuda", torch.cuda.current_device())


def identity_map(sz, dtype=np.float32, device=None):
    """(dim, X, Y, Z) identity map in [-1,1] (net_utils.py:59-87), on the GPU."""
    if len(sz) != 3:
        raise ValueError("Only 3-D identity maps are on this path")
    device = device or _default_device()
    t0, t1, t2 = (torch.from_numpy(t).to(device) for t in identity_axis_tables(sz, dtype))
    out = torch.empty((3,) + tuple(sz), dtype=torch.float32, device=device)
    out[0] = t0[:, None, None]
    out[1] = t1[None, :, None]
    out[2] = t2[None, None, :]
    return out


def not_normalized_identity_map(sz, device=None):
    """Voxel-index identity map (net_utils.py:90-110)."""
    device = device or _default_device()
    axes = [torch.arange(n, dtype=torch.float32, device=device) for n in sz]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"))


def gen_identity_map(img_sz, resize_factor=1., normalized=True, device=None):
    """net_utils.py:113-125."""
    if isinstance(resize_factor, list):
        img_sz = [int(img_sz[i] * resize_factor[i]) for i in range(dim)]
    else:
        img_sz = [int(img_sz[i] * resize_factor) for i in range(dim)]
    return identity_map(img_sz, device=device) if normalized else not_normalized_identity_map(img_sz, device)


class Bilinear(Module):
    """Spatial transform in BCXYZ format (net_utils.py:9-56): `forward(img, phi)`.

    phi is the deformation map in [-1,1] with channel c ↔ spatial axis c; the channel
    reorder (2,1,0), the (I+1)/2 … *2-1 intensity scaling and the trilinear/nearest
    sampling all run in one HIP kernel.
    """

    def __init__(self, zero_boundary=False, using_scale=True, mode="bilinear"):
        super().__init__()
        self.zero_boundary = 'zeros' if zero_boundary else 'border'
        self.using_scale = using_scale
        self.mode = mode

    def forward_stn(self, input1, input2):
        _, out = ops.warp(input1, input2, None, None, using_scale=False,
                          zero_boundary=self.zero_boundary == 'zeros', mode=self.mode, want_phi=False)
        return out

    def forward(self, input1, input2):
        _, out = ops.warp(input1, input2, None, None, using_scale=self.using_scale,
                          zero_boundary=self.zero_boundary == 'zeros', mode=self.mode, want_phi=False)
        return out
