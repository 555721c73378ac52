"""Network building blocks used by the configured model, with the reference's class names,
constructor arguments and state-dict keys (src/liftreg/layers/layers.py), running on the
HIP kernels, forward and backward (liftreg_amd.autograd): `loss.backward()` of the reference's training
step reaches the parameters through the HIP backward kernels, never through another backend."""
import math
import numbers

import torch
import torch.nn as nn

from .. import ops
from ..autograd import ConvBlockFn, LinearFn, NCCFn
from .._hip import NCC_SQUARED


class convBlock(nn.Module):
    """Conv3d(k3) + LeakyReLU(0.2) (layers/layers.py:335-372) on the fp32 MFMA implicit GEMM.

    `in_layout`/`out_layout` select NCDHW (the reference's layout, default) or channels-last
    NDHWC for activations that stay inside the encoder.
    """

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True,
                 batchnorm=False, residual=False, nonlinear=nn.LeakyReLU(0.2),
                 in_layout=ops.LAYOUT_NCDHW, out_layout=ops.LAYOUT_NCDHW):
        super().__init__()
        if kernel_size != 3 or padding != 1 or batchnorm or residual:
            raise NotImplementedError("only the configuration the model uses is built: k3, p1, no BN/residual")
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        self.bn = None
        self.nonlinear = nonlinear
        self.residual = None
        self.stride = stride
        self.in_layout, self.out_layout = in_layout, out_layout
        # backward chaining inside an encoder (set by the model, see autograd.ConvBlockFn)
        self.premasked_grad = False
        self.mask_input_slope = None
        if nonlinear is None:
            self._slope = 1.0
        elif isinstance(nonlinear, nn.LeakyReLU):
            self._slope = float(nonlinear.negative_slope)
        else:
            raise NotImplementedError("fused epilogue supports LeakyReLU or None")

    def forward(self, x, packed=None):
        # Training, chained blocks: a 16-channel block whose consumer applies this block's LeakyReLU mask in its data
        # gradient (premasked_grad) also writes that mask as one byte per channel quad; it travels to the consumer as an
        # attribute of the activation tensor (`_lr_sign4`) and replaces the consumer's re-read of the activation.
        mask_out = None
        x_sign4 = getattr(x, "_lr_sign4", None) if self.mask_input_slope is not None else None
        if (self.premasked_grad and torch.is_grad_enabled() and self.conv.weight.requires_grad and
                ops.conv3d_mask_supported(x, self.conv.weight, self.stride, self.in_layout, self.out_layout)):
            B, _, D, W, H = x.shape
            mask_out = torch.empty((B, D, W, H, self.conv.out_channels // 4), dtype=torch.uint8, device=x.device)
        y = ConvBlockFn.apply(x, self.conv.weight, self.conv.bias, self.stride, self.in_layout, self.out_layout,
                              self._slope, packed, self.premasked_grad, self.mask_input_slope, mask_out, x_sign4)
        if mask_out is not None:
            y._lr_sign4 = mask_out
        return y


class FullyConnectBlock(nn.Module):
    """Linear + LeakyReLU(0.2) (layers/layers.py:413-439)."""

    def __init__(self, in_channels, out_channels, bias=True, nonlinear=nn.LeakyReLU(0.2)):
        super().__init__()
        self.fc = nn.Linear(in_channels, out_channels, bias=bias)
        self.nonlinear = nonlinear
        if nonlinear is None:
            self._slope = 1.0
        elif isinstance(nonlinear, nn.LeakyReLU):
            self._slope = float(nonlinear.negative_slope)
        else:
            raise NotImplementedError("fused epilogue supports LeakyReLU or None")

    def forward(self, x):
        return LinearFn.apply(x, self.fc.weight, self.fc.bias, self._slope)


class NCCLoss(nn.Module):
    """Squared, per-channel NCC variant (layers/layers.py:238-255)."""

    def forward(self, x, y):
        return NCCFn.apply(x, y, NCC_SQUARED)


class GaussianSmoothing(nn.Module):
    """Holds the Gaussian kernel buffer the reference model registers but never applies
    (layers/layers.py:441-504; constructed at …Backproj.py:24) so that `state_dict()` carries
    the same `gaussian_smooth.weight` entry and reference checkpoints load strictly."""

    def __init__(self, channels, kernel_size, sigma, dim=2):
        super().__init__()
        if isinstance(kernel_size, numbers.Number):
            kernel_size = [kernel_size] * dim
        if isinstance(sigma, numbers.Number):
            sigma = [sigma] * dim
        grids = torch.meshgrid([torch.arange(s, dtype=torch.float32) for s in kernel_size], indexing="ij")
        kernel = torch.ones(())
        for size, std, g in zip(kernel_size, sigma, grids):
            kernel = kernel * (1 / (std * math.sqrt(2 * math.pi)) * torch.exp(-((g - (size - 1) / 2) / std) ** 2 / 2))
        kernel = kernel / torch.sum(kernel)
        self.register_buffer('weight', kernel.view(1, 1, *kernel.size()).repeat(channels, *[1] * (kernel.dim() + 1)))
        self.groups = channels

    def forward(self, input):
        raise NotImplementedError("unused by the configured model (never called in the reference either)")
