"""Similarity losses (src/liftreg/layers/losses.py)."""
import torch
import torch.nn as nn

from ..autograd import NCCFn
from .._hip import NCC_CONFIGURED


class NCCLoss(nn.Module):
    """Configured NCC similarity (layers/losses.py:14-29; cur_task_setting.json:51):
    1 - mean_b( mean(ab) / sqrt(mean(a²)·mean(b²)) ), a = x-mean(x)+1e-10.

    One streaming pass over both volumes on the GPU.  Like the reference it asserts the
    result is not NaN (a tiny D2H read); pass check_nan=False to stay asynchronous.
    """

    def __init__(self, check_nan=True):
        super().__init__()
        self.check_nan = check_nan

    def forward(self, input, target, moments=None):
        """`moments` (optional, non-reference): the (B,5) fp64 moments of exactly (input, target) when the model's one-pass
        decode already accumulated them (output key "ncc_moments": training with ncc_grad_via_moments, or the opt key fuse_ncc) — the
        pass over both volumes is skipped.  The caller vouches that they describe THESE tensors (SubspaceLoss checks
        "ncc_moments_of")."""
        loss = NCCFn.apply(input, target, NCC_CONFIGURED, moments)
        if self.check_nan:
            assert not torch.isnan(loss), 'NCC loss is Nan.'
        return loss
