"""Loss plugin with the reference's interface (src/liftreg/losses/SubspaceLoss.py:10-67):

    "loss_class": "liftreg_amd.losses.SubspaceLoss.loss",  "loss": {"sim_class": "liftreg_amd.layers.losses.NCCLoss", …}

total = sim(warped, target) + reg_factor(epoch) · reg(params); reg on the HIP one-pass kernel.
The regulariser's finite-difference stencil is mermaid's (un-vendored, absent): PARITY UNPINNED —
see liftreg_amd/csrc/reg.hip for the assumed stencil.  `total_loss.backward()` runs the HIP backward kernels.
"""
import torch.nn as nn

from ..autograd import DispRegFn
from ..utils.general import get_class
from ..utils.utils import sigmoid_decay


def _opt(opt, key, default, comment=""):
    """ParameterDict-style `opt[(key, default, comment)]` with a plain-dict fallback."""
    try:
        return opt[(key, default, comment)]
    except (KeyError, TypeError):
        return opt.get(key, default) if hasattr(opt, "get") else default


class loss(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.sim_factor = 1.
        self.sim = get_class(_opt(opt, "sim_class", "liftreg_amd.layers.losses.NCCLoss", "Similarity class"))()
        self.initial_reg_factor = _opt(opt, 'initial_reg_factor', 10, 'initial regularization factor')
        self.min_reg_factor = _opt(opt, 'min_reg_factor', 1e-3, 'minimum regularization factor')
        self.reg_factor_decay_from = _opt(opt, 'reg_factor_decay_from', 10,
                                          'regularization factor starts to decay from # epoch')

    def forward(self, input):
        warped, target, params = input["warped"], input["target"], input["params"]
        epoch = input["epoch"]
        sim_loss = self.sim(warped, target)
        reg_loss = self.compute_reg_loss(params)
        total_loss = self.sim_factor * sim_loss + self.get_reg_factor(epoch) * reg_loss
        return {"total_loss": total_loss, "sim_loss": sim_loss.item(), "reg_loss": reg_loss.item()}

    def get_reg_factor(self, epoch):
        decay_factor = 2
        return float(max(sigmoid_decay(epoch, static=self.reg_factor_decay_from, k=decay_factor) *
                         self.initial_reg_factor, self.min_reg_factor))

    def compute_reg_loss(self, affine_param):
        return DispRegFn.apply(affine_param)
