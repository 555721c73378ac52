"""The loss plugin of the subspace model on the HIP kernels.

Drop-in for `train.loss_class` (interface of the reference's src/liftreg/losses/SubspaceLoss.py:10-67):

    "loss_class": "liftreg_amd.losses.SubspaceLoss.loss",
    "loss": {"sim_class": "liftreg_amd.layers.losses.NCCLoss", "initial_reg_factor": …, …}

    total = similarity(warped, target) + λ(epoch) · R(params)
    λ(epoch) = max(sigmoid_decay(epoch, static=reg_factor_decay_from, k=2) · initial_reg_factor, min_reg_factor)
    R        = mean over voxels of Σ_{component, axis} (∂_axis disp_component)²      (one streaming HIP pass — or, when the
               model hands over "pca_coefs" and "pca_reg_gram", the same quadratic form evaluated on the coefficients)

R's finite-difference stencil lives in `mermaid` (un-vendored, absent from the reference checkout): PARITY UNPINNED,
the assumed stencil is documented in liftreg_amd/csrc/reg.hip.  Both terms are autograd nodes whose backward runs
HIP kernels (liftreg_amd.autograd), so `out["total_loss"].backward()` works as in the reference's training step.
"""
import torch.nn as nn

from ..autograd import DispRegFn, SubspaceRegFn
from ..utils.general import get_class
from ..utils.utils import sigmoid_decay

# option name → (default, description); read once at construction, ParameterDict- or dict-style
_OPTIONS = {
    "sim_class": ("liftreg_amd.layers.losses.NCCLoss", "Similarity class"),
    "initial_reg_factor": (10, "initial regularization factor"),
    "min_reg_factor": (1e-3, "minimum regularization factor"),
    "reg_factor_decay_from": (10, "regularization factor starts to decay from # epoch"),
}
_DECAY_K = 2


def _read(opt, name):
    default, text = _OPTIONS[name]
    try:                                   # ParameterDict registers the default and the comment on first access
        return opt[(name, default, text)]
    except (KeyError, TypeError):          # plain mapping
        getter = getattr(opt, "get", None)
        return getter(name, default) if getter else default


class loss(nn.Module):
    """`forward(model_output_dict_with_epoch) -> {"total_loss": Tensor, "sim_loss": float, "reg_loss": float}`."""

    def __init__(self, opt):
        super().__init__()
        for name in ("initial_reg_factor", "min_reg_factor", "reg_factor_decay_from"):
            setattr(self, name, _read(opt, name))
        self.sim = get_class(_read(opt, "sim_class"))()
        try:
            import inspect
            self._sim_takes_moments = "moments" in inspect.signature(self.sim.forward).parameters
        except (TypeError, ValueError):
            self._sim_takes_moments = False
        self.sim_factor = 1.

    def get_reg_factor(self, epoch):
        """λ(epoch): constant `initial_reg_factor` until `reg_factor_decay_from`, then a sigmoid decay, floored."""
        scheduled = sigmoid_decay(epoch, static=self.reg_factor_decay_from, k=_DECAY_K) * self.initial_reg_factor
        return float(max(scheduled, self.min_reg_factor))

    def compute_reg_loss(self, disp):
        """R(disp) for a (B,3,D,W,H) displacement field in normalised coordinates."""
        return DispRegFn.apply(disp)

    def forward(self, input):
        get = input.get if hasattr(input, "get") else (lambda k: None)
        moments = get("ncc_moments")
        if moments is not None and not self._sim_takes_moments:
            moments = None               # a similarity class without the (non-reference) `moments` argument: the plain call
        if moments is not None:
            # the decode node already accumulated them (model opt keys ncc_grad_via_moments, fuse_ncc) — for the tensors named
            # in "ncc_moments_of".  A caller that masked or replaced warped / target between model and loss gets the plain pass.
            of = get("ncc_moments_of")
            if of is None or input["warped"] is not of[0] or of[0]._version != of[1] or input["target"] is not of[2] or \
                    of[2]._version != of[3]:
                moments = None
        if moments is not None:
            similarity = self.sim(input["warped"], input["target"], moments=moments)
        else:
            similarity = self.sim(input["warped"], input["target"])
        gram = get("pca_reg_gram")
        of = get("pca_reg_gram_of")
        if gram is not None and of is not None and input["params"] is of[0] and of[0]._version == of[1] and \
                get("pca_coefs") is of[2] and of[2]._version == of[3]:
            # the subspace model in training: params = pca_coefs . basis^T + mean, so R(params) is a quadratic form of the
            # coefficients — the same number to fp32 rounding, without the passes over the field (model opt key
            # reg_in_coef_space); only while "params" and "pca_coefs" are the tensors the model produced, unmodified
            smoothness = SubspaceRegFn.apply(input["pca_coefs"], *gram)
        else:
            smoothness = self.compute_reg_loss(input["params"])
        total = self.sim_factor * similarity + self.get_reg_factor(input["epoch"]) * smoothness
        return {"total_loss": total, "sim_loss": similarity.item(), "reg_loss": smoothness.item()}
