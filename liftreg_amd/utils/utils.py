"""Output helpers with the reference's file formats (src/liftreg/utils/utils.py)."""
import os

import numpy as np


def save_deformations(phis, fname_list, output_path):
    """`{id}_phi.npy` = (phi+1)/2 as float32 (B,3,D,W,H → one file per sample), utils/utils.py:57-68.

    The reference also writes a `.nii.gz` through nibabel (un-installed here); only the `.npy` the
    evaluation scripts read (tools/evaluate_dir_lab.py:176-178) is produced.
    """
    phis = (phis.detach().cpu().numpy() if hasattr(phis, "detach") else np.asarray(phis))
    phis = (phis + 1.) / 2.
    for i, name in enumerate(fname_list):
        np.save(os.path.join(output_path, f"{name}_phi.npy"), phis[i].astype(np.float32))


def sigmoid_decay(ep, static=5, k=5):
    """factor = k/(k+exp((ep-static)/k)) after `static` epochs, 1 before (utils/utils.py:93-107)."""
    if ep < static:
        return float(1.)
    ep = ep - static
    return float(k / (k + np.exp(ep / k)))


def compute_jacobi_map(map, spacing, crop_boundary=True, use_01=False):
    """(mean over the batch of Σ|det J| where det J < 0, mean number of folded voxels) — utils/utils.py:20-55.

    Like the reference, the cropped statistics are computed-and-discarded there (`:46-52` overwrite them), so the
    values returned are over the whole volume.  One pass on the GPU (`lr_jacobi_det_stats_f32`); the derivative
    stencil is the assumed mermaid one (parity unpinned, see csrc/metrics.hip)."""
    import torch

    from .. import ops
    t = map if isinstance(map, torch.Tensor) else torch.from_numpy(np.asarray(map, dtype=np.float32))
    if not t.is_cuda:
        t = t.to("cuda")
    span = 1.0 if use_01 else 2.0
    sp = np.asarray(spacing, dtype=np.float64) * span
    abs_sum, count = ops.jacobi_det_stats(t.float().contiguous(), sp).tolist()
    return abs_sum / t.shape[0], count / t.shape[0]
