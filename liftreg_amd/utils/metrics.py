"""Overlap metrics with the reference's formulas (src/liftreg/utils/metrics.py:83-121); the counting pass runs on the GPU."""
import torch

from .. import ops


def cal_metric(label_pred, label_gt, label=1):
    """{'iou','dice','recall','precision'} of the binary masks `== label` (same eps and empty-set rules)."""
    eps = 1e-11
    pred = label_pred if isinstance(label_pred, torch.Tensor) else torch.as_tensor(label_pred)
    gt = label_gt if isinstance(label_gt, torch.Tensor) else torch.as_tensor(label_gt)
    pred = pred.to("cuda", torch.float32).contiguous()
    gt = gt.to("cuda", torch.float32).contiguous()
    n_pred, n_gt, n_both = ops.label_overlap(pred, gt, float(label)).tolist()
    tp, fn, fp = float(n_both), float(n_gt - n_both), float(n_pred - n_both)
    union = n_pred + n_gt - n_both
    if n_gt != 0:
        return {'iou': tp / (float(union) + eps), 'dice': 2 * tp / (2 * tp + fn + fp + eps),
                'recall': tp / (tp + fn + eps), 'precision': tp / (tp + fp + eps)}
    v = 0. if n_pred > 0 else 1.
    return {'iou': v, 'dice': v, 'recall': v, 'precision': v}
