"""Multi-GPU decomposition of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).

Two decompositions:

* **Replicas** (registrations/s at 1/2/4/8 GPUs): registrations and DRR cases are independent —
  `shard_items` deals them to ranks; no data-path collective exists or is invented.
* **z-slab sharding of ONE registration** along D (axis 0).  Every kernel takes the slab bounds
  [d0,d1) in its ABI.  Only two steps exchange data, both tiny and latency-bound:
    - DRR forward: each rank integrates the taps that fall in its slab → partial (P,Rd,Rh) images
      **sum** (`all_reduce`, 0.5 MB at C3) — the "slab-boundary partial sums" of the north star;
    - NCC: five fp64 moments per row **add** (`all_reduce`, 40·B bytes).
  Backprojection, PCA reconstruction (basis column slab) and the warp (output slab, whole moving
  volume replicated) need no collective.

The compute backend is the module-level name `ops` (liftreg_amd.ops — HIP only).
"""
import torch
import torch.distributed as dist

from . import ops
from ._hip import NCC_CONFIGURED


def world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def slab_bounds(D, world_size, rank):
    """Contiguous rows [d0,d1) of axis 0 for `rank`; the first D % world ranks get one extra row."""
    q, r = divmod(int(D), int(world_size))
    d0 = rank * q + min(rank, r)
    return d0, d0 + q + (1 if rank < r else 0)


def shard_items(n_items, world_size, rank):
    """Indices of the independent work items (registrations, DRR cases) this rank owns (round robin)."""
    return list(range(rank, int(n_items), int(world_size)))


def _all_reduce_sum(t, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def drr_forward_sharded(vol_slab, poses, resolution, spacing, D, d0, d1, group=None, **kw):
    """DRR of a volume whose rows [d0,d1) live on this rank → the FULL (P,Rd,Rh) DRR on every rank."""
    part = ops.drr_forward(vol_slab, poses, resolution, spacing, d0=d0, d1=d1, full_D=D, **kw)
    return _all_reduce_sum(part, group)


def backproject_slab(proj, poses, img_shape, d0, d1, **kw):
    """Rows [d0,d1) of the backprojected feature volume; views are replicated → no collective."""
    return ops.backproject(proj, poses, img_shape, d0=d0, d1=d1, **kw)


def pca_reconstruct_slab(coefs, basis_LxM, mean, img_shape, d0, d1):
    """Rows [d0,d1) of the displacement field (B,3,Dn,W,H) from this rank's column slabs of the basis.

    `basis_LxM`/`mean` may be the full arrays (sliced here as three column runs, one per channel) —
    a rank that stores only its slab passes d0=0,d1=Dn with its compact (L,3·Dn·W·H) basis.
    """
    D, W, H = (int(v) for v in img_shape)
    B = coefs.shape[0]
    plane = W * H
    Dn = d1 - d0
    if basis_LxM.shape[1] == 3 * Dn * plane:  # already a compact slab
        return ops.pca_reconstruct(coefs, basis_LxM, mean).view(B, 3, Dn, W, H)
    out = []
    for c in range(3):
        lo, hi = (c * D + d0) * plane, (c * D + d1) * plane
        out.append(ops.pca_reconstruct(coefs, basis_LxM[:, lo:hi], mean[lo:hi].contiguous()).view(B, 1, Dn, W, H))
    return torch.cat(out, dim=1)


def warp_slab(img, disp_slab, id_tables, d0, d1, seg=None, **kw):
    """Output rows [d0,d1): phi and warped slabs; the moving volume `img` is whole on every rank."""
    ids = (id_tables[0][d0:d1].contiguous(), id_tables[1], id_tables[2])
    return ops.warp(img, disp_slab, ids, seg, d0=d0, d1=d1, **kw)


def ncc_loss_sharded(x_slab, y_slab, n_total, group=None, variant=NCC_CONFIGURED):
    """NCC loss of volumes whose slabs live on different ranks: moments add, then the scalar epilogue."""
    n_batch = x_slab.shape[0]
    rows = n_batch if variant == NCC_CONFIGURED else n_batch * x_slab.shape[1]
    m = ops.ncc_moments(x_slab, y_slab, rows)
    m = _all_reduce_sum(m, group)
    loss, _ = ops.ncc_loss_from_moments(m, n_total, n_batch, variant)
    return loss
