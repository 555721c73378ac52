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


# ======================================================================================================
# z-slab sharded forward of the whole model ("a8 convs: 1-voxel halo exchange per layer")
# ======================================================================================================
class DistComm:
    """Communication of ONE local rank over torch.distributed (nccl = RCCL over xGMI, or gloo).

    Under gloo (CPU tests; several ranks sharing one GPU on a 1-GPU box) GPU tensors are staged through the host:
    gloo has no GPU point-to-point or all-gather.  The measured configuration is nccl, one GPU per rank, where the
    tensors go over xGMI as they are."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = world(group)
        self.ranks = [self.rank]
        self.host_stage = self.world > 1 and dist.get_backend(group) == "gloo"

    def _peer(self, group_rank):
        """P2POp takes GLOBAL ranks: translate a rank of `self.group`."""
        return group_rank if self.group is None else dist.get_global_rank(self.group, group_rank)

    def shift_up(self, planes):
        """planes[0] = this rank's top plane → returns [the plane of rank-1] ([None] on rank 0).
        Neighbour point-to-point: on xGMI each pair has its own link, so all exchanges run concurrently.  The send and
        the receive of a rank go out as ONE batch (`batch_isend_irecv` = a grouped RCCL call): un-grouped isend/irecv
        would lazily create one communicator per peer pair and serialise the two operations of an interior rank."""
        if self.world == 1:
            return [None]
        src = planes[0].contiguous()
        send = src.cpu() if (self.host_stage and src.is_cuda) else src
        recv = torch.empty_like(send) if self.rank > 0 else None
        ops_ = []
        if self.rank + 1 < self.world:
            ops_.append(dist.P2POp(dist.isend, send, self._peer(self.rank + 1), group=self.group))
        if self.rank > 0:
            ops_.append(dist.P2POp(dist.irecv, recv, self._peer(self.rank - 1), group=self.group))
        if ops_:
            for r in dist.batch_isend_irecv(ops_):
                r.wait()
        if recv is not None and recv.device != src.device:
            recv = recv.to(src.device)
        return [recv]

    def shift_up_start(self, planes):
        """Post `shift_up` without waiting: returns a handle whose `.wait()` gives [the plane of rank-1].  Over nccl (RCCL)
        the grouped send/recv runs on the communicator's own stream, so kernels launched between start and wait overlap the
        transfer (the sharded forward posts group A's halo, runs block 0 of group B, then waits)."""
        if self.world == 1:
            return _Done([None])
        if self.host_stage:                      # gloo has no GPU point-to-point: staged through the host, synchronously
            return _Done(self.shift_up(planes))
        src = planes[0].contiguous()
        recv = torch.empty_like(src) if self.rank > 0 else None
        ops_ = []
        if self.rank + 1 < self.world:
            ops_.append(dist.P2POp(dist.isend, src, self._peer(self.rank + 1), group=self.group))
        if self.rank > 0:
            ops_.append(dist.P2POp(dist.irecv, recv, self._peer(self.rank - 1), group=self.group))
        reqs = dist.batch_isend_irecv(ops_) if ops_ else []
        return _Pending(reqs, [recv], keep=src)

    def all_gather_cat(self, pieces, dim):
        if self.world == 1:
            return [pieces[0]]
        src = pieces[0].contiguous()
        mine = src.cpu() if (self.host_stage and src.is_cuda) else src
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        return [torch.cat(parts, dim=dim).to(src.device)]

    def all_gather_planes(self, slabs):
        """slabs[0] = this rank's (B, rows, ...) plane slab of a channels-last activation → [the whole (B, world*rows, ...)
        activation].  ONE collective, `all_gather_into_tensor` into a (world, B, rows, ...) buffer — no Python list of
        per-rank tensors, no torch.cat — and one strided copy that brings a sample's planes together."""
        if self.world == 1:
            return [slabs[0]]
        src = slabs[0].contiguous()
        mine = src.cpu() if (self.host_stage and src.is_cuda) else src
        B, rows = mine.shape[0], mine.shape[1]
        buf = torch.empty((self.world * B,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)   # rank-major
        dist.all_gather_into_tensor(buf, mine, group=self.group)
        full = buf.view((self.world, B) + tuple(mine.shape[1:])).transpose(0, 1).reshape((B, self.world * rows) + tuple(mine.shape[2:]))
        return [full.to(src.device)]

    def all_to_all_samples(self, slabs):
        """slabs[0] = this rank's (B, rows, ...) plane slab of a channels-last activation -> [the WHOLE (nb, world*rows, ...) activation
        of the nb = |slab_bounds(B, world, rank)| samples this rank owns].  ONE `all_to_all_single`: a rank sends every peer that
        peer's samples of its planes (1 / world of what `all_gather_planes` moves) and receives its own samples' planes from all."""
        if self.world == 1:
            return [slabs[0]]
        src = slabs[0].contiguous()
        mine = src.cpu() if (self.host_stage and src.is_cuda) else src
        B, rows = mine.shape[0], mine.shape[1]
        cnt = [slab_bounds(B, self.world, q)[1] - slab_bounds(B, self.world, q)[0] for q in range(self.world)]
        nb = cnt[self.rank]
        buf = torch.empty((self.world * nb,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)   # sender-major
        dist.all_to_all_single(buf, mine, output_split_sizes=[nb] * self.world, input_split_sizes=cnt, group=self.group)
        full = buf.view((self.world, nb) + tuple(mine.shape[1:])).transpose(0, 1).reshape((nb, self.world * rows) + tuple(mine.shape[2:]))
        return [full.to(src.device)]

    def all_gather_rows(self, pieces, counts):
        """pieces[0] = this rank's (counts[rank], L) rows -> [the (sum(counts), L) matrix, rank-major] on every rank (rows padded to
        max(counts) on the wire: one small `all_gather_into_tensor`)."""
        if self.world == 1:
            return [pieces[0]]
        src = pieces[0]
        nmax = max(counts)
        pad = torch.zeros((nmax,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        pad[:src.shape[0]].copy_(src)
        mine = pad.cpu() if (self.host_stage and pad.is_cuda) else pad
        buf = torch.empty((self.world * nmax,) + tuple(src.shape[1:]), dtype=src.dtype, device=mine.device)
        dist.all_gather_into_tensor(buf, mine, group=self.group)
        rows = torch.cat([buf[q * nmax:q * nmax + c] for q, c in enumerate(counts)], dim=0)
        return [rows.to(src.device)]

    def all_reduce_sum(self, ts):
        if self.world == 1:
            return ts
        if self.host_stage and ts[0].is_cuda:
            h = ts[0].cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            ts[0].copy_(h)
        else:
            dist.all_reduce(ts[0], op=dist.ReduceOp.SUM, group=self.group)
        return ts

    def all_reduce_sum_fused(self, tensors):
        """Several small same-dtype tensors (NCC moments, partial DRR images …) summed over the ranks in ONE collective:
        packed into one flat buffer, reduced, unpacked in place.  Returns the same list."""
        if self.world == 1 or not tensors:
            return tensors
        flat = torch.cat([t.reshape(-1) for t in tensors])
        self.all_reduce_sum([flat])
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        return tensors


class _Done:
    """A finished exchange (world 1, host-staged gloo, virtual ranks)."""

    def __init__(self, result):
        self.result = result

    def wait(self):
        return self.result


class _Pending:
    """A posted grouped send/recv: wait() blocks the current stream on it and returns the received planes."""

    def __init__(self, reqs, result, keep=None):
        self.reqs, self.result, self.keep = reqs, result, keep     # `keep`: the send buffer stays alive until the wait

    def wait(self):
        for r in self.reqs:
            r.wait()
        self.keep = None
        return self.result


class LocalComm:
    """All `world` ranks simulated in one process (tests: N virtual ranks on one GPU)."""

    def __init__(self, world_size):
        self.world = int(world_size)
        self.ranks = list(range(self.world))

    def shift_up(self, planes):
        return [None] + [p.clone() for p in planes[:-1]]

    def shift_up_start(self, planes):
        return _Done(self.shift_up(planes))

    def all_gather_cat(self, pieces, dim):
        full = torch.cat(pieces, dim=dim)
        return [full for _ in pieces]

    def all_gather_planes(self, slabs):
        full = torch.cat(slabs, dim=1)
        return [full for _ in slabs]

    def all_to_all_samples(self, slabs):
        full = torch.cat(slabs, dim=1)
        B = full.shape[0]
        return [full[slice(*slab_bounds(B, self.world, r))] for r in self.ranks]

    def all_gather_rows(self, pieces, counts):
        rows = torch.cat(pieces, dim=0)
        return [rows for _ in pieces]

    def all_reduce_sum(self, ts):
        total = sum(ts[1:], ts[0].clone())
        return [total.clone() for _ in ts]


class SlabShardedRegistration:
    """One registration batch sharded by z-slab (axis D) over `comm.world` ranks.

    Per rank: rows [d0,d1) of the feature volume, of every encoder activation (halved per stride-2 block),
    of the displacement field, phi and the warped image.  Replicated: the moving volume and the 2-D views
    (small), the FC head.  Exchanged: one activation plane per stride-2 block from block 2 on from the rank below (halo; blocks 0
    and 1 recompute theirs from the replicated inputs),
    the 32·(n/32)³ encoder features (all-gather), five NCC moments per sample (all-reduce).
    Needs D % (m·world) == 0, m = the product of the strides of the blocks that run on slabs (`slab_multiple`: 8 at 256³).
    """

    GATHER_DEPTH = 32      # all-gather the activation behind the first block with at most this many output planes

    def __init__(self, net, comm, sim_variant=NCC_CONFIGURED):
        self.net, self.comm, self.variant = net, comm, sim_variant
        D = net.img_sz[0]
        need = self.slab_multiple(net) * comm.world
        if D % need:
            raise ValueError(f"D={D} must be a multiple of {need} (= {need // comm.world} planes per rank x world {comm.world}) for slab sharding")

    @classmethod
    def last_sharded_block(cls, net):
        """The last encoder block that runs on slabs: the first one whose output has at most GATHER_DEPTH planes (behind it the
        activation is all-gathered and the remaining blocks run replicated), or block 5."""
        depth = [net.img_sz[0]]
        for i in range(6):
            depth.append((depth[-1] - 1) // net.strides[i] + 1)
        return next((i for i in range(1, 6) if depth[i + 1] <= cls.GATHER_DEPTH), 5)

    @classmethod
    def slab_multiple(cls, net):
        """Planes per rank must be a multiple of this: the product of the strides of the blocks that run on slabs (256^3: blocks
        1..3 -> 8; 384^3: blocks 1..4 -> 16; volumes whose last block is still deeper than GATHER_DEPTH: 32)."""
        m = 1
        for i in range(cls.last_sharded_block(net) + 1):
            m *= net.strides[i]
        return m

    def _d_axis(self, layout):
        return 2 if layout == ops.LAYOUT_NCDHW else 1

    def forward(self, inputs):
        """inputs: one dict per LOCAL rank (len(comm.ranks)) with the replicated `source`, `target_proj`,
        `target_poses`, optional whole `target` (sliced here).  Returns one dict per local rank with the
        slabs `warped`, `phi`, `params` (rows d0:d1), replicated `pca_coefs`, the scalar `sim_loss`."""
        net, comm = self.net, self.comm
        D, W, H = net.img_sz
        P = net.drr_feature_num
        bf16 = getattr(net, "conv_dtype", "fp32") == "bf16"
        bounds = [slab_bounds(D, comm.world, r) for r in comm.ranks]

        def layouts(i):
            blk = net.encoders[i]
            return net._bf16_layouts[i] if bf16 else (blk.in_layout, blk.out_layout)

        def conv(i, x, out, d0=0):
            blk = net.encoders[i]
            lin, lout = layouts(i)
            if not bf16:
                # stride-2 blocks see [zero, halo, slab]: local output plane 1 is global plane (d0 of the block's input
                # level) / 2, so local plane 0 has the parity of d0/2 - 1 (the Winograd kernel orders by global parity)
                zp = ((d0 >> (i - 1) >> 1) - 1) & 1 if i >= 1 else 0
                return ops.conv3d_k3_lrelu(x, blk.conv.weight, blk.conv.bias, blk.stride, in_layout=lin, out_layout=lout,
                                           negative_slope=blk._slope, packed=net._packed_weight(i), out=out, z_phase=zp)
            if i == 0:
                return ops.conv3d_first_bf16(x, blk.conv.weight, blk.conv.bias, out_layout=lout, negative_slope=blk._slope,
                                             packed=net._packed_weight(0, bf16=True), out=out)
            return ops.conv3d_k3_lrelu_bf16(x, blk.conv.weight, blk.conv.bias, blk.stride, in_layout=lin, out_layout=lout,
                                            negative_slope=blk._slope, packed=net._packed_weight(i, bf16=True), out=out)

        # Every channels-last activation of a rank lives, per batch element, in a halo-padded run of planes
        # [filler (stride phase: its content never reaches a kept output) | halo plane from the rank below | slab].
        # Blocks 0 and 1 — the two big ones — run as ONE launch per group of samples (round 2: one launch per sample, +7 % at
        # world 1 and the largest part of the decomposition's overhead):
        #   * block 0 (stride 1; both halo planes come from replicated inputs, no communication) writes its rows+2 output
        #     planes [d0-1 | slab | d1] of sample b into planes 1 + b*(rows+2) .. of ONE buffer (strided-batch output for the
        #     edge ranks, whose input has one plane less).  Seen one plane EARLIER, the same memory is the dense batch
        #     (B, rows+2, W, H, C) of block 1's inputs [filler = the unused last plane of the sample before | halo slot |
        #     slab]: no copy, no per-sample launch;
        #   * block 1 (stride 2) writes planes 1.. of the (B, 1 + n_out, ...) buffer the small blocks continue from
        #     (strided-batch output, lr_conv3d_k3_lrelu_obs_*).
        # The batch goes in (up to) two groups so that the halo exchange of group A (posted right after its block 0) travels
        # while block 0 of group B computes: the only wait is in front of block 1 of group A.
        act_dt = torch.bfloat16 if bf16 else torch.float32
        o = lambda n: (n - 1) // 2 + 1
        B = inputs[0]["source"].shape[0]
        ngroups = 2 if (B >= 2 and comm.world > 1 and not getattr(self, "halo_free01", True)) else 1   # (two groups only to hide the exchange)
        gb = [(g * B // ngroups, (g + 1) * B // ngroups) for g in range(ngroups)]
        c0, c1 = net.encoders[0].conv.out_channels, net.encoders[1].conv.out_channels
        # fp32, <= 4 views: blocks 0 and 1 as the ONE fused kernel of the unsharded model (csrc/conv01_fused.hip) on the rank's
        # slab — output planes [d0/2, d1/2) read input planes d0-2 .. d1 of the replicated moving volume (a view) and of the
        # rank's OWN backprojection: no halo exchange for the two big blocks, no 16-channel activation, and the same bits as
        # the unsharded model
        b0_, b1_ = net.encoders[0], net.encoders[1]
        # The decision gates the block-0/1 halo collectives, so it must be the SAME on every rank: it looks at the slabs of ALL
        # ranks of the world (shapes only), never at this process's pointers.
        all_bounds = [slab_bounds(D, comm.world, r) for r in range(comm.world)]
        pair = (not bf16 and getattr(net, "fuse_pair01", False) and P <= getattr(net, "PAIR01_MAX_VIEWS", 2) and b1_.stride == 2 and
                b0_.out_layout == b1_.in_layout and all(d0 % 2 == 0 and d1 % 2 == 0 for d0, d1 in all_bounds) and
                all(ops.conv3d_pair01_shapes_supported(B, P + 1, min(d1 + 1, D) - max(d0 - 2, 0), W, H, b0_.conv.weight,
                                                       b1_.conv.weight, layouts(1)[1]) for d0, d1 in all_bounds))
        if pair:   # what the shape rule cannot see is a property of this process alone: fail loudly, never diverge silently
            for inp, (d0, d1) in zip(inputs, bounds):
                v = inp["source"][:, :, max(d0 - 2, 0):min(d1 + 1, D)]
                if v.data_ptr() % 16 or not v[0].is_contiguous() or (B > 1 and v.stride(0) % 4):
                    raise ValueError("SlabShardedRegistration: `source` must be a contiguous, 16-byte aligned tensor")
        st = []                                     # per local rank: buffers and views
        for inp, (d0, d1) in zip(inputs, bounds):
            moving = inp["source"]
            if net._poses is None:
                p = inp["target_poses"]
                p = p.detach().cpu().numpy() if isinstance(p, torch.Tensor) else p
                net._poses = p[0].astype("float32").copy()
            r = d1 - d0
            n_in = r + 2
            lo, hi = max(d0 - 1, 0), min(d1 + 1, D)
            a0 = lo - (d0 - 1)                      # first output plane block 0 really computes (1 on rank 0)
            n_out = o(n_in)                         # planes block 1 produces from [filler, halo, slab]: 1 junk + r/2
            nb = torch.empty((B, 1 + n_out, o(W), o(H), c1), dtype=act_dt, device=moving.device)
            if pair:                                # (no block-0 output buffer at all)
                st.append(dict(r=r, nb=nb, n_out=n_out))
                continue
            buf0 = torch.empty((1 + B * n_in, W, H, c0), dtype=act_dt, device=moving.device)
            buf0[0].zero_()                         # the leading filler plane only feeds a discarded output plane: keep it finite
            y0 = buf0[1:].view(B, n_in, W, H, c0)   # block 0's outputs [d0-1 | slab | d1] per sample
            in1 = buf0[:B * n_in].view(B, n_in, W, H, c0)   # block 1's inputs [filler | halo | slab] per sample: one plane earlier
            if hi - lo < n_in:                      # edge ranks: the planes block 0 does not write (halo slot / last filler)
                if a0:
                    y0[:, 0].zero_()
                if hi - lo + a0 < n_in:
                    y0[:, n_in - 1].zero_()
            st.append(dict(r=r, n_in=n_in, lo=lo, hi=hi, a0=a0, y0=y0, in1=in1, nb=nb, n_out=n_out))
        if pair:
            for inp, (d0, d1), t in zip(inputs, bounds, st):
                moving, proj = inp["source"], inp["target_proj"]
                lo2, hi2 = max(d0 - 2, 0), min(d1 + 1, D)
                tv = torch.empty((B, P, hi2 - lo2, W, H), dtype=torch.float32, device=moving.device)
                ops.backproject(proj.contiguous(), net._poses, (D, W, H), d0=lo2, d1=hi2, out=tv)
                rows1 = t["r"] // 2
                ops.conv3d_pair01(moving[:, :, lo2:hi2], tv, b0_.conv.weight, b0_.conv.bias, b1_.conv.weight, b1_.conv.bias,
                                  out_layout=layouts(1)[1], slope0=b0_._slope, slope1=b1_._slope, packed=net._packed_pair01(),
                                  out=t["nb"][:, 2:2 + rows1], slab=(D, lo2, d0 // 2, rows1))
                del tv
        def block0(inp, g0, g1, lo, hi, out):
            """Block 0 of samples g0..g1 on input planes [lo, hi) (its own zero padding at both ends) into `out` (g1-g0, hi-lo, W, H, c0):
            the kernels, and bits, of the unsharded model's first block."""
            moving, proj = inp["source"], inp["target_proj"]
            blk = net.encoders[0]
            if (bf16 and P >= getattr(net, "ENCIN_MIN_VIEWS", 99) and c0 == 16 and ops.encoder_input_bf16_supported(moving, proj)):
                # many views (C4): the slab's encoder input as bf16 channels-last records straight from the views and the
                # replicated moving volume
                mvg = moving[g0:g1] if moving[g0:g1].is_contiguous() else moving[g0:g1].contiguous()
                e = ops.backproject_encoder_input_bf16(mvg, proj[g0:g1].contiguous(), net._poses, d0=lo, d1=hi)
                ops.conv3d_first_clin_bf16(e, blk.conv.weight, blk.conv.bias, out_layout=layouts(0)[1], negative_slope=blk._slope,
                                           packed=net._packed_weight(0, bf16=True), out=out)
                return
            mvs = moving[g0:g1, :, lo:hi]                      # a z-slab view of the replicated moving volume
            if not bf16:
                tv = torch.empty((g1 - g0, P, hi - lo, W, H), dtype=torch.float32, device=moving.device)
                if ops.conv3d_first_split_supported(mvs, tv):
                    # fp32, <= 2 views: the first block reads the moving slab in place (no copy, no concatenation)
                    ops.backproject(proj[g0:g1].contiguous(), net._poses, (D, W, H), d0=lo, d1=hi, out=tv)
                    ops.conv3d_first_split(mvs, tv, blk.conv.weight, blk.conv.bias, out_layout=layouts(0)[1], negative_slope=blk._slope,
                                           packed=net._packed_weight(0), out=out)
                    return
                del tv
            x = torch.empty((g1 - g0, P + 1, hi - lo, W, H), dtype=torch.float32, device=moving.device)
            x[:, 0:1].copy_(mvs)
            ops.backproject(proj[g0:g1].contiguous(), net._poses, (D, W, H), d0=lo, d1=hi, out=x[:, 1:],
                            out_batch_stride=(P + 1) * (hi - lo) * W * H)
            conv(0, x, out)

        # Blocks 0 and 1 exchange NOTHING: block 1's halo plane (block 0's output plane d0 - 1) is recomputed here from input
        # planes d0 - 2 .. d0 of the replicated moving volume and the rank's own backprojection — three planes of block 0
        # instead of one point-to-point transfer per group (round 4 did this for the fused fp32 pair kernel only;
        # `halo_free01 = False` keeps the exchange of the rank below's top plane: A/B aid).
        halo_free = bool(getattr(self, "halo_free01", True))
        pend = []
        for g0, g1 in (() if pair else gb):         # ---- block 0 of every group (exchange form: its halo posted at once)
            tops = []
            for inp, (d0, d1), t in zip(inputs, bounds, st):
                lo, hi, n_real = t["lo"], t["hi"], t["hi"] - t["lo"]
                block0(inp, g0, g1, lo, hi, t["y0"][g0:g1, t["a0"]:t["a0"] + n_real])
                if halo_free:
                    if d0 > 0:
                        h3 = torch.empty((g1 - g0, 3, W, H, c0), dtype=act_dt, device=t["y0"].device)
                        block0(inp, g0, g1, d0 - 2, d0 + 1, h3)
                        tops.append(h3[:, 1:2])
                    else:
                        tops.append(None)
                else:
                    tops.append(t["y0"][g0:g1, t["r"]:t["r"] + 1])       # the slab's top plane (d1 - 1) -> the rank above
            pend.append(_Done(tops) if halo_free else comm.shift_up_start(tops))
        for (g0, g1), h in zip(() if pair else gb, pend):   # ---- block 1 of every group, behind its halo plane
            halos = h.wait()
            for (d0, _), t, hl in zip(bounds, st, halos):
                if hl is None:
                    t["y0"][g0:g1, 0].zero_()       # rank 0: the conv's own zero padding
                else:
                    t["y0"][g0:g1, 0].copy_(hl[:, 0])
                conv(1, t["in1"][g0:g1], t["nb"][g0:g1, 1:1 + t["n_out"]], d0)
        acts_p = [t["nb"] for t in st]
        rows = [t["r"] // 2 for t in st]
        # ---- blocks 2..: small activations, batched launches; the (small) output is copied behind the two leading planes.
        # "gather once spatial <= 32^3" — behind the first block whose output has at most GATHER_DEPTH planes the
        # slabs are all-gathered (ONE collective) and the remaining blocks + the FC head run replicated on the whole (tiny)
        # activation: at 256^3 that is block 3 (32^3, 4 MB per sample), and the halo rounds of blocks 4 and 5 are gone.
        last_sharded = self.last_sharded_block(net)
        for i in range(2, last_sharded + 1):
            tops = [a[:, 1 + r:2 + r].contiguous() for a, r in zip(acts_p, rows)]
            halos = comm.shift_up(tops)
            nxt, nrows = [], []
            for a, r, h, (d0, _) in zip(acts_p, rows, halos, bounds):
                a[:, 0].zero_()
                if h is None:
                    a[:, 1].zero_()
                else:
                    a[:, 1].copy_(h[:, 0])
                if i == 5:
                    y = conv(i, a, None, d0)      # (B, 32, 1 + r/2, ...): the last block writes NCDHW
                    nxt.append(y[:, :, 1:].contiguous())
                else:
                    # (B, 1 + r/2, ...) channels-last, written straight behind the next block's filler plane (strided-batch output:
                    # no copy of the block's output)
                    cout = net.encoders[i].conv.out_channels
                    nb = torch.empty((a.shape[0], 2 + (a.shape[1] - 1) // 2, o(a.shape[2]), o(a.shape[3]), cout), dtype=act_dt, device=a.device)
                    conv(i, a, nb[:, 1:], d0)
                    nxt.append(nb)
                nrows.append(r // 2)
            acts_p, rows = nxt, nrows
        if last_sharded == 5:
            # (volumes whose last block still has more than GATHER_DEPTH planes) the features themselves are gathered
            feats = [f.permute(0, 2, 1, 3, 4) for f in comm.all_gather_planes([a.permute(0, 2, 1, 3, 4) for a in acts_p])]
        coefs_all = None
        Bfull = inputs[0]["source"].shape[0]
        if last_sharded < 5 and getattr(self, "sample_sharded_tail", True) and hasattr(comm, "all_to_all_samples"):
            # ---- the small tail (blocks behind the gather depth + the FC head) SHARDED BY SAMPLE: one all-to-all hands every rank the
            # whole (<= GATHER_DEPTH planes deep) activation of ITS B / world samples — 1 / world of the bytes an all-gather moves —,
            # the rank runs the remaining blocks and the three Linear layers on them (the kernels, and bits, of the unsharded model:
            # none of them reduces over the batch), and the (B, L) coefficients meet in one small all-gather.  Replicated (round 5)
            # every rank ran these latency-bound launches on the WHOLE batch: 8 x the work at 8 ranks for the same wall time.
            mine = comm.all_to_all_samples([a[:, 2:2 + r] for a, r in zip(acts_p, rows)])
            counts = [slab_bounds(Bfull, comm.world, q)[1] - slab_bounds(Bfull, comm.world, q)[0] for q in range(comm.world)]
            pieces = []
            for x in mine:
                if x.shape[0] == 0:
                    pieces.append(torch.empty((0, net.latent_dim), dtype=torch.float32, device=x.device))
                    continue
                for j in range(last_sharded + 1, 6):
                    blk = net.encoders[j]
                    if bf16:
                        lin, lout = layouts(j)
                        x = ops.conv3d_k3_lrelu_bf16(x, blk.conv.weight, blk.conv.bias, blk.stride, in_layout=lin, out_layout=lout,
                                                     negative_slope=blk._slope, packed=net._packed_weight(j, bf16=True))
                    else:
                        x = blk(x, packed=net._packed_weight(j))
                pieces.append(net.encoders[6](x.contiguous()))
            coefs_all = comm.all_gather_rows(pieces, counts)
        elif last_sharded < 5:
            whole = comm.all_gather_planes([a[:, 2:2 + r] for a, r in zip(acts_p, rows)])   # real planes sit behind [filler | halo]
            feats = []
            for x in whole:
                for j in range(last_sharded + 1, 6):
                    blk = net.encoders[j]
                    if bf16:
                        lin, lout = layouts(j)
                        x = ops.conv3d_k3_lrelu_bf16(x, blk.conv.weight, blk.conv.bias, blk.stride, in_layout=lin, out_layout=lout,
                                                     negative_slope=blk._slope, packed=net._packed_weight(j, bf16=True))
                    else:
                        x = blk(x, packed=net._packed_weight(j))
                feats.append(x)                      # last block: NCDHW, so nn.Flatten sees the reference's element order
        # ---- FC head, then the slab-local decode.  The first layer (Linear(32 (n/32)^3, 800): 52 MB of weights at 256^3, …Backproj.py:34-36)
        # is sharded by OUTPUT neurons — a rank reads 1/world of the weight and computes its neurons' dot products whole (the bits
        # of the replicated layer: lr_linear_lrelu_f32 reduces every neuron on its own) — and the (B, 800/world) pieces meet in one
        # small all-gather; layers 2 and 3 (0.9 MB) run replicated.
        head = net.encoders[6]
        if coefs_all is None:      # (the replicated tail: `sample_sharded_tail = False`, or volumes still deeper than GATHER_DEPTH at block 5)
            fc1 = head[1] if (len(head) == 4 and hasattr(head[1], "fc")) else None     # (another head shape: the replicated fallback)
            O1 = fc1.fc.out_features if fc1 is not None else 0
            if comm.world > 1 and fc1 is not None and O1 % comm.world == 0:
                per = O1 // comm.world
                pieces = [ops.linear_lrelu(f.contiguous().flatten(1), fc1.fc.weight[r * per:(r + 1) * per], fc1.fc.bias[r * per:(r + 1) * per],
                                           fc1._slope) for r, f in zip(comm.ranks, feats)]
                coefs_all = [head[3](head[2](h)) for h in comm.all_gather_cat(pieces, dim=1)]
            else:
                coefs_all = [head(f.contiguous()) for f in feats]
        outs = []
        moms = []
        for inp, coefs, (d0, d1) in zip(inputs, coefs_all, bounds):
            moving = inp["source"]
            tgt = None
            if "target" in inp:
                tgt = inp["target"][:, :, d0:d1].contiguous()
                if "source_label" in inp and "target_label" in inp:   # (target+1)*target_seg-1 (…Backproj.py:57-58)
                    tgt = ops.mask_compose(tgt, inp["target_label"][:, :, d0:d1].contiguous())
            # the rank's COMPACT column slab of the basis (L, 3·Dn·W·H): 11.3 GB → 1.4 GB per GPU at 8 ranks (C3)
            basis_s, mean_s = net.pca_slab(d0, d1, moving.device)
            ids = (net._id0[d0:d1].contiguous(), net._id1, net._id2)
            mom = None
            if "source_label" not in inp and ops.pca_warp_supported(coefs, basis_s, moving, d0, d1):
                # ONE launch: PCA reconstruction of the slab + identity + warp (+ the NCC moments of the slab)
                if tgt is not None and moving.shape[1] == 1 and getattr(net, "fuse_ncc", False):
                    disp, phi, warped, mom = ops.pca_warp(coefs, basis_s, mean_s, ids, moving, d0=d0, d1=d1, target=tgt)
                else:
                    disp, phi, warped = ops.pca_warp(coefs, basis_s, mean_s, ids, moving, d0=d0, d1=d1)
            else:
                disp = pca_reconstruct_slab(coefs, basis_s, mean_s, (D, W, H), d0, d1)
                phi, warped = ops.warp(moving, disp.contiguous(), ids, inp.get("source_label"), d0=d0, d1=d1)
            out = {"warped": warped, "phi": phi, "params": disp, "pca_coefs": coefs}
            if tgt is not None:
                out["target"] = tgt
                rows = warped.shape[0] if self.variant == NCC_CONFIGURED else warped.shape[0] * warped.shape[1]
                moms.append(mom if (mom is not None and mom.shape[0] == rows) else ops.ncc_moments(warped, tgt, rows))
            outs.append(out)
        if moms:
            moms = comm.all_reduce_sum_fused(moms) if hasattr(comm, "all_reduce_sum_fused") else comm.all_reduce_sum(moms)
            for out, m, inp in zip(outs, moms, inputs):
                B = inp["source"].shape[0]
                out["sim_loss"], _ = ops.ncc_loss_from_moments(m, D * W * H * (inp["source"].shape[1]), B, self.variant)
        return outs


class GradientAllReduce:
    """Data-parallel training ("only training needs a collective: all-reduce of gradients").

    The parameters' `.grad` tensors are views into a few flat fp32 buckets, so a step needs one collective per
    bucket (≈54 MB in all at 256³: fewer, larger messages — what point-to-point xGMI rings want) and no
    flatten/unflatten copies.  Buckets follow the order gradients appear in backward: the FC head
    (`head_prefix`, 52 of the 54 MB) is complete before the conv backward even starts, so its all-reduce is
    launched from an autograd hook and runs on RCCL's stream underneath the ≈40 ms of conv backward kernels;
    the small encoder bucket goes out when backward ends.

        ddp = GradientAllReduce(net)            # after net.to(device)
        for batch in loader:
            ddp.zero_grad()                     # one memset per bucket (do not use set_to_none=True)
            loss(net(batch)).backward()
            ddp.finish()                        # wait + average
            optimizer.step()
    """

    def __init__(self, module, group=None, head_prefix="encoders.6."):
        self.group = group
        self.rank, self.world = world(group)
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        split = [[p for n, p in named if n.startswith(head_prefix)], [p for n, p in named if not n.startswith(head_prefix)]]
        self.buckets = []
        for params in split:
            if not params:
                continue
            flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
            o = 0
            for p in params:
                p.grad = flat[o:o + p.numel()].view_as(p)
                o += p.numel()
            self.buckets.append({"flat": flat, "params": params, "pending": len(params), "work": None})
        self._hooks = []
        for bk in self.buckets:
            for p in bk["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bk)))

    def _make_hook(self, bk):
        def hook(_param):
            bk["pending"] -= 1
            if bk["pending"] == 0:
                self._launch(bk)
        return hook

    def _launch(self, bk):
        if self.world > 1 and bk["work"] is None:
            bk["work"] = dist.all_reduce(bk["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def zero_grad(self):
        for bk in self.buckets:
            bk["flat"].zero_()
            bk["pending"] = len(bk["params"])
            bk["work"] = None

    def finish(self):
        """Wait for the collectives (launching any whose hook did not fire, e.g. unused parameters) and average."""
        for bk in self.buckets:
            self._launch(bk)
            if bk["work"] is not None:
                bk["work"].wait()
                bk["flat"].mul_(1.0 / self.world)
            for p in bk["params"]:          # autograd must have accumulated in place into the bucket views
                if p.grad is None or p.grad.data_ptr() < bk["flat"].data_ptr() or \
                        p.grad.data_ptr() >= bk["flat"].data_ptr() + bk["flat"].numel() * bk["flat"].element_size():
                    raise RuntimeError("a parameter's .grad left its bucket (zero_grad(set_to_none=True)?)")

    def nbytes(self):
        return sum(bk["flat"].numel() * bk["flat"].element_size() for bk in self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
