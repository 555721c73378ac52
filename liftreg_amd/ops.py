"""Tensor-level front-end of the HIP kernels (liftreg_amd/csrc → libliftreg_hip.so).

PyTorch is plumbing here: device memory, the current HIP stream and (elsewhere)
torch.distributed.  Every function validates its tensors, allocates outputs with
torch.empty and launches through the C ABI on torch's current stream.  There is no
CPU path: a CPU tensor or a missing library raises.
"""
import contextlib
import os

import numpy as np
import torch

from . import _hip

LAYOUT_NCDHW, LAYOUT_NDHWC, LAYOUT_NDHWC_HPS = _hip.LAYOUT_NCDHW, _hip.LAYOUT_NDHWC, _hip.LAYOUT_NDHWC_HPS
LAYOUT_BF16_NDHWC, LAYOUT_BF16_NDHWC_HPS = _hip.LAYOUT_BF16_NDHWC, _hip.LAYOUT_BF16_NDHWC_HPS
LAYOUT_NCDHW_RBF16 = _hip.LAYOUT_NCDHW_RBF16
MFMA_BF16_PEAK_TF = 2516.6   # dense bf16 MFMA peak of MI355X (TFLOP/s) — roofline denominator of the bf16 blocks


# ----------------------------------------------------------------------------- helpers
def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise _hip.LiftRegHipError(f"{name}: tensor is on {t.device}; liftreg_amd ops run on the GPU only "
                                   "(no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _out_arg(out, shape, name="out", dtype=torch.float32):
    """A caller-supplied output buffer: it is written in place, so it is never copied — a non-contiguous, wrongly
    shaped or wrongly typed `out` raises instead of silently writing into a temporary."""
    if not isinstance(out, torch.Tensor) or not out.is_cuda:
        raise _hip.LiftRegHipError(f"{name}: must be a GPU tensor (no CPU fallback)")
    if out.dtype != dtype or tuple(out.shape) != tuple(shape) or not out.is_contiguous():
        raise ValueError(f"{name}: must be a contiguous {dtype} tensor of shape {tuple(shape)}, got "
                         f"{out.dtype} {tuple(out.shape)} (contiguous={out.is_contiguous()})")
    return out


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    if _hip.AUTOSYNC:            # tests / A-B tools only: the library reads its LIFTREG_* switches once per process
        _hip.sync_switches()
    return torch.cuda.current_stream().cuda_stream


def _host_f32(a, shape_tail, name):
    """Small host-side parameter (poses, spacing) as a contiguous float32 numpy array."""
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(np.asarray(a), dtype=np.float32)
    if a.shape[-len(shape_tail):] != tuple(shape_tail):
        raise ValueError(f"{name}: expected trailing shape {shape_tail}, got {a.shape}")
    return a


class KernelTimer:
    """Records a HIP event pair (on the launch stream) around every op while active — or, with `only`, around the launches of
    that one op (an event pair costs ~4 us of stream time: 1 % of a 10 ms step when all twelve kernels carry one)."""

    def __init__(self, only=None):
        self.records = []  # (name, start, end, info)
        self.only = only

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, s, e, info in self.records:
            d = out.setdefault(name, {"ms": [], "info": info})
            d["ms"].append(s.elapsed_time(e))
        return out


_timer = None


@contextlib.contextmanager
def kernel_timer(only=None):
    """`with kernel_timer() as t:` … t.summary() → {op: {"ms": [per-launch], "info": …}}; only="op name": that op alone."""
    global _timer
    prev, _timer = _timer, KernelTimer(only)
    try:
        yield _timer
    finally:
        _timer = prev


@contextlib.contextmanager
def _timed(name, **info):
    if _timer is None or (_timer.only is not None and name != _timer.only):
        yield
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    yield
    e.record()
    _timer.records.append((name, s, e, info))


# ----------------------------------------------------------------------------- K1 DRR
def hu_to_mu(vol):
    """calc_relative_atten_coef (reference sdct_projection_utils.py:6-9) on the GPU: HU → attenuation."""
    vol = _dev(vol, "vol")
    mu = torch.empty_like(vol)
    with _timed("hu_to_mu", bytes=8 * vol.numel()):
        _hip.check(_hip.lib().lr_hu_to_mu_f32(vol.data_ptr(), mu.data_ptr(), vol.numel(), _stream()), "lr_hu_to_mu_f32")
    return mu


def drr_forward(vol, poses, resolution, spacing=(2.2, 2.2, 2.2), *, d0=0, d1=None, full_D=None,
                hu_input=False, flip_w=False, nseg=0, out=None, fold_hu=None):
    """Cone-beam DRR of `vol` (Ds,W,H) = rows [d0,d1) of a (D,W,H) volume → (P,Rd,Rh).

    Replaces calculate_projection (reference sdct_projection_utils.py:59-100).  `hu_input`: the volume is in HU;
    calc_relative_atten_coef (:6-9) is folded into the projector's tap loads (the division by 1000 as a multiplication +
    one exact correction step, lr_drr_forward_f32 with LR_DRR_HU_INPUT: same bits, no temporary volume, no extra pass);
    `fold_hu=False` converts once per voxel with `hu_to_mu` first — same bits; `fold_hu=None` (default) picks by `_fold_hu_pays`.
    """
    vol = _dev(vol, "vol")
    if vol.dim() != 3:
        raise ValueError("vol must be (D,W,H)")
    poses = _host_f32(poses, (3,), "poses").reshape(-1, 3)
    sp = _host_f32(spacing, (3,), "spacing")
    P = poses.shape[0]
    Rd, Rh = int(resolution[0]), int(resolution[1])
    if hu_input and (fold_hu is False or (fold_hu is None and not _fold_hu_pays(P, (Rd, Rh), vol.shape[0], vol.shape[2]))):
        vol, hu_input = hu_to_mu(vol), False       # one conversion per voxel (same bits as the folded form)
    Ds, W, H = vol.shape
    D = Ds if full_D is None else int(full_D)
    d1 = D if d1 is None else int(d1)
    if d1 - d0 != Ds:
        raise ValueError(f"slab rows [{d0},{d1}) do not match vol.shape[0]={Ds}")
    if out is None:
        out = torch.empty((P, Rd, Rh), dtype=torch.float32, device=vol.device)
    else:
        out = _out_arg(out, (P, Rd, Rh))
    flags = (_hip.DRR_HU_INPUT if hu_input else 0) | (_hip.DRR_FLIP_W if flip_w else 0)
    with _timed("drr_forward", bytes=4 * (Ds * W * H + P * Rd * Rh)):
        _hip.check(_hip.lib().lr_drr_forward_f32(vol.data_ptr(), poses.ctypes.data, sp.ctypes.data,
                                                 out.data_ptr(), D, W, H, d0, d1, P, Rd, Rh, flags, nseg,
                                                 _stream()), "lr_drr_forward_f32")
    return out


def _fold_hu_pays(P, resolution, Dn, H):
    """HU input: convert inside the projector's tap loads (8 conversions per SAMPLE, no temporary volume) or once per VOXEL in a
    pass of its own (lr_hu_to_mu_f32: read + write of the volume)?  Measured (tools/ab_drr.py, round 6): C3 (2 views of 256^2 on
    256^3: 2 samples per voxel) folded 0.104-0.108 vs 0.116 ms per volume; the reference's shipped shape (4 views of 240^2 on 160^3:
    9 samples per voxel) folded 0.117 vs 0.105 — the pass pays above ~4 samples per voxel.  Same bits either way."""
    return int(P) * int(resolution[0]) * int(resolution[1]) <= 4 * int(Dn) * int(H)


def drr_forward_batch(vols, poses, resolution, spacing=(2.2, 2.2, 2.2), *, hu_input=False, flip_w=False, nseg=0, out=None, fold_hu=None):
    """Cone-beam DRRs of B volumes `vols` (B,D,W,H) (or (B,1,D,W,H)) with ONE geometry in one launch → (B,P,Rd,Rh): the bits of
    B calls of `drr_forward` (reference: tools/preprocessingDRR.py:123-154 projects every case's source and target with the same
    emitter poses, sdct_projection_utils.py:59-100 each)."""
    vols = _dev(vols, "vols")
    if vols.dim() == 5 and vols.shape[1] == 1:
        vols = vols[:, 0]
    if vols.dim() != 4 or not vols[0].is_contiguous():
        raise ValueError("vols must be (B,D,W,H) with dense (D,W,H) volumes")
    B, D, W, H = vols.shape
    poses = _host_f32(poses, (3,), "poses").reshape(-1, 3)
    sp = _host_f32(spacing, (3,), "spacing")
    P = poses.shape[0]
    Rd, Rh = int(resolution[0]), int(resolution[1])
    if hu_input and (fold_hu is False or (fold_hu is None and not _fold_hu_pays(P, (Rd, Rh), D, H))):
        vols, hu_input = hu_to_mu(vols.contiguous()), False     # one conversion per voxel (same bits as the folded form)
    out = torch.empty((B, P, Rd, Rh), dtype=torch.float32, device=vols.device) if out is None else _out_arg(out, (B, P, Rd, Rh))
    flags = (_hip.DRR_HU_INPUT if hu_input else 0) | (_hip.DRR_FLIP_W if flip_w else 0)
    with _timed("drr_forward_batch", bytes=4 * B * (D * W * H + P * Rd * Rh), samples=B):
        _hip.check(_hip.lib().lr_drr_forward_batch_f32(vols.data_ptr(), int(vols.stride(0)) if B > 1 else D * W * H,
                                                       poses.ctypes.data, sp.ctypes.data, out.data_ptr(), B, D, W, H, 0, D, P,
                                                       Rd, Rh, flags, nseg, _stream()), "lr_drr_forward_batch_f32")
    return out


def drr_sample_coords(poses, spacing, shape, resolution, device, normalized=False):
    """(pix (P,Rd,Rh,W,3) ordered (d,w,h), dx (P,Rd,Rh)) — the projector's sample grid."""
    D, W, H = (int(v) for v in shape)
    Rd, Rh = int(resolution[0]), int(resolution[1])
    poses = _host_f32(poses, (3,), "poses").reshape(-1, 3)
    sp = _host_f32(spacing, (3,), "spacing")
    P = poses.shape[0]
    pix = torch.empty((P, Rd, Rh, W, 3), dtype=torch.float32, device=device)
    dx = torch.empty((P, Rd, Rh), dtype=torch.float32, device=device)
    _dev(pix, "pix")
    _hip.check(_hip.lib().lr_drr_sample_coords_f32(poses.ctypes.data, sp.ctypes.data, pix.data_ptr(),
                                                   dx.data_ptr(), D, W, H, P, Rd, Rh, int(normalized),
                                                   _stream()), "lr_drr_sample_coords_f32")
    return pix, dx


# ----------------------------------------------------------------------------- K2 backprojection
def backproject(proj, poses, img_shape, *, d0=0, d1=None, out=None, out_batch_stride=None):
    """(B,P,Pw,Ph) views → (B,P,Ds,W,H) feature volume for ONE emitter geometry `poses` (P,3).

    Replaces backproj_grids_with_poses + F.grid_sample (reference …Backproj.py:85-93).
    `out` may be a view into a larger buffer whose batch stride is `out_batch_stride`
    elements (writing straight into channels 1..P of the encoder input).
    """
    proj = _dev(proj, "proj")
    B, P, Pw, Ph = proj.shape
    D, W, H = (int(v) for v in img_shape)
    d1 = D if d1 is None else int(d1)
    Ds = d1 - d0
    poses = _host_f32(poses, (3,), "poses").reshape(-1, 3)
    if poses.shape[0] != P:
        raise ValueError(f"poses has {poses.shape[0]} views, proj has {P}")
    if out is None:
        out = torch.empty((B, P, Ds, W, H), dtype=torch.float32, device=proj.device)
        out_batch_stride = P * Ds * W * H
        optr = out.data_ptr()
    else:
        if out_batch_stride is None:
            out = _out_arg(out, (B, P, Ds, W, H))
            out_batch_stride = P * Ds * W * H
        else:
            # a view into a larger buffer (channels 1..P of the encoder input): each batch element's (P,Ds,W,H)
            # block must itself be dense, the batch stride is the caller's
            if not out.is_cuda or out.dtype != torch.float32:
                raise TypeError("out must be a float32 GPU tensor")
            if tuple(out.shape) != (B, P, Ds, W, H) or (B > 1 and out.stride(0) != int(out_batch_stride)) or \
                    tuple(out.stride()[1:]) != (Ds * W * H, W * H, H, 1):
                raise ValueError(f"out must be a {(B, P, Ds, W, H)} view with dense (P,Ds,W,H) blocks and batch stride "
                                 f"{out_batch_stride}, got shape {tuple(out.shape)} strides {tuple(out.stride())}")
        optr = out.data_ptr()
    with _timed("backproject", bytes=4 * (B * P * Ds * W * H + B * P * Pw * Ph), samples=B):
        _hip.check(_hip.lib().lr_backproject_f32(proj.data_ptr(), poses.ctypes.data, optr, B, P, Pw, Ph, D, W, H, d0, d1,
                                                 int(out_batch_stride), _stream()), "lr_backproject_f32")
    return out


def backproject_coords(poses, img_shape, proj_shape, device, normalized=False):
    """(P,D,W,H,2) detector coordinates (Pw axis, Ph axis) of every voxel's shadow."""
    D, W, H = (int(v) for v in img_shape)
    Pw, Ph = int(proj_shape[0]), int(proj_shape[1])
    poses = _host_f32(poses, (3,), "poses").reshape(-1, 3)
    P = poses.shape[0]
    pix = torch.empty((P, D, W, H, 2), dtype=torch.float32, device=device)
    _dev(pix, "pix")
    _hip.check(_hip.lib().lr_backproject_coords_f32(poses.ctypes.data, pix.data_ptr(), P, Pw, Ph, D, W, H,
                                                    int(normalized), _stream()), "lr_backproject_coords_f32")
    return pix


def backproject_coords_poseless_f64(poses64, img_shape, proj_shape, device):
    """(P,2,D,W,H) float64 grid of the pose-less backproj_grids (reference sdct_projection_utils.py:179-202)."""
    D, W, H = (int(v) for v in img_shape)
    Pw, Ph = int(proj_shape[0]), int(proj_shape[1])
    poses = np.ascontiguousarray(np.asarray(poses64, dtype=np.float64).reshape(-1, 3))
    P = poses.shape[0]
    grid = torch.empty((P, 2, D, W, H), dtype=torch.float64, device=device)
    if not grid.is_cuda:
        raise _hip.LiftRegHipError("backproject_coords_poseless_f64 runs on the GPU only (no CPU fallback)")
    _hip.check(_hip.lib().lr_backproject_coords_poseless_f64(poses.ctypes.data, grid.data_ptr(), P, Pw, Ph, D, W, H,
                                                             _stream()), "lr_backproject_coords_poseless_f64")
    return grid


# ----------------------------------------------------------------------------- K3 conv
def conv3d_pack_weights(weight, in_layout):
    """(Cout,Cin,3,3,3) → MFMA B-operand order for lr_conv3d_k3_lrelu_f32."""
    weight = _dev(weight.detach(), "weight")
    Cout, Cin = weight.shape[0], weight.shape[1]
    if tuple(weight.shape[2:]) != (3, 3, 3):
        raise ValueError("kernel must be 3x3x3")
    n = _hip.lib().lr_conv3d_packed_floats(Cin, Cout, in_layout)
    if n < 0:
        _hip.check(int(n), "lr_conv3d_packed_floats")
    packed = torch.empty((n,), dtype=torch.float32, device=weight.device)
    _hip.check(_hip.lib().lr_conv3d_pack_weights_f32(weight.data_ptr(), packed.data_ptr(), Cin, Cout, in_layout,
                                                     _stream()), "lr_conv3d_pack_weights_f32")
    return packed


def _conv_out(out, shape, dtype, device, strided_batch=False):
    """The output tensor of a conv wrapper: freshly allocated, or the caller's `out` — contiguous, or (strided_batch) a
    batch of dense per-sample blocks whose batch stride is larger than a block (a plane range of per-sample
    halo-padded buffers, see parallel.SlabShardedRegistration: ONE launch for the whole batch)."""
    if out is None:
        return torch.empty(shape, dtype=dtype, device=device)
    if tuple(out.shape) != tuple(shape) or out.dtype != dtype or not out.is_cuda:
        raise ValueError(f"out must be a {dtype} GPU tensor of shape {tuple(shape)}")
    if not out.is_contiguous():
        dense = int(np.prod(shape[1:]))
        if not (strided_batch and out[0].is_contiguous() and (out.shape[0] == 1 or out.stride(0) >= dense)):
            raise ValueError(f"out must be contiguous (or, for the stride-2 blocks, dense per batch element with a larger "
                             f"batch stride); got strides {tuple(out.stride())}")
    return out


def _batch_stride(y):
    """Elements between two batch elements of a conv output (dense blocks; see _conv_out)."""
    return int(y.stride(0)) if (y.shape[0] > 1 and not y.is_contiguous()) else 0


def conv3d_mask_supported(x, weight, stride, in_layout, out_layout):
    """True when conv3d_k3_lrelu can also emit the LeakyReLU sign mask (LAYOUT_SIGN4; the encoder's first block in training)."""
    return (in_layout == LAYOUT_NCDHW and out_layout in (LAYOUT_NDHWC, LAYOUT_NDHWC_HPS) and stride == 1 and
            weight.shape[0] == 16 and weight.shape[1] <= 3 and x.dim() == 5 and x.shape[4] % 4 == 0 and x.data_ptr() % 16 == 0)


def conv3d_k3_lrelu(x, weight, bias, stride, *, in_layout=LAYOUT_NCDHW, out_layout=LAYOUT_NCDHW,
                    negative_slope=0.2, packed=None, out=None, mask_out=None, z_phase=0):
    """LeakyReLU(Conv3d(k3,p1,stride)(x)+b).  x is (B,Cin,D,W,H) for NCDHW, (B,D,W,H,Cin) for NDHWC.
    LAYOUT_NDHWC_HPS is NDHWC with every H row parity-split (even voxels, then odd): the private layout
    between a block and a following stride-2 block (see `hps_to_ndhwc`).
    z_phase (0|1): x is a z-slab of a larger volume and local output plane 0 is an odd (1) / even (0) GLOBAL output
    plane — the stride-2 Winograd kernel then reproduces the unsharded launch bit for bit (lr_conv3d_k3_lrelu_zphase_f32).

    Replaces convBlock (reference layers/layers.py:335-372).
    """
    x = _dev(x, "x")
    if x.dim() != 5:
        raise ValueError("x must be 5-D")
    if in_layout == LAYOUT_NCDHW:
        B, Cin, D, W, H = x.shape
    else:
        B, D, W, H, Cin = x.shape
    Cout = weight.shape[0]
    if weight.shape[1] != Cin:
        raise ValueError(f"weight expects Cin={weight.shape[1]}, input has {Cin}")
    if packed is None:
        packed = conv3d_pack_weights(weight, in_layout)
    b = None if bias is None else _dev(bias.detach(), "bias")
    o = lambda n: (n - 1) // stride + 1
    Do, Wo, Ho = o(D), o(W), o(H)
    shape = (B, Cout, Do, Wo, Ho) if out_layout == LAYOUT_NCDHW else (B, Do, Wo, Ho, Cout)
    bf16_out = out_layout in (_hip.LAYOUT_BF16_NDHWC, _hip.LAYOUT_BF16_NDHWC_HPS)   # fp32 compute, bf16 storage
    y = _conv_out(out, shape, torch.bfloat16 if bf16_out else torch.float32, x.device, strided_batch=mask_out is None)
    obs = _batch_stride(y)
    flops = 2.0 * 27 * Cin * Cout * B * Do * Wo * Ho
    with _timed(f"conv3d_c{Cin}x{Cout}_s{stride}_{D}" + ("_bf16out" if bf16_out else ""), flops=flops,
                bytes=4 * x.numel() + y.numel() * y.element_size(), samples=B):
        if mask_out is not None:
            # training forward of the first block: also the (B,D,W,H,C/4) uint8 sign mask for the next block's data gradient
            if not mask_out.is_cuda or mask_out.dtype != torch.uint8 or tuple(mask_out.shape) != (B, Do, Wo, Ho, Cout // 4) or \
                    not mask_out.is_contiguous():
                raise ValueError(f"mask_out must be a contiguous uint8 GPU tensor of shape {(B, Do, Wo, Ho, Cout // 4)}")
            _hip.check(_hip.lib().lr_conv3d_k3_lrelu_mask_f32(x.data_ptr(), packed.data_ptr(), _ptr(b), y.data_ptr(),
                                                              mask_out.data_ptr(), B, Cin, Cout, D, W, H, stride, in_layout,
                                                              out_layout, float(negative_slope), _stream()),
                       "lr_conv3d_k3_lrelu_mask_f32")
        elif obs:
            _hip.check(_hip.lib().lr_conv3d_k3_lrelu_obs_f32(x.data_ptr(), packed.data_ptr(), _ptr(b), y.data_ptr(), B,
                                                             Cin, Cout, D, W, H, stride, in_layout, out_layout,
                                                             float(negative_slope), int(z_phase), obs, _stream()),
                       "lr_conv3d_k3_lrelu_obs_f32")
        elif z_phase:
            _hip.check(_hip.lib().lr_conv3d_k3_lrelu_zphase_f32(x.data_ptr(), packed.data_ptr(), _ptr(b), y.data_ptr(), B,
                                                                Cin, Cout, D, W, H, stride, in_layout, out_layout,
                                                                float(negative_slope), int(z_phase), _stream()),
                       "lr_conv3d_k3_lrelu_zphase_f32")
        else:
            _hip.check(_hip.lib().lr_conv3d_k3_lrelu_f32(x.data_ptr(), packed.data_ptr(), _ptr(b), y.data_ptr(), B,
                                                         Cin, Cout, D, W, H, stride, in_layout, out_layout,
                                                         float(negative_slope), _stream()), "lr_conv3d_k3_lrelu_f32")
    return y


def conv3d_first_split_supported(x0, rest):
    """True when `conv3d_first_split` can take these tensors (else: concatenate and call conv3d_k3_lrelu).  x0 may be a
    z-slab view of a larger volume: dense per batch element, any batch stride."""
    return (x0.dim() == 5 and rest.dim() == 5 and x0.shape[1] == 1 and rest.shape[1] in (1, 2) and x0.shape[4] % 4 == 0 and
            x0.shape[0] == rest.shape[0] and x0.shape[2:] == rest.shape[2:] and x0[0].is_contiguous() and rest.is_contiguous() and
            (x0.shape[0] == 1 or x0.stride(0) % 4 == 0) and x0.data_ptr() % 16 == 0 and rest.data_ptr() % 16 == 0)


def conv3d_first_split(x0, rest, weight, bias, *, out_layout=LAYOUT_NCDHW, negative_slope=0.2, packed=None, out=None):
    """The encoder's first block on cat([x0, rest], dim=1) WITHOUT the concatenation: x0 (B,1,D,W,H) = the moving image,
    rest (B,P,D,W,H) = the backprojected views, P in {1,2}.  Same kernel and bits as conv3d_k3_lrelu on the
    concatenated tensor (reference …Backproj.py:95-98 + layers.py:365-369)."""
    if not (isinstance(x0, torch.Tensor) and x0.is_cuda and x0.dtype == torch.float32):
        raise _hip.LiftRegHipError("x0: must be a float32 GPU tensor (no CPU fallback)")
    rest = _dev(rest, "rest")
    if not conv3d_first_split_supported(x0, rest):
        raise ValueError("conv3d_first_split: unsupported shapes (concatenate and use conv3d_k3_lrelu)")
    B, _, D, W, H = x0.shape
    Cin, Cout = 1 + rest.shape[1], weight.shape[0]
    if weight.shape[1] != Cin:
        raise ValueError(f"weight expects Cin={weight.shape[1]}, inputs have {Cin}")
    if packed is None:
        packed = conv3d_pack_weights(weight, LAYOUT_NCDHW)
    b = None if bias is None else _dev(bias.detach(), "bias")
    shape = (B, Cout, D, W, H) if out_layout == LAYOUT_NCDHW else (B, D, W, H, Cout)
    y = _conv_out(out, shape, torch.float32, x0.device, strided_batch=True)
    obs = _batch_stride(y)
    ibs = int(x0.stride(0)) if (B > 1 and not x0.is_contiguous()) else 0
    with _timed(f"conv3d_c{Cin}x{Cout}_s1_{D}", flops=2.0 * 27 * Cin * Cout * B * D * W * H,
                bytes=4 * (x0.numel() + rest.numel()) + 4 * y.numel(), samples=B):
        if obs or ibs:
            _hip.check(_hip.lib().lr_conv3d_first_split_obs_f32(x0.data_ptr(), ibs, rest.data_ptr(), packed.data_ptr(), _ptr(b),
                                                                y.data_ptr(), B, Cin, Cout, D, W, H, out_layout,
                                                                float(negative_slope), obs, _stream()), "lr_conv3d_first_split_obs_f32")
        else:
            _hip.check(_hip.lib().lr_conv3d_first_split_f32(x0.data_ptr(), rest.data_ptr(), packed.data_ptr(), _ptr(b),
                                                            y.data_ptr(), B, Cin, Cout, D, W, H, out_layout,
                                                            float(negative_slope), _stream()), "lr_conv3d_first_split_f32")
    return y


def conv3d_pair01_shapes_supported(B, Cin, D, W, H, w0, w1, out_layout=LAYOUT_NDHWC_HPS):
    """The shape half of `conv3d_pair01_supported` (no pointers, no strides): the same answer in every process."""
    if not (2 <= Cin <= 5 and H % 4 == 0 and B >= 1 and D >= 1):
        return False
    if tuple(w0.shape) != (16, Cin, 3, 3, 3) or tuple(w1.shape) != (32, 16, 3, 3, 3):
        return False
    if out_layout not in (LAYOUT_NDHWC, LAYOUT_NDHWC_HPS) or (out_layout == LAYOUT_NDHWC_HPS and ((H - 1) // 2 + 1) % 2):
        return False
    V = D * W * H
    return 4 * max(Cin - 1, 3) * V + 32 * W * H < 2 ** 31 - 1 and ((W - 1) // 2 + 1) * ((H - 1) // 2 + 1) * 128 < 2 ** 31 - 1


def conv3d_pair01_supported(x0, rest, w0, w1, out_layout=LAYOUT_NDHWC_HPS, probe=False):
    """True when `conv3d_pair01` (encoder blocks 0 and 1 as one kernel, csrc/conv01_fused.hip) can take these tensors
    (probe: `rest` is a shape-only stand-in — a freshly allocated tensor of that shape is contiguous and aligned)."""
    if not (x0.dim() == 5 and rest.dim() == 5 and x0.shape[1] == 1 and rest.shape[1] in (1, 2, 3, 4) and x0.shape[4] % 4 == 0 and
            x0.shape[0] == rest.shape[0] and x0.shape[2:] == rest.shape[2:] and x0[0].is_contiguous() and
            (probe or rest.is_contiguous()) and (x0.shape[0] == 1 or x0.stride(0) % 4 == 0) and x0.data_ptr() % 16 == 0 and
            (probe or rest.data_ptr() % 16 == 0)):
        return False
    B, _, D, W, H = x0.shape
    if tuple(w0.shape) != (16, 1 + rest.shape[1], 3, 3, 3) or tuple(w1.shape) != (32, 16, 3, 3, 3):
        return False
    if out_layout not in (LAYOUT_NDHWC, LAYOUT_NDHWC_HPS) or (out_layout == LAYOUT_NDHWC_HPS and ((H - 1) // 2 + 1) % 2):
        return False
    V = D * W * H
    return 4 * max(rest.shape[1], 3) * V + 32 * W * H < 2 ** 31 - 1 and ((W - 1) // 2 + 1) * ((H - 1) // 2 + 1) * 128 < 2 ** 31 - 1


def _pair01_mfmas_per_step(Cin):
    """What the matrix pipe is really asked for by the pair kernel: v_mfma_f32_16x16x32_bf16 per column of 4 x 8 outputs and
    step (one block-1 output plane).  Block 0: 20 tiles of 16 voxels x 17 MFMAs (three channels: K packed densely, 135 record
    slots of 136), x 27 (five channels: 216 of 216) or x 24 (K padded 27 taps -> 32, channels -> 4); block 1: 96 + 96 + 72 + 72
    (see the kernel header)."""
    dense = Cin == 3 and os.environ.get("LIFTREG_PAIR01_DENSE", "1") != "0"
    return 20 * (27 if Cin == 5 else 17 if dense else 24) + 336


def conv3d_pair01_pack(w0, w1):
    """The two (Cout,Cin,3,3,3) weights as the three-way bf16 split fragments of lr_conv3d_pair01_f32."""
    w0, w1 = _dev(w0.detach(), "w0"), _dev(w1.detach(), "w1")
    Cin = w0.shape[1]
    n = _hip.lib().lr_conv3d_pair01_packed_floats(Cin, w0.shape[0], w1.shape[0])
    if n <= 0:
        raise ValueError("conv3d_pair01_pack: the pair kernel is built for Cin in 1..5 -> 16 -> 32 channels")
    packed = torch.empty((n,), dtype=torch.float32, device=w0.device)
    _hip.check(_hip.lib().lr_conv3d_pair01_pack_f32(w0.data_ptr(), w1.data_ptr(), packed.data_ptr(), Cin, w0.shape[0],
                                                    w1.shape[0], _stream()), "lr_conv3d_pair01_pack_f32")
    return packed


def conv3d_pair01(x0, rest, w0, b0, w1, b1, *, out_layout=LAYOUT_NDHWC_HPS, slope0=0.2, slope1=0.2, packed=None, out=None,
                  slab=None):
    """Encoder blocks 0 and 1 on cat([x0, rest], dim=1) as ONE kernel: Conv3d(Cin->16, s1) + LeakyReLU, Conv3d(16->32, s2) +
    LeakyReLU (reference layers.py:365-369 twice, …Backproj.py:95-100); the 16-channel activation never reaches memory.
    fp32 in / fp32 out, exact three-way bf16 operand splits on the bf16 MFMA.  Returns (B,Do,Wo,Ho,32) in `out_layout`.
    slab = (D_global, z_lo, oz_lo, n_oz): x0 / rest hold the global planes [z_lo, z_lo + D) of a D_global-plane volume and
    the output planes [oz_lo, oz_lo + n_oz) are computed (z-slab sharding, parallel.py) — same bits as the whole volume."""
    if not (isinstance(x0, torch.Tensor) and x0.is_cuda and x0.dtype == torch.float32):
        raise _hip.LiftRegHipError("x0: must be a float32 GPU tensor (no CPU fallback)")
    rest = _dev(rest, "rest")
    if not conv3d_pair01_supported(x0, rest, w0, w1, out_layout):
        raise ValueError("conv3d_pair01: unsupported shapes (run the two blocks separately)")
    B, _, D, W, H = x0.shape
    Cin = 1 + rest.shape[1]
    if packed is None:
        packed = conv3d_pair01_pack(w0, w1)
    b0 = None if b0 is None else _dev(b0.detach(), "b0")
    b1 = None if b1 is None else _dev(b1.detach(), "b1")
    Dg, z_lo, oz_lo, Do = (D, 0, 0, (D - 1) // 2 + 1) if slab is None else (int(v) for v in slab)
    Wo, Ho = (W - 1) // 2 + 1, (H - 1) // 2 + 1
    y = _conv_out(out, (B, Do, Wo, Ho, 32), torch.float32, x0.device, strided_batch=True)
    obs = _batch_stride(y)
    ibs = int(x0.stride(0)) if (B > 1 and not x0.is_contiguous()) else 0
    V = 2 * Do * W * H
    flops = 2.0 * 27 * B * (Cin * 16 * V + 16 * 32 * Do * Wo * Ho)
    issued = 16384.0 * _pair01_mfmas_per_step(Cin) * Do * B * (-(-Wo // 4)) * (-(-Ho // 8))
    with _timed(f"conv3d_pair01_c{Cin}x16x32_{Dg}" + ("" if slab is None else "_slab"), flops=flops, issued_bf16_flops=issued,
                bytes=4 * B * Cin * 2 * Do * W * H + 4 * y.numel(), samples=B):
        _hip.check(_hip.lib().lr_conv3d_pair01_slab_f32(x0.data_ptr(), ibs, rest.data_ptr(), 0, packed.data_ptr(), _ptr(b0),
                                                        _ptr(b1), y.data_ptr(), B, Cin, D, W, H, out_layout, float(slope0),
                                                        float(slope1), obs, Dg, z_lo, oz_lo, Do, _stream()),
                   "lr_conv3d_pair01_slab_f32")
    return y


def conv3d_pair01_train_supported(x, w0, w1, mid_layout, out_layout, rest=None):
    """True when `conv3d_pair01_train` can take the encoder's input (training forward): `x` the (B,Cin,D,W,H) NCDHW input, or —
    with `rest` (B,P,D,W,H) — `x` = the (B,1,D,W,H) moving image alone (dense per sample, any even batch stride)."""
    if rest is None:
        if not (x.dim() == 5 and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] in (1, 2, 3, 4, 5) and
                x.shape[4] % 4 == 0 and x.data_ptr() % 16 == 0):
            return False
        Cin = x.shape[1]
    else:
        if not (x.dim() == 5 and rest.dim() == 5 and x.is_cuda and rest.is_cuda and x.dtype == torch.float32 and rest.dtype == torch.float32 and
                x.shape[1] == 1 and rest.shape[1] in (1, 2, 3, 4) and x.shape[0] == rest.shape[0] and x.shape[2:] == rest.shape[2:] and
                x[0].is_contiguous() and rest.is_contiguous() and x.shape[4] % 4 == 0 and x.data_ptr() % 16 == 0 and
                rest.data_ptr() % 16 == 0 and (x.shape[0] == 1 or x.stride(0) % 4 == 0)):
            return False
        Cin = 1 + rest.shape[1]
    B, _, D, W, H = x.shape
    if tuple(w0.shape) != (16, Cin, 3, 3, 3) or tuple(w1.shape) != (32, 16, 3, 3, 3):
        return False
    if mid_layout not in (LAYOUT_NDHWC, LAYOUT_NDHWC_HPS) or out_layout not in (LAYOUT_NDHWC, LAYOUT_NDHWC_HPS):
        return False
    if out_layout == LAYOUT_NDHWC_HPS and ((H - 1) // 2 + 1) % 2:
        return False
    V = D * W * H
    return 64 * V < 2 ** 31 - 1 and 4 * max(3, Cin - 1) * V + 32 * W * H < 2 ** 31 - 1


def conv3d_pair01_train(x, w0, b0, w1, b1, *, mid_layout=LAYOUT_NDHWC_HPS, out_layout=LAYOUT_NDHWC_HPS, slope0=0.2, slope1=0.2,
                        packed=None, rest=None):
    """Training forward of encoder blocks 0 and 1 as the fused pair kernel (conv3d_pair01) that also writes what the backward
    reads: returns (y1 (B,Do,Wo,Ho,32), y0 (B,D,W,H,16) in mid_layout, mask0 (B,D,W,H,4) uint8).  `rest`: `x` is the moving image
    alone and `rest` the backprojected views — the input is read from the two buffers, never concatenated."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
        raise _hip.LiftRegHipError("x: must be a float32 GPU tensor (no CPU fallback)")
    if not conv3d_pair01_train_supported(x, w0, w1, mid_layout, out_layout, rest):
        raise ValueError("conv3d_pair01_train: unsupported shapes (run the two blocks separately)")
    B, _, D, W, H = x.shape
    Cin = x.shape[1] if rest is None else 1 + rest.shape[1]
    if packed is None:
        packed = conv3d_pair01_pack(w0, w1)
    b0 = None if b0 is None else _dev(b0.detach(), "b0")
    b1 = None if b1 is None else _dev(b1.detach(), "b1")
    Do, Wo, Ho = (D - 1) // 2 + 1, (W - 1) // 2 + 1, (H - 1) // 2 + 1
    y1 = torch.empty((B, Do, Wo, Ho, 32), dtype=torch.float32, device=x.device)
    y0 = torch.empty((B, D, W, H, 16), dtype=torch.float32, device=x.device)
    mask0 = torch.empty((B, D, W, H, 4), dtype=torch.uint8, device=x.device)
    V = D * W * H
    flops = 2.0 * 27 * B * (Cin * 16 * 2 * Do * W * H + 16 * 32 * Do * Wo * Ho)
    issued = 16384.0 * _pair01_mfmas_per_step(Cin) * Do * B * (-(-Wo // 4)) * (-(-Ho // 8))
    if rest is None:
        p0, s0, pr, sr = x.data_ptr(), Cin * V, x.data_ptr() + 4 * V, Cin * V
    else:
        p0, s0, pr, sr = x.data_ptr(), (int(x.stride(0)) if B > 1 else V), rest.data_ptr(), (Cin - 1) * V
    with _timed(f"conv3d_pair01_train_c{Cin}x16x32_{D}", flops=flops, issued_bf16_flops=issued,
                bytes=4 * Cin * B * V + 4 * y1.numel() + 4 * y0.numel() + mask0.numel(), samples=B):
        _hip.check(_hip.lib().lr_conv3d_pair01_train_f32(p0, s0, pr, sr, packed.data_ptr(),
                                                         _ptr(b0), _ptr(b1), y1.data_ptr(), y0.data_ptr(), mask0.data_ptr(), B, Cin,
                                                         D, W, H, mid_layout, out_layout, float(slope0), float(slope1), _stream()),
                   "lr_conv3d_pair01_train_f32")
    return y1, y0, mask0


def conv3d_first_fused_bp_supported(x0, proj):
    """True when `conv3d_first_fused_bp` can take these tensors (else: backproject + conv3d_first_split).  EXPERIMENTAL: only the
    `make exp` build of the library exports the kernel (include/liftreg_hip.h, last section); False on the product library."""
    return (_hip.has_experimental() and x0.dim() == 5 and proj.dim() == 4 and x0.shape[1] == 1 and proj.shape[1] in (1, 2) and x0.shape[4] % 4 == 0 and
            x0.shape[0] == proj.shape[0] and x0.is_contiguous() and x0.data_ptr() % 16 == 0 and proj.shape[2] >= 2 and
            proj.shape[3] >= 2 and x0.shape[2] * x0.shape[3] * x0.shape[4] * 4 * 3 < 2 ** 31 - 2 ** 26)


def conv3d_first_fused_bp(x0, proj, poses, weight, bias, *, out_layout=LAYOUT_NCDHW, negative_slope=0.2, packed=None, out=None):
    """The encoder's first block with the backprojection computed inside its staging: x0 (B,1,D,W,H) =
    the moving image, proj (B,P,Pw,Ph) = the views, poses (P,3) = ONE emitter geometry.  The (B,P,D,W,H) feature volume
    of …Backproj.py:85-98 is never written; output bits = backproject + conv3d_first_split."""
    x0, proj = _dev(x0, "x0"), _dev(proj, "proj")
    if not conv3d_first_fused_bp_supported(x0, proj):
        raise ValueError("conv3d_first_fused_bp: unsupported shapes (use backproject + conv3d_first_split)")
    B, _, D, W, H = x0.shape
    P, Pw, Ph = proj.shape[1], proj.shape[2], proj.shape[3]
    poses = _host_f32(poses, (3,), "poses").reshape(-1, 3)
    if poses.shape[0] != P:
        raise ValueError(f"poses has {poses.shape[0]} views, proj has {P}")
    Cin, Cout = P + 1, weight.shape[0]
    if weight.shape[1] != Cin:
        raise ValueError(f"weight expects Cin={weight.shape[1]}, inputs have {Cin}")
    if packed is None:
        packed = conv3d_pack_weights(weight, LAYOUT_NCDHW)
    b = None if bias is None else _dev(bias.detach(), "bias")
    shape = (B, Cout, D, W, H) if out_layout == LAYOUT_NCDHW else (B, D, W, H, Cout)
    y = _conv_out(out, shape, torch.float32, x0.device)
    with _timed(f"conv3d_bp_c{Cin}x{Cout}_s1_{D}", flops=2.0 * 27 * Cin * Cout * B * D * W * H,
                bytes=4 * (x0.numel() + proj.numel()) + 4 * y.numel(), samples=B):
        _hip.check(_hip.lib().lr_conv3d_first_fused_bp_f32(x0.data_ptr(), proj.data_ptr(), poses.ctypes.data, packed.data_ptr(),
                                                           _ptr(b), y.data_ptr(), B, P, Pw, Ph, Cout, D, W, H, out_layout,
                                                           float(negative_slope), _stream()), "lr_conv3d_first_fused_bp_f32")
    return y


def conv3d_pack_weights_bf16(weight):
    """(Cout,Cin,3,3,3) fp32 parameter → bf16 MFMA operand order for lr_conv3d_k3_lrelu_bf16 (uint8 buffer)."""
    weight = _dev(weight.detach(), "weight")
    Cout, Cin = weight.shape[0], weight.shape[1]
    n = _hip.lib().lr_conv3d_packed_bf16_bytes(Cin, Cout)
    if n < 0:
        _hip.check(int(n), "lr_conv3d_packed_bf16_bytes")
    packed = torch.empty((n,), dtype=torch.uint8, device=weight.device)
    _hip.check(_hip.lib().lr_conv3d_pack_weights_bf16(weight.data_ptr(), packed.data_ptr(), Cin, Cout, _stream()),
               "lr_conv3d_pack_weights_bf16")
    return packed


def conv3d_k3_lrelu_bf16(x, weight, bias, stride, *, in_layout, out_layout, negative_slope=0.2, packed=None, out=None):
    """bf16-storage variant of conv3d_k3_lrelu for the stride-2 blocks (C4/C5 "bf16 convs"): x is a bfloat16
    (B,D,W,H,Cin) tensor in LAYOUT_BF16_NDHWC[_HPS]; fp32 accumulate, bias and LeakyReLU; the output is bfloat16
    channels-last, or float32 NCDHW for the last block."""
    x = _dev(x, "x", torch.bfloat16)
    B, D, W, H, Cin = x.shape
    Cout = weight.shape[0]
    if weight.shape[1] != Cin:
        raise ValueError(f"weight expects Cin={weight.shape[1]}, input has {Cin}")
    if packed is None:
        packed = conv3d_pack_weights_bf16(weight)
    b = None if bias is None else _dev(bias.detach(), "bias")
    o = lambda n: (n - 1) // stride + 1
    Do, Wo, Ho = o(D), o(W), o(H)
    if out_layout == LAYOUT_NCDHW:
        y = _conv_out(out, (B, Cout, Do, Wo, Ho), torch.float32, x.device, strided_batch=True)
    else:
        y = _conv_out(out, (B, Do, Wo, Ho, Cout), torch.bfloat16, x.device, strided_batch=True)
    obs = _batch_stride(y)
    with _timed(f"conv3d_bf16_c{Cin}x{Cout}_s{stride}_{D}", flops=2.0 * 27 * Cin * Cout * B * Do * Wo * Ho,
                bytes=2 * x.numel() + y.numel() * y.element_size(), samples=B, peak_tf=MFMA_BF16_PEAK_TF, bound="hbm"):
        if obs:
            _hip.check(_hip.lib().lr_conv3d_k3_lrelu_obs_bf16(x.data_ptr(), packed.data_ptr(), _ptr(b), y.data_ptr(), B, Cin,
                                                              Cout, D, W, H, stride, in_layout, out_layout,
                                                              float(negative_slope), obs, _stream()), "lr_conv3d_k3_lrelu_obs_bf16")
        else:
            _hip.check(_hip.lib().lr_conv3d_k3_lrelu_bf16(x.data_ptr(), packed.data_ptr(), _ptr(b), y.data_ptr(), B, Cin,
                                                          Cout, D, W, H, stride, in_layout, out_layout,
                                                          float(negative_slope), _stream()), "lr_conv3d_k3_lrelu_bf16")
    return y


def conv3d_pack_weights_bf16_planar(weight):
    """(Cout,Cin,3,3,3) fp32 → bf16 operand order of lr_conv3d_first_bf16 (passes of 3 input channels)."""
    weight = _dev(weight.detach(), "weight")
    Cout, Cin = weight.shape[0], weight.shape[1]
    n = _hip.lib().lr_conv3d_packed_bf16_planar_bytes(Cin, Cout)
    if n < 0:
        _hip.check(int(n), "lr_conv3d_packed_bf16_planar_bytes")
    packed = torch.empty((n,), dtype=torch.uint8, device=weight.device)
    _hip.check(_hip.lib().lr_conv3d_pack_weights_bf16_planar(weight.data_ptr(), packed.data_ptr(), Cin, Cout, _stream()),
               "lr_conv3d_pack_weights_bf16_planar")
    return packed


def conv3d_first_bf16(x, weight, bias, *, out_layout, negative_slope=0.2, packed=None, out=None, mask_out=None):
    """The encoder's first block in the bf16 variant: x float32 (B,Cin,D,W,H), stride 1, output bfloat16
    (B,D,W,H,Cout) in LAYOUT_BF16_NDHWC[_HPS].  Inputs are rounded to bf16 on the way into the MFMA.
    mask_out (training, Cin <= 3): a (B,D,W,H,Cout/4) uint8 tensor that receives the LeakyReLU sign mask of the stored
    output (LAYOUT_SIGN4) for the next block's data gradient."""
    x = _dev(x, "x")
    B, Cin, D, W, H = x.shape
    Cout = weight.shape[0]
    if weight.shape[1] != Cin:
        raise ValueError(f"weight expects Cin={weight.shape[1]}, input has {Cin}")
    if packed is None:
        packed = conv3d_pack_weights_bf16_planar(weight)
    b = None if bias is None else _dev(bias.detach(), "bias")
    y = _conv_out(out, (B, D, W, H, Cout), torch.bfloat16, x.device, strided_batch=True)
    obs = _batch_stride(y)
    with _timed(f"conv3d_bf16_c{Cin}x{Cout}_s1_{D}", flops=2.0 * 27 * Cin * Cout * B * D * W * H,
                bytes=4 * x.numel() + 2 * y.numel(), samples=B, peak_tf=MFMA_BF16_PEAK_TF, bound="hbm"):
        if mask_out is not None:
            if obs or not mask_out.is_cuda or mask_out.dtype != torch.uint8 or tuple(mask_out.shape) != (B, D, W, H, Cout // 4) or \
                    not mask_out.is_contiguous():
                raise ValueError(f"mask_out must be a contiguous uint8 GPU tensor of shape {(B, D, W, H, Cout // 4)} (dense output only)")
            _hip.check(_hip.lib().lr_conv3d_first_mask_bf16(x.data_ptr(), packed.data_ptr(), _ptr(b), y.data_ptr(), mask_out.data_ptr(),
                                                            B, Cin, Cout, D, W, H, out_layout, float(negative_slope), _stream()),
                       "lr_conv3d_first_mask_bf16")
        elif obs:
            _hip.check(_hip.lib().lr_conv3d_first_obs_bf16(x.data_ptr(), packed.data_ptr(), _ptr(b), y.data_ptr(), B, Cin, Cout,
                                                           D, W, H, out_layout, float(negative_slope), obs, _stream()),
                       "lr_conv3d_first_obs_bf16")
        else:
            _hip.check(_hip.lib().lr_conv3d_first_bf16(x.data_ptr(), packed.data_ptr(), _ptr(b), y.data_ptr(), B, Cin, Cout,
                                                       D, W, H, out_layout, float(negative_slope), _stream()),
                       "lr_conv3d_first_bf16")
    return y


def encoder_input_bf16_supported(moving, proj):
    """True when `backproject_encoder_input_bf16` + `conv3d_first_clin_bf16` can replace backproject + cat + the first block
    (bf16 variant, 1..15 views, single-channel moving image)."""
    return (moving.dim() == 5 and proj.dim() == 4 and moving.shape[1] == 1 and 1 <= proj.shape[1] <= 15 and
            moving.shape[0] == proj.shape[0] and moving.shape[2] * moving.shape[3] * moving.shape[4] * 32 < 2 ** 31 - 2 ** 24 and
            moving.shape[4] <= 256 and proj.shape[3] % 4 == 0 and proj.data_ptr() % 16 == 0 and
            proj.shape[1] * 9 * (proj.shape[3] + 8) * 4 <= 140 * 1024 and proj.shape[1] * 9 * ((proj.shape[3] + 8) // 4) <= 13 * 512)


def backproject_encoder_input_bf16(moving, proj, poses, *, d0=0, d1=None, out=None):
    """Rows [d0,d1) of cat([moving, backprojection of proj], dim=1) as bfloat16 channels-last records (B, d1-d0, W, H, 16):
    channel 0 = moving, 1..P = the views' backprojection (lr_backproject_f32's values), the rest 0, rounded to nearest-even
    bf16.  The fp32 (B,P,D,W,H) feature volume of …Backproj.py:89-93 is never written (C4: 2.95 GB).  Feeds
    `conv3d_first_clin_bf16`."""
    moving, proj = _dev(moving, "moving"), _dev(proj, "proj")
    if not encoder_input_bf16_supported(moving, proj):
        raise ValueError("backproject_encoder_input_bf16: unsupported shapes")
    B, _, D, W, H = moving.shape
    P, Pw, Ph = proj.shape[1], proj.shape[2], proj.shape[3]
    d1 = D if d1 is None else int(d1)
    poses = _host_f32(poses, (3,), "poses").reshape(-1, 3)
    if poses.shape[0] != P:
        raise ValueError(f"poses has {poses.shape[0]} views, proj has {P}")
    y = _conv_out(out, (B, d1 - d0, W, H, 16), torch.bfloat16, moving.device, strided_batch=True)
    obs = _batch_stride(y) or 16 * (d1 - d0) * W * H
    with _timed("backproject_encin_bf16", bytes=4 * B * (d1 - d0) * W * H + 32 * B * (d1 - d0) * W * H + 4 * proj.numel(), samples=B):
        _hip.check(_hip.lib().lr_backproject_encin_bf16(proj.data_ptr(), moving.data_ptr(), poses.ctypes.data, y.data_ptr(), B, P, Pw,
                                                        Ph, D, W, H, int(d0), d1, obs, _stream()), "lr_backproject_encin_bf16")
    return y


def conv3d_first_clin_bf16(x_cl, weight, bias, *, out_layout, negative_slope=0.2, packed=None, out=None):
    """The bf16 variant's first block on the channels-last bf16 encoder input (B,D,W,H,16) of
    `backproject_encoder_input_bf16`: same weights and results as `conv3d_first_bf16` on the fp32 NCDHW concatenation."""
    x_cl = _dev(x_cl, "x_cl", torch.bfloat16)
    B, D, W, H, C16 = x_cl.shape
    Cout, Cin = weight.shape[0], weight.shape[1]
    if C16 != 16 or Cin > 16:
        raise ValueError("x_cl must be (B,D,W,H,16) and the weight have at most 16 input channels")
    if packed is None:
        packed = conv3d_pack_weights_bf16_planar(weight)
    b = None if bias is None else _dev(bias.detach(), "bias")
    y = _conv_out(out, (B, D, W, H, Cout), torch.bfloat16, x_cl.device, strided_batch=True)
    with _timed(f"conv3d_bf16_c{Cin}x{Cout}_s1_{D}_clin", flops=2.0 * 27 * Cin * Cout * B * D * W * H,
                bytes=2 * x_cl.numel() + 2 * y.numel(), samples=B, peak_tf=MFMA_BF16_PEAK_TF, bound="hbm"):
        _hip.check(_hip.lib().lr_conv3d_first_clin_bf16(x_cl.data_ptr(), packed.data_ptr(), _ptr(b), y.data_ptr(), B, Cin, Cout, D, W,
                                                        H, out_layout, float(negative_slope), _batch_stride(y), _stream()),
                   "lr_conv3d_first_clin_bf16")
    return y


def cast_bf16(x):
    """fp32 → bfloat16 (round to nearest even), same shape/order."""
    x = _dev(x, "x")
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _hip.check(_hip.lib().lr_cast_f32_to_bf16(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "lr_cast_f32_to_bf16")
    return y


def bf16_hps_to_ndhwc(y):
    """LAYOUT_BF16_NDHWC_HPS → plain channels-last (tests): rows are [parity][H/2][C]."""
    H = y.shape[3]
    h = torch.arange(H, device=y.device)
    return y[:, :, :, (h & 1) * (H // 2) + (h >> 1)]


def hps_to_ndhwc(y):
    """LAYOUT_NDHWC_HPS → plain NDHWC (tests / debugging): rows are [C/16][parity][H/2][16]."""
    B, D, W, H, C = y.shape
    h = torch.arange(H, device=y.device)
    rows = y.reshape(B, D, W, C // 16, H, 16)[:, :, :, :, (h & 1) * (H // 2) + (h >> 1)]
    return rows.permute(0, 1, 2, 4, 3, 5).reshape(B, D, W, H, C)


# ----------------------------------------------------------------------------- K4 linear
def linear_lrelu(x, weight, bias, negative_slope=1.0):
    """act(x @ W^T + b), B <= 32.  Replaces FullyConnectBlock (reference layers/layers.py:413-439)."""
    x = _dev(x, "x")
    w = _dev(weight.detach(), "weight")
    b = None if bias is None else _dev(bias.detach(), "bias")
    B, K = x.shape
    O = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"weight expects K={w.shape[1]}, input has {K}")
    if B > 32:     # the kernel holds the whole batch per block (B <= 32): larger batches go in chunks
        return torch.cat([linear_lrelu(x[i:i + 32], weight, bias, negative_slope) for i in range(0, B, 32)], 0)
    y = torch.empty((B, O), dtype=torch.float32, device=x.device)
    with _timed(f"linear_{K}x{O}", bytes=4 * (w.numel() + x.numel() + y.numel())):
        _hip.check(_hip.lib().lr_linear_lrelu_f32(x.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), B, K, O,
                                                  float(negative_slope), _stream()), "lr_linear_lrelu_f32")
    return y


# ----------------------------------------------------------------------------- K5 PCA
def pca_reconstruct(coefs, basis_LxM, mean, *, out=None):
    """disp (B,M) = coefs (B,L) @ basis (L,M) + mean (M).  Replaces F.linear at …Backproj.py:102.

    `basis_LxM` may be a column slab view (stride(0) >= M) of the full (L,3V) basis.
    """
    coefs = _dev(coefs, "coefs")
    if not basis_LxM.is_cuda or basis_LxM.dtype not in (torch.float32, torch.bfloat16) or basis_LxM.stride(1) != 1:
        raise TypeError("basis must be a float32 (or bfloat16-stored) GPU tensor with unit column stride")
    bf = basis_LxM.dtype == torch.bfloat16
    mean = _dev(mean, "mean")
    B, L = coefs.shape
    M = basis_LxM.shape[1]
    if basis_LxM.shape[0] != L or mean.shape[0] != M:
        raise ValueError("basis/mean shape mismatch")
    if out is None:
        out = torch.empty((B, M), dtype=torch.float32, device=coefs.device)
    else:
        out = _out_arg(out, (B, M))
    if B > 32:     # the entry point tiles up to 32 batch rows: larger batches in chunks (the basis is re-read per chunk)
        for i in range(0, B, 32):
            pca_reconstruct(coefs[i:i + 32], basis_LxM, mean, out=out[i:i + 32])
        return out
    fn = _hip.lib().lr_pca_reconstruct_bf16basis_f32 if bf else _hip.lib().lr_pca_reconstruct_f32
    with _timed("pca_reconstruct" + ("_bf16basis" if bf else ""), bytes=(2 if bf else 4) * L * M + 4 * (M + B * M), samples=B):
        _hip.check(fn(coefs.data_ptr(), basis_LxM.data_ptr(), mean.data_ptr(), out.data_ptr(), B, L, M,
                      basis_LxM.stride(0), M, _stream()), "lr_pca_reconstruct_f32")
    return out


# ----------------------------------------------------------------------------- K6/K7 warp
def warp(img, disp, ids=None, seg=None, *, using_scale=True, zero_boundary=True, mode="bilinear",
         d0=0, d1=None, want_phi=True):
    """phi = disp + identity ; warped = Bilinear(img', phi).  Returns (phi or None, warped).

    img (B,C,D,W,H) whole volume; disp (B,3,Dn,W,H) rows [d0,d1); ids = three per-axis
    identity tables (rows d0.. for axis 0) or None when `disp` already is phi.
    Replaces Bilinear (reference net_utils.py:9-56) + the add at …Backproj.py:68.
    """
    img = _dev(img, "img")
    disp = _dev(disp, "disp")
    B, Cc, D, W, H = img.shape
    d1 = D if d1 is None else int(d1)
    Dn = d1 - d0
    if tuple(disp.shape) != (B, 3, Dn, W, H):
        raise ValueError(f"disp must be {(B, 3, Dn, W, H)}, got {tuple(disp.shape)}")
    if mode not in ("bilinear", "nearest"):
        raise ValueError("mode must be 'bilinear' or 'nearest'")
    sg = None if seg is None else _dev(seg, "seg")
    if sg is not None and sg.shape != img.shape:
        raise ValueError("seg must match img")
    i0 = i1 = i2 = None
    if ids is not None:
        i0, i1, i2 = (_dev(t, "id table") for t in ids)
        if i0.numel() != Dn or i1.numel() != W or i2.numel() != H:
            raise ValueError("identity tables must have (Dn, W, H) entries")
    phi = torch.empty_like(disp) if want_phi else None
    warped = torch.empty((B, Cc, Dn, W, H), dtype=torch.float32, device=img.device)
    flags = ((_hip.WARP_USING_SCALE if using_scale else 0) | (0 if zero_boundary else _hip.WARP_BORDER) |
             (_hip.WARP_NEAREST if mode == "nearest" else 0))
    nb = 4 * (disp.numel() + (phi.numel() if want_phi else 0) + warped.numel() + B * Cc * Dn * W * H)
    with _timed("warp_trilinear", bytes=nb, samples=B):
        _hip.check(_hip.lib().lr_warp_trilinear_f32(img.data_ptr(), _ptr(sg), disp.data_ptr(), _ptr(i0),
                                                    _ptr(i1), _ptr(i2), _ptr(phi), warped.data_ptr(), B, Cc, D,
                                                    W, H, d0, d1, flags, _stream()), "lr_warp_trilinear_f32")
    return phi, warped


def pca_warp_supported(coefs, basis_LxM, img, d0=0, d1=None):
    """True when the one-pass decode `pca_warp` can run these tensors (else: pca_reconstruct + warp).  The basis is
    the full (L,3V) array, or — for rows [d0,d1) — a rank's compact (L,3·Dn·W·H) slab of it."""
    B, C, D, W, H = img.shape
    V = D * W * H
    Dn = (D if d1 is None else d1) - d0
    cols = basis_LxM.shape[1]
    return (H % 4 == 0 and cols in (3 * V, 3 * Dn * W * H) and basis_LxM.stride(1) == 1 and basis_LxM.stride(0) % 4 == 0 and
            basis_LxM.shape[0] <= 2048 and 4 * V + 4 * W * H <= 2 ** 31 and W * H < 2 ** 22 and D <= 65535 and
            coefs.shape[0] == B)


def pca_warp(coefs, basis_LxM, mean, ids, img, *, using_scale=True, d0=0, d1=None, target=None):
    """disp = coefs·basis + mean ; phi = disp + identity ; warped = Bilinear(img, phi) in ONE kernel:
    the displacement field is written once and never read back.  Returns (disp, phi, warped), bit-identical to
    `pca_reconstruct` followed by `warp`.  Batches above 8: chunks of 8 rows inside one launch (basis read from HBM once).
    Replaces …Backproj.py:102 + :68-69 in inference.

    Rows [d0,d1) (z-slab sharding): outputs are slabs; `basis_LxM`/`mean` are the full (L,3V)/(3V,) arrays or a rank's
    compact (L,3·Dn·W·H)/(3·Dn·W·H,) slabs; ids[0] = the D-axis identity table of the slab's rows (Dn entries) or the
    whole table.  `target` (B,1,Dn,W,H), single-channel images only: the five fp64 NCC moments per batch row are
    accumulated in the kernel's epilogue and returned as a 4th value — the caller hands them to the similarity
    explicitly (`NCCLoss(warped, target, moments=…)`, model output key "ncc_moments"); nothing is cached behind the
    caller's back, so a later in-place change of `warped`/`target` cannot meet stale moments."""
    coefs, mean, img = _dev(coefs, "coefs"), _dev(mean, "mean"), _dev(img, "img")
    if not basis_LxM.is_cuda or basis_LxM.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("basis must be a float32 (or bfloat16-stored) GPU tensor")
    bf = basis_LxM.dtype == torch.bfloat16
    basis = basis_LxM
    if coefs.shape[1] != basis.shape[0] or mean.shape[0] != basis.shape[1]:
        raise ValueError("basis/mean shape mismatch")
    B, C, D, W, H = img.shape
    d1 = D if d1 is None else int(d1)
    Dn = d1 - d0
    if not pca_warp_supported(coefs, basis, img, d0, d1):
        raise ValueError("pca_warp: unsupported shapes (use pca_reconstruct + warp)")
    L = basis.shape[0]
    V, Vs = D * W * H, Dn * W * H
    compact = basis.shape[1] == 3 * Vs and Dn != D
    bcs = Vs if compact else V                       # columns between the three component thirds
    col0 = 0 if compact else d0 * W * H              # column of (component 0, row d0)
    esz = 2 if bf else 4
    i0, i1, i2 = (_dev(t, "id table") for t in ids)
    if i0.numel() == D and Dn != D:
        i0 = i0[d0:d1].contiguous()
    if i0.numel() != Dn or i1.numel() != W or i2.numel() != H:
        raise ValueError("identity tables must have (Dn, W, H) entries")
    disp = torch.empty((B, 3, Dn, W, H), dtype=torch.float32, device=img.device)
    phi = torch.empty_like(disp)
    warped = torch.empty((B, C, Dn, W, H), dtype=torch.float32, device=img.device)
    moments = None
    if target is not None:
        if C != 1:
            raise ValueError("pca_warp(target=…): single-channel images only")
        target = _dev(target, "target")
        if tuple(target.shape) != (B, 1, Dn, W, H):
            raise ValueError(f"target must be {(B, 1, Dn, W, H)}, got {tuple(target.shape)}")
        nblk = ((W * H // 4 + 255) // 256) * Dn * 4        # one partial per wave
        partial = torch.empty((min(B, 8), nblk, 5), dtype=torch.float64, device=img.device)
        moments = torch.empty((B, 5), dtype=torch.float64, device=img.device)
    # without the moments ONE launch takes the whole batch (chunks of 8 rows of one tile share an XCD's L2: the basis leaves HBM
    # once); with them: a launch per chunk of 8 (the partials are indexed by block)
    step = 8 if target is not None else 256
    for lo in range(0, B, step):
        hi = min(B, lo + step)
        nb = esz * L * 3 * Vs + 4 * 3 * Vs + (hi - lo) * 4 * Vs * (6 + 2 * C) + (4 * (hi - lo) * Vs if target is not None else 0)
        name = "pca_warp" + ("_ncc" if target is not None else "") + ("_bf16basis" if bf else "")
        with _timed(name, bytes=nb, samples=hi - lo):
            _hip.check(_hip.lib().lr_pca_warp_slab_f32(
                coefs[lo:hi].data_ptr(), basis.data_ptr() + col0 * esz, int(bf), mean.data_ptr() + col0 * 4,
                img[lo:hi].data_ptr(), i0.data_ptr(), i1.data_ptr(), i2.data_ptr(), disp[lo:hi].data_ptr(),
                phi[lo:hi].data_ptr(), warped[lo:hi].data_ptr(), hi - lo, L, C, D, W, H, d0, d1, basis.stride(0), bcs,
                _hip.WARP_USING_SCALE if using_scale else 0,
                None if target is None else target[lo:hi].data_ptr(), None if target is None else partial.data_ptr(),
                None if target is None else moments[lo:hi].data_ptr(), _stream()), "lr_pca_warp_slab_f32")
    if target is not None:
        return disp, phi, warped, moments
    return disp, phi, warped


def mask_compose(img, seg):
    """(img+1)*seg-1  (reference …Backproj.py:57-58)."""
    img = _dev(img, "img")
    seg = _dev(seg, "seg")
    out = torch.empty_like(img)
    with _timed("mask_compose", bytes=12 * img.numel()):
        _hip.check(_hip.lib().lr_mask_compose_f32(img.data_ptr(), seg.data_ptr(), out.data_ptr(), img.numel(),
                                                  _stream()), "lr_mask_compose_f32")
    return out


# ----------------------------------------------------------------------------- K8 NCC
def ncc_moments(x, y, rows, nblk=None):
    """Five raw fp64 moments per row: (R,5) = Σx, Σy, Σxy, Σxx, Σyy.  Slab moments add."""
    x = _dev(x, "x")
    y = _dev(y, "y")
    if x.shape != y.shape:
        raise ValueError("x and y must have the same shape")
    R = int(rows)
    N = x.numel() // R
    if nblk is None:
        nblk = max(1, min(2048 // R if R < 2048 else 1, (N + 4095) // 4096))
    partial = torch.empty((R, nblk, 5), dtype=torch.float64, device=x.device)
    moments = torch.empty((R, 5), dtype=torch.float64, device=x.device)
    with _timed("ncc_moments", bytes=8 * x.numel()):
        _hip.check(_hip.lib().lr_ncc_moments_f32(x.data_ptr(), y.data_ptr(), partial.data_ptr(),
                                                 moments.data_ptr(), R, N, nblk, _stream()), "lr_ncc_moments_f32")
    return moments


def ncc_loss_from_moments(moments, n_total, n_batch, variant=_hip.NCC_CONFIGURED):
    """(loss scalar tensor, per-row ncc) from (R,5) moments over n_total elements per row."""
    moments = _dev(moments, "moments", torch.float64)
    R = moments.shape[0]
    loss = torch.empty((), dtype=torch.float32, device=moments.device)
    rows = torch.empty((R,), dtype=torch.float32, device=moments.device)
    _hip.check(_hip.lib().lr_ncc_loss_from_moments(moments.data_ptr(), loss.data_ptr(), rows.data_ptr(), R,
                                                   int(n_total), int(n_batch), variant, _stream()),
               "lr_ncc_loss_from_moments")
    return loss, rows


def ncc_loss(x, y, variant=_hip.NCC_CONFIGURED):
    """1 - mean NCC.  variant CONFIGURED: rows = batch (layers/losses.py:14-29);
    SQUARED: rows = batch*channels, squared NCC (layers/layers.py:238-255)."""
    n_batch = x.shape[0]
    R = n_batch if variant == _hip.NCC_CONFIGURED else n_batch * x.shape[1]
    m = ncc_moments(x, y, R)
    loss, _ = ncc_loss_from_moments(m, x.numel() // R, n_batch, variant)
    return loss


# ----------------------------------------------------------------------------- a16 regulariser
def disp_reg(disp, nblk=None):
    """mean_{b,voxel} Σ_{c,axis} (∂_axis disp_c)² — the regulariser of SubspaceLoss (reference
    losses/SubspaceLoss.py:51-67).  PARITY UNPINNED: mermaid's stencil is assumed (central differences,
    linearly extrapolated faces, spacing 2/(shape-1))."""
    disp = _dev(disp, "disp")
    B, C3, D, W, H = disp.shape
    if C3 != 3:
        raise ValueError("disp must be (B,3,D,W,H)")
    if nblk is None:
        nblk = max(1, min(4096, (D * W * H + 1023) // 1024))   # partial sums per batch element (the marching kernel: 3 x row blocks x plane chunks)
    partial = torch.empty((B, nblk), dtype=torch.float64, device=disp.device)
    out = torch.empty((), dtype=torch.float32, device=disp.device)
    with _timed("disp_reg", bytes=4 * disp.numel()):
        _hip.check(_hip.lib().lr_disp_reg_f32(disp.data_ptr(), partial.data_ptr(), out.data_ptr(), B, D, W, H, nblk,
                                              _stream()), "lr_disp_reg_f32")
    return out


def subspace_reg_gram(basis_LxM, mean, img_sz):
    """(gram (L,L) fp64, lin (L,) fp64, r0 (1,) fp64) of the regulariser's bilinear form q on the PCA basis:
    gram[k][k'] = q(basis_k, basis_k'), lin[k] = q(mean, basis_k), r0 = q(mean, mean), with q(u,u) = disp_reg(u) for one
    field — what `subspace_reg` needs to evaluate the regulariser (reference losses/SubspaceLoss.py:51-67) on the
    coefficients instead of on the field.  One-off per basis, computed with the FIELD kernels so that both routes share the
    stencil: A·u comes from the regulariser's own gradient kernel (8 basis rows at a time as a batch of fields), the inner
    products with all rows from the PCA-gradient kernel with its partial sums added in fp64."""
    from . import ops_bwd
    L, M = basis_LxM.shape
    D, W, H = (int(v) for v in img_sz)
    if M != 3 * D * W * H or mean.numel() != M:
        raise ValueError("basis / mean do not match the image size")
    dev = basis_LxM.device
    one = torch.ones((), dtype=torch.float32, device=dev)
    gram = torch.empty((L, L), dtype=torch.float64, device=dev)
    for k0 in range(0, L, 8):
        k1 = min(L, k0 + 8)
        n = k1 - k0
        fields = basis_LxM[k0:k1].to(torch.float32).contiguous().view(n, 3, D, W, H)   # (a view for an fp32 basis)
        g = ops_bwd.disp_reg_bwd(fields, one)              # d/du_i of (1/n) sum_i q(u_i,u_i) = (2/n) A u_i
        gram[k0:k1] = ops_bwd.pca_bwd_coef_f64(g, basis_LxM) * (n / 2.0)
        del g, fields
    gram = 0.5 * (gram + gram.t())
    mu = mean.to(torch.float32).contiguous().view(1, 3, D, W, H)
    g = ops_bwd.disp_reg_bwd(mu, one)                      # 2 A mean
    lin = ops_bwd.pca_bwd_coef_f64(g, basis_LxM)[0] * 0.5
    r0 = disp_reg(mu).to(torch.float64).reshape(1)
    return gram.contiguous(), lin.contiguous(), r0


def subspace_reg(coefs, gram, lin, r0, want_grad=True):
    """The regulariser of SubspaceLoss evaluated on the PCA coefficients: R = r0 + mean_b(2 lin.c_b + c_b^T gram c_b)
    (see `subspace_reg_gram`).  Returns (R as a 0-d fp32 tensor, dR/dcoefs (B,L) fp32 or None)."""
    coefs = _dev(coefs, "coefs")
    B, L = coefs.shape
    if tuple(gram.shape) != (L, L) or tuple(lin.shape) != (L,) or gram.dtype != torch.float64 or lin.dtype != torch.float64 or \
            r0.dtype != torch.float64 or not (gram.is_cuda and lin.is_cuda and r0.is_cuda) or not gram.is_contiguous():
        raise ValueError("gram / lin / r0 must be contiguous float64 GPU tensors of shapes (L,L) / (L,) / (1,)")
    out = torch.empty((), dtype=torch.float32, device=coefs.device)
    gc = torch.empty_like(coefs) if want_grad else None
    with _timed("subspace_reg", bytes=8 * L * L):
        _hip.check(_hip.lib().lr_subspace_reg_f32(coefs.data_ptr(), gram.data_ptr(), lin.data_ptr(), r0.data_ptr(),
                                                  out.data_ptr(), _ptr(gc), B, L, _stream()), "lr_subspace_reg_f32")
    return out, gc


# ----------------------------------------------------------------------------- f3/f4: prologue and evaluation
def normalize_clip(img, lo, hi, out=None):
    """((clamp(img, lo, hi) - lo)/(hi - lo))*2 - 1 — the dataset's intensity normalisation
    (reference dataset/Registration2D3DDataset.py:196-199,207) as a GPU-side prologue."""
    img = _dev(img, "img")
    if out is None:
        out = torch.empty_like(img)
    with _timed("normalize_clip", bytes=8 * img.numel()):
        _hip.check(_hip.lib().lr_normalize_clip_f32(img.data_ptr(), out.data_ptr(), img.numel(), float(lo), float(hi),
                                                    _stream()), "lr_normalize_clip_f32")
    return out


def sample_points_f64(vol, pts):
    """Trilinear samples (zeros padding, align_corners=True) of a float64 (C,D,W,H) map at N normalised points
    (N,3) ordered (x,y,z) = (H,W,D axes) → (N,C) float64.  The landmark sampler of the reference's
    tools/evaluate_dir_lab.py:46-59 (F.grid_sample on doubles)."""
    vol = _dev(vol, "vol", torch.float64)
    pts = _dev(pts, "pts", torch.float64)
    if vol.dim() != 4 or pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("vol must be (C,D,W,H), pts (N,3)")
    C, D, W, H = vol.shape
    N = pts.shape[0]
    out = torch.empty((N, C), dtype=torch.float64, device=vol.device)
    _hip.check(_hip.lib().lr_sample_points_f64(vol.data_ptr(), pts.data_ptr(), out.data_ptr(), C, D, W, H, N, _stream()),
               "lr_sample_points_f64")
    return out


def label_overlap(pred, gt, label=1.0, nblk=1024):
    """int64 GPU tensor [|pred==label|, |gt==label|, |both|] (reference utils/metrics.py:83-121)."""
    pred, gt = _dev(pred, "pred"), _dev(gt, "gt")
    if pred.numel() != gt.numel():
        raise ValueError("pred and gt must have the same number of elements")
    nblk = max(1, min(int(nblk), (pred.numel() + 255) // 256))
    partial = torch.empty((nblk, 3), dtype=torch.int64, device=pred.device)
    counts = torch.empty((3,), dtype=torch.int64, device=pred.device)
    with _timed("label_overlap", bytes=8 * pred.numel()):
        _hip.check(_hip.lib().lr_label_overlap_f32(pred.data_ptr(), gt.data_ptr(), float(label), pred.numel(),
                                                   partial.data_ptr(), nblk, counts.data_ptr(), _stream()),
                   "lr_label_overlap_f32")
    return counts


def jacobi_det_stats(phi, spacing, nblk=None):
    """float64 GPU tensor [Σ|det J| over folded voxels, number of folded voxels] for a (B,3,D,W,H) map;
    `spacing` = the three finite-difference spacings (reference utils/utils.py:20-55).  Assumed mermaid stencil."""
    phi = _dev(phi, "phi")
    B, C3, D, W, H = phi.shape
    if C3 != 3:
        raise ValueError("map must be (B,3,D,W,H)")
    if nblk is None:
        nblk = max(1, min(2048 // B if B < 2048 else 1, (D * W * H + 255) // 256))
    partial = torch.empty((B * nblk * 2,), dtype=torch.float64, device=phi.device)
    out = torch.empty((2,), dtype=torch.float64, device=phi.device)
    sp = [float(v) for v in spacing]
    with _timed("jacobi_det_stats", bytes=4 * phi.numel()):
        _hip.check(_hip.lib().lr_jacobi_det_stats_f32(phi.data_ptr(), B, D, W, H, sp[0], sp[1], sp[2],
                                                      partial.data_ptr(), nblk, out.data_ptr(), _stream()),
                   "lr_jacobi_det_stats_f32")
    return out
