"""Tensor-level front-end of the backward kernels (liftreg_amd/csrc/backward.hip, conv3d_bwd.hip, conv3d_bf16.hip, reg.hip).

Same rules as liftreg_amd.ops: GPU tensors only, outputs allocated here, launches on torch's current stream,
no CPU/PyTorch fallback.
"""
import torch

from . import _hip
from .ops import _dev, _ptr, _stream, _timed


def ncc_bwd(x, y, moments, gout, n_total, variant=_hip.NCC_CONFIGURED):
    """d loss / d x for loss = 1 - mean_r ncc_r (x = warped).  `gout`: 0-dim GPU tensor (upstream gradient)."""
    x, y = _dev(x, "x"), _dev(y, "y")
    moments = _dev(moments, "moments", torch.float64)
    gout = _dev(gout.reshape(()).to(torch.float32), "gout")
    R = moments.shape[0]
    N = x.numel() // R
    gx = torch.empty_like(x)
    with _timed("ncc_bwd", bytes=12 * x.numel()):
        _hip.check(_hip.lib().lr_ncc_bwd_f32(x.data_ptr(), y.data_ptr(), moments.data_ptr(), gout.data_ptr(),
                                             gx.data_ptr(), R, N, int(n_total), variant, _stream()), "lr_ncc_bwd_f32")
    return gx


def warp_bwd_disp(img, disp, ids, seg, gwarped, *, using_scale=True, zero_boundary=True, d0=0, d1=None, gadd=None):
    """d / d disp of ops.warp (bilinear).  Returns (B,3,Dn,W,H); with `gadd` (same shape: the gradient reaching the
    displacement field by another path, e.g. the regulariser) the sum of the two, formed in the same pass."""
    img, disp, gwarped = _dev(img, "img"), _dev(disp, "disp"), _dev(gwarped, "gwarped")
    B, C, D, W, H = img.shape
    d1 = D if d1 is None else int(d1)
    sg = None if seg is None else _dev(seg, "seg")
    i0 = i1 = i2 = None
    if ids is not None:
        i0, i1, i2 = (_dev(t, "id table") for t in ids)
    gdisp = torch.empty_like(disp)
    flags = (_hip.WARP_USING_SCALE if using_scale else 0) | (0 if zero_boundary else _hip.WARP_BORDER)
    if gadd is not None:
        gadd = _dev(gadd, "gadd")
        if gadd.shape != disp.shape:
            raise ValueError("gadd must have the displacement field's shape")
        with _timed("warp_bwd_disp_acc", bytes=4 * (3 * disp.numel() + gwarped.numel())):
            _hip.check(_hip.lib().lr_warp_bwd_disp_acc_f32(img.data_ptr(), _ptr(sg), disp.data_ptr(), _ptr(i0), _ptr(i1),
                                                           _ptr(i2), gwarped.data_ptr(), gadd.data_ptr(), gdisp.data_ptr(),
                                                           B, C, D, W, H, d0, d1, flags, _stream()),
                       "lr_warp_bwd_disp_acc_f32")
        return gdisp
    with _timed("warp_bwd_disp", bytes=4 * (2 * disp.numel() + gwarped.numel())):
        _hip.check(_hip.lib().lr_warp_bwd_disp_f32(img.data_ptr(), _ptr(sg), disp.data_ptr(), _ptr(i0), _ptr(i1),
                                                   _ptr(i2), gwarped.data_ptr(), gdisp.data_ptr(), B, C, D, W, H, d0,
                                                   d1, flags, _stream()), "lr_warp_bwd_disp_f32")
    return gdisp


def ncc_bwd_moments(moments, gout, n, variant):
    """d loss / d moments (R,5) fp64 for the loss ops.ncc_loss_from_moments computes from the five sums of (x, y) — the
    similarity's gradient as far as the moments; `warp_bwd_disp_ncc` carries it on to the displacement field."""
    if moments.dtype != torch.float64 or not moments.is_cuda or moments.dim() != 2 or moments.shape[1] != 5:
        raise ValueError("moments must be a float64 (R,5) GPU tensor")
    gout = _dev(gout.reshape(()).to(torch.float32), "gout")
    m = moments.contiguous()
    gm = torch.empty_like(m)
    with _timed("ncc_bwd_moments", bytes=80 * m.shape[0]):
        _hip.check(_hip.lib().lr_ncc_bwd_moments(m.data_ptr(), gout.data_ptr(), gm.data_ptr(), m.shape[0], int(n), int(variant),
                                                 _stream()), "lr_ncc_bwd_moments")
    return gm


def warp_bwd_disp_ncc_supported(img, target=None, ids=None):
    """True when `warp_bwd_disp_ncc` can take this image (single channel, the vectorised kernel's shape rules) and — when
    given — this target and these identity tables (the kernel reads them with 16-byte loads: lr_warp_bwd_disp_ncc_f32 returns
    LR_EUNSUPPORTED for an unaligned view, and a decode node that has committed to the moments route has no other backward)."""
    B, C, D, W, H = img.shape
    sD = W * H
    ok = C == 1 and H % 4 == 0 and 4 * D * sD + 8 * sD <= 2 ** 31 and sD < 2 ** 23 and sD // 4 <= 2 ** 20 and D <= 65535
    if ok and target is not None:
        ok = target.is_contiguous() and target.data_ptr() % 16 == 0
    if ok and ids is not None:
        ok = ids[2].data_ptr() % 16 == 0
    return bool(ok)


def warp_bwd_disp_ncc(img, disp, ids, warped, target, gmoments, *, using_scale=True, gadd=None):
    """`warp_bwd_disp` with the gradient of `warped` given THROUGH THE SIMILARITY'S MOMENTS (`gmoments` (B,5) fp64 from
    `ncc_bwd_moments`): gw = gm0 + gm2·target + 2·gm3·warped is formed inside the kernel — the pass that wrote it
    (lr_ncc_bwd_f32: two volumes read, one written) and its read here are gone."""
    img, disp, warped, target = _dev(img, "img"), _dev(disp, "disp"), _dev(warped, "warped"), _dev(target, "target")
    B, C, D, W, H = img.shape
    if not warp_bwd_disp_ncc_supported(img) or tuple(disp.shape) != (B, 3, D, W, H) or not disp.is_contiguous() or \
            warped.shape != img.shape or target.shape != img.shape or not warped.is_contiguous() or not target.is_contiguous():
        raise ValueError("warp_bwd_disp_ncc: unsupported shapes")
    if gmoments.dtype != torch.float64 or tuple(gmoments.shape) != (B, 5) or not gmoments.is_cuda:
        raise ValueError("gmoments must be a float64 (B,5) GPU tensor")
    i0 = i1 = i2 = None
    if ids is not None:
        i0, i1, i2 = (_dev(t, "id table") for t in ids)
    if gadd is not None:
        gadd = _dev(gadd, "gadd")
        if gadd.shape != disp.shape:
            raise ValueError("gadd must have the displacement field's shape")
    gdisp = torch.empty_like(disp)
    gm = gmoments.contiguous()
    with _timed("warp_bwd_disp_ncc", bytes=4 * ((2 if gadd is None else 3) * disp.numel() + 2 * warped.numel())):
        _hip.check(_hip.lib().lr_warp_bwd_disp_ncc_f32(img.data_ptr(), disp.data_ptr(), _ptr(i0), _ptr(i1), _ptr(i2),
                                                       warped.data_ptr(), target.data_ptr(), gm.data_ptr(), _ptr(gadd),
                                                       gdisp.data_ptr(), B, D, W, H, 0, D,
                                                       _hip.WARP_USING_SCALE if using_scale else 0, _stream()),
                   "lr_warp_bwd_disp_ncc_f32")
    return gdisp


def pca_bwd_coef(gdisp, basis_LxM, nblk=None):
    """gcoefs (B,L) = gdisp (B,M) @ basis (L,M)^T — the second read of the PCA basis."""
    gdisp = _dev(gdisp, "gdisp")
    B = gdisp.shape[0]
    g2 = gdisp.reshape(B, -1)
    L, M = basis_LxM.shape
    if g2.shape[1] != M or basis_LxM.stride(1) != 1:
        raise ValueError("gdisp/basis shape mismatch")
    one_launch = (M % 4 == 0 and basis_LxM.stride(0) % 4 == 0 and g2.stride(0) % 4 == 0 and g2.data_ptr() % 16 == 0 and
                  basis_LxM.data_ptr() % 16 == 0)
    if B > 64 or (B > 8 and not one_launch):      # the kernel keeps 8 batch rows of accumulators per thread; up to 64 rows go as row
        # chunks of ONE launch (the chunks of an l-group on one XCD: the basis leaves HBM once), more — or unaligned views — in chunks of 8
        return torch.cat([pca_bwd_coef(g2[i:i + 8], basis_LxM, nblk) for i in range(0, B, 8)], 0)
    if nblk is None:
        nblk = max(1, min(256, M // 4096))   # one block per CU and l-group: 2.59 ms at C3 against 2.67 with 512, 2.82 with 1024 (NOTES_r03)
    partial = torch.empty((nblk, B, L), dtype=torch.float32, device=gdisp.device)
    gcoefs = torch.empty((B, L), dtype=torch.float32, device=gdisp.device)
    bf = basis_LxM.dtype == torch.bfloat16
    fn = _hip.lib().lr_pca_bwd_coef_bf16basis_f32 if bf else _hip.lib().lr_pca_bwd_coef_f32
    # compulsory HBM bytes: the basis once + the gradient ONCE (the ceil(L/8) l-groups that share a range of the gradient
    # run together on one XCD and take their re-reads from that L2 — they are not HBM traffic and are not counted)
    with _timed("pca_bwd_coef" + ("_bf16basis" if bf else ""), bytes=(2 if bf else 4) * L * M + 4 * B * M):
        _hip.check(fn(g2.data_ptr(), basis_LxM.data_ptr(), partial.data_ptr(), gcoefs.data_ptr(), B, L, M,
                      basis_LxM.stride(0), int(g2.stride(0)), nblk, _stream()), "lr_pca_bwd_coef_f32")
    return gcoefs


def pca_bwd_coef_f64(gdisp, basis_LxM, nblk=2048):
    """`pca_bwd_coef` with the kernel's per-block fp32 partial sums added in fp64 — for one-off precomputations
    (ops.subspace_reg_gram), not for the training step.  Returns (B,L) float64."""
    gdisp = _dev(gdisp, "gdisp")
    B = gdisp.shape[0]
    g2 = gdisp.reshape(B, -1)
    L, M = basis_LxM.shape
    if g2.shape[1] != M or basis_LxM.stride(1) != 1 or B > 8:
        raise ValueError("gdisp/basis shape mismatch (at most 8 rows)")
    nblk = max(8, (min(int(nblk), M // 1024 + 8) // 8) * 8)     # a multiple of 8: the vector kernel launches whole groups of 8 blocks
    partial = torch.zeros((nblk, B, L), dtype=torch.float32, device=gdisp.device)
    gcoefs = torch.empty((B, L), dtype=torch.float32, device=gdisp.device)
    bf = basis_LxM.dtype == torch.bfloat16
    fn = _hip.lib().lr_pca_bwd_coef_bf16basis_f32 if bf else _hip.lib().lr_pca_bwd_coef_f32
    _hip.check(fn(g2.data_ptr(), basis_LxM.data_ptr(), partial.data_ptr(), gcoefs.data_ptr(), B, L, M,
                  basis_LxM.stride(0), M, nblk, _stream()), "lr_pca_bwd_coef_f32")
    return partial.to(torch.float64).sum(0)


def _act_dims(t, layout):
    """(B, C, D, W, H) of an activation tensor stored in `layout`."""
    if layout in (_hip.LAYOUT_NCDHW, _hip.LAYOUT_NCDHW_RBF16):
        B, C, D, W, H = t.shape
    else:
        B, D, W, H, C = t.shape
    return B, C, D, W, H


def conv3d_bwd(x, x_layout, weight, y, y_layout, gy, gy_layout, stride, negative_slope=0.2, need_gx=True,
               nblk=1024, gy_is_gpre=False, mask_input_slope=None, round_weights=False, x_sign4=None):
    """Backward of ops.conv3d_k3_lrelu.  x / y: the block's saved input / output (any layout), gy: gradient of
    the output.  Returns (gx, gw (Cout,Cin,3,3,3), gb (Cout)); gx is (B,D,W,H,Cin) in x's own channels-last
    layout, or None.

    Chaining across blocks (what the model does): with `mask_input_slope` = the PRODUCER block's LeakyReLU slope,
    the data-gradient epilogue also applies the producer's mask (its output is this block's saved input), so the
    returned gx is the producer's pre-activation gradient in plain NDHWC; the producer then passes it with
    `gy_is_gpre=True` and skips its own mask pass.  The bias gradient comes out of the weight-gradient kernel.

    `x_sign4` (with `mask_input_slope`): the producer's (B,D,W,H,C/4) uint8 sign mask (ops.conv3d_k3_lrelu(mask_out=…))
    — the data gradient then reads one byte per channel quad for the producer's LeakyReLU mask instead of 16.

    bf16-forward training (conv_dtype="bf16"): x / y may be bfloat16 tensors in LAYOUT_BF16_NDHWC[_HPS] (the first
    block: its fp32 input with x_layout=LAYOUT_NCDHW_RBF16); gradients stay fp32; `round_weights` makes the data
    gradient use the bf16-rounded weights the forward multiplied by.
    """
    bf_layouts = (_hip.LAYOUT_BF16_NDHWC, _hip.LAYOUT_BF16_NDHWC_HPS)
    x = _dev(x, "x", torch.bfloat16 if x_layout in bf_layouts else torch.float32)
    y = _dev(y, "y", torch.bfloat16 if y_layout in bf_layouts else torch.float32)
    gy = _dev(gy, "gy")
    if y_layout in bf_layouts and not gy_is_gpre:
        raise ValueError("a bf16 block output needs the chained form (gy_is_gpre=True)")
    w = _dev(weight.detach(), "weight")
    if round_weights:
        w = w.to(torch.bfloat16).to(torch.float32)
    Cout, Cin = w.shape[0], w.shape[1]
    B, Cx, D, W, H = _act_dims(x, x_layout)
    _, Cy, Do, Wo, Ho = _act_dims(y, y_layout)
    if gy_is_gpre:
        gy_layout = _hip.LAYOUT_NDHWC
    if Cx != Cin or Cy != Cout or _act_dims(gy, gy_layout) != (B, Cout, Do, Wo, Ho):
        raise ValueError("conv3d_bwd: shape mismatch")
    dev = x.device
    lib = _hip.lib()
    # 1. LeakyReLU mask → gpre, plain NDHWC (skipped when the consumer block's data gradient already applied it)
    if gy_is_gpre:
        gpre = gy
    else:
        gpre = torch.empty((B, Do, Wo, Ho, Cout), dtype=torch.float32, device=dev)
        nb1 = max(1, min(1024, (gpre.numel() // 4 + 255) // 256))
        with _timed(f"lrelu_bwd_c{Cout}_{Do}", bytes=12 * gpre.numel()):
            _hip.check(lib.lr_lrelu_bwd_f32(gy.data_ptr(), gy_layout, y.data_ptr(), y_layout, gpre.data_ptr(),
                                            None, None, B, Cout, Do, Wo, Ho, float(negative_slope), nb1, _stream()),
                       "lr_lrelu_bwd_f32")
    # 2. data gradient (stride-2 blocks only; the encoder's first block has no input gradient)
    gx = None
    if need_gx:
        from .ops import conv3d_pack_weights
        packed_t = conv3d_pack_weights(w.transpose(0, 1).contiguous(), _hip.LAYOUT_NDHWC)
        gx = torch.empty((B, D, W, H, Cin), dtype=torch.float32, device=dev)
        fuse = mask_input_slope is not None
        if fuse and x_layout in (_hip.LAYOUT_NCDHW, _hip.LAYOUT_NCDHW_RBF16):
            raise ValueError("mask_input_slope needs a channels-last saved input")
        if x_layout in bf_layouts and not fuse:
            raise ValueError("a bf16 saved input needs the chained form (mask_input_slope)")
        gxl = _hip.LAYOUT_NDHWC if (fuse or x_layout == _hip.LAYOUT_NCDHW) else x_layout
        with _timed(f"conv3d_dgrad_c{Cout}x{Cin}_{D}", flops=2.0 * 27 * Cin * Cout * B * Do * Wo * Ho,
                    bytes=4 * (gpre.numel() + gx.numel() * (2 if (fuse and x_sign4 is None) else 1))):
            use_s4 = fuse and x_sign4 is not None
            if use_s4 and (x_sign4.dtype != torch.uint8 or tuple(x_sign4.shape) != (B, D, W, H, Cin // 4) or not x_sign4.is_cuda or
                            not x_sign4.is_contiguous()):
                raise ValueError(f"x_sign4 must be a contiguous uint8 GPU tensor of shape {(B, D, W, H, Cin // 4)}")
            _hip.check(lib.lr_conv3d_dgrad_f32(gpre.data_ptr(), packed_t.data_ptr(), gx.data_ptr(), B, Cout, Cin, D, W,
                                               H, stride, gxl,
                                               (x_sign4.data_ptr() if use_s4 else x.data_ptr()) if fuse else None,
                                               _hip.LAYOUT_SIGN4 if use_s4 else x_layout,
                                               float(mask_input_slope) if fuse else 1.0, _stream()),
                       "lr_conv3d_dgrad_f32")
    # 3. weight gradient (+ bias gradient)
    npart = lib.lr_conv3d_wgrad_partial_floats(Cin, Cout, x_layout, nblk)
    partial = torch.empty((npart,), dtype=torch.float32, device=dev)
    gw = torch.empty_like(w)
    gb = torch.empty((Cout,), dtype=torch.float32, device=dev)
    with _timed(f"conv3d_wgrad_c{Cin}x{Cout}_{D}", flops=2.0 * 27 * Cin * Cout * B * Do * Wo * Ho,
                bytes=x.numel() * x.element_size() + 4 * gpre.numel()):
        _hip.check(lib.lr_conv3d_wgrad_f32(x.data_ptr(), x_layout, gpre.data_ptr(), partial.data_ptr(), gw.data_ptr(),
                                           gb.data_ptr(), B, Cin, Cout, D, W, H, stride, nblk, _stream()),
                   "lr_conv3d_wgrad_f32")
    return gx, gw, gb


def lrelu_bwd(gy, gy_layout, y, y_layout, negative_slope=0.2):
    """gpre = gy * (y > 0 ? 1 : slope) as plain NDHWC fp32 (the first step of conv3d_bwd on its own)."""
    gy, y = _dev(gy, "gy"), _dev(y, "y")
    B, C, D, W, H = _act_dims(y, y_layout)
    gpre = torch.empty((B, D, W, H, C), dtype=torch.float32, device=y.device)
    nb1 = max(1, min(1024, (gpre.numel() // 4 + 255) // 256))
    with _timed(f"lrelu_bwd_c{C}_{D}", bytes=12 * gpre.numel()):
        _hip.check(_hip.lib().lr_lrelu_bwd_f32(gy.data_ptr(), gy_layout, y.data_ptr(), y_layout, gpre.data_ptr(), None,
                                               None, B, C, D, W, H, float(negative_slope), nb1, _stream()),
                   "lr_lrelu_bwd_f32")
    return gpre


def conv3d_dgrad_wgrad0_rest_ok(x0, rest):
    """The two-buffer input form of `conv3d_dgrad_wgrad0` (x0 = the moving image, rest = the views) as far as it can be judged in
    the FORWARD, before a mask exists: autograd.ConvPair01Fn saves the split input only when the backward will take it."""
    return (x0.dim() == 5 and rest.dim() == 5 and x0.shape[1] == 1 and rest.shape[1] in (1, 2, 3, 4) and x0.shape[0] == rest.shape[0] and
            x0.shape[2:] == rest.shape[2:] and x0[0].is_contiguous() and rest.is_contiguous() and rest.dtype == torch.float32 and
            x0.dtype == torch.float32 and x0.is_cuda and rest.is_cuda and (x0.shape[0] == 1 or x0.stride(0) % 2 == 0) and
            x0.shape[4] % 4 == 0 and (1 + rest.shape[1]) * x0[0].numel() * 4 < 2 ** 31 - 1)


def conv3d_dgrad_wgrad0_supported(x0, mask0, w1, rest=None):
    """True when `conv3d_dgrad_wgrad0` covers these tensors (the encoder's blocks 0/1 in fp32 training).  `rest`: block 0's input
    comes in two buffers — x0 (B,1,D,W,H) the moving image (dense per sample; a batch stride is fine), rest (B,P,D,W,H) the views."""
    if rest is not None:
        if not (x0.dim() == 5 and rest.dim() == 5 and x0.shape[1] == 1 and rest.shape[1] in (1, 2, 3, 4) and x0.shape[0] == rest.shape[0] and
                x0.shape[2:] == rest.shape[2:] and x0[0].is_contiguous() and rest.is_contiguous() and rest.dtype == torch.float32 and
                rest.is_cuda and
                (x0.shape[0] == 1 or x0.stride(0) % 2 == 0)):
            return False
        cin, per = 1 + rest.shape[1], x0[0].numel()
    else:
        if not (x0.dim() == 5 and x0.shape[1] in (2, 3, 4, 5) and x0.is_contiguous()):
            return False
        cin, per = x0.shape[1], x0[0, :1].numel()
    return (x0.shape[4] % 4 == 0 and x0.dtype == torch.float32 and x0.is_cuda and
            mask0 is not None and mask0.dtype == torch.uint8 and mask0.is_contiguous() and mask0.shape[-1] == 4 and
            tuple(w1.shape[:2]) == (32, 16) and cin * per * 4 < 2 ** 31 - 1)


def conv3d_dgrad_wgrad0(gpre1, w1, mask0, slope0, x0, rest=None):
    """(gw0 (16,Cin0,3,3,3), gb0 (16)) of the encoder's FIRST block from the pre-activation gradient `gpre1` (B,Do,Wo,Ho,32)
    of the SECOND one, w1 (32,16,3,3,3) its weight, mask0 the first block's (B,D,W,H,4) uint8 LeakyReLU sign mask
    (ops.conv3d_k3_lrelu(mask_out=…)), x0 (B,Cin0,D,W,H) the first block's input — or, with `rest`, x0 (B,1,D,W,H) + rest
    (B,Cin0-1,D,W,H) as the model holds them (no concatenated copy).  One kernel (lr_conv3d_dgrad_wgrad0[_split]_f32): block 1's
    data gradient, block 0's LeakyReLU mask and block 0's weight gradient — the (B,D,W,H,16) gradient between the two never
    reaches memory.  Same results as conv3d_bwd(block 1, need_gx) + conv3d_bwd(block 0) up to summation order."""
    gpre1 = _dev(gpre1, "gpre1")
    if not (isinstance(x0, torch.Tensor) and x0.is_cuda and x0.dtype == torch.float32):
        raise _hip.LiftRegHipError("x0: must be a float32 GPU tensor (no CPU fallback)")
    w = _dev(w1.detach(), "w1")
    B, _, D, W, H = x0.shape
    Cin0 = x0.shape[1] if rest is None else 1 + rest.shape[1]
    o = lambda n: (n - 1) // 2 + 1
    if tuple(gpre1.shape) != (B, o(D), o(W), o(H), 32) or not conv3d_dgrad_wgrad0_supported(x0, mask0, w, rest):
        raise ValueError("conv3d_dgrad_wgrad0: unsupported shapes")
    if tuple(mask0.shape) != (B, D, W, H, 4) or not mask0.is_cuda:
        raise ValueError(f"mask0 must be a uint8 GPU tensor of shape {(B, D, W, H, 4)}")
    from .ops import conv3d_pack_weights
    lib, dev = _hip.lib(), x0.device
    V = D * W * H
    packed_t = conv3d_pack_weights(w.transpose(0, 1).contiguous(), _hip.LAYOUT_NDHWC)
    partial = torch.empty((lib.lr_conv3d_dgrad_wgrad0_partial_floats(Cin0),), dtype=torch.float32, device=dev)
    gw0 = torch.empty((16, Cin0, 3, 3, 3), dtype=torch.float32, device=dev)
    gb0 = torch.empty((16,), dtype=torch.float32, device=dev)
    nvo = B * o(D) * o(W) * o(H)
    if rest is None:
        p0, s0, pr, sr = x0.data_ptr(), Cin0 * V, x0.data_ptr() + 4 * V, Cin0 * V
    else:
        p0, s0, pr, sr = x0.data_ptr(), (int(x0.stride(0)) if B > 1 else V), rest.data_ptr(), (Cin0 - 1) * V
    with _timed(f"conv3d_dgrad_wgrad0_c32x16x{Cin0}_{D}", flops=2.0 * 27 * 16 * 32 * nvo + 2.0 * 27 * Cin0 * 16 * B * D * W * H,
                bytes=4 * gpre1.numel() + mask0.numel() + 4 * Cin0 * B * V):
        _hip.check(lib.lr_conv3d_dgrad_wgrad0_split_f32(gpre1.data_ptr(), packed_t.data_ptr(), mask0.data_ptr(), float(slope0),
                                                        p0, s0, pr, sr, partial.data_ptr(), gw0.data_ptr(), gb0.data_ptr(), B, Cin0,
                                                        D, W, H, _stream()), "lr_conv3d_dgrad_wgrad0_split_f32")
    return gw0, gb0


def conv3d_bwd_bf16g(x, x_layout, weight, gpre, stride, mask_input_slope=None, nblk=1024, x_sign4=None):
    """conv3d_bwd of the bf16-GRADIENT training variant: `gpre` (B,Do,Wo,Ho,Cout) is a bfloat16 plain channels-last
    pre-activation gradient; x the block's saved input (bf16 LAYOUT_BF16_NDHWC[_HPS], or the first block's fp32 input
    with LAYOUT_NCDHW_RBF16).  Returns (the PRODUCER's pre-activation gradient as bf16 plain channels-last — its
    LeakyReLU mask applied with `mask_input_slope` — or None, gw fp32, gb fp32).  `x_sign4`: the producer's
    (B,D,W,H,Cin/4) uint8 sign mask (ops.conv3d_first_bf16(mask_out=…)): the data gradient reads it instead of x for the mask."""
    bf_layouts = (_hip.LAYOUT_BF16_NDHWC, _hip.LAYOUT_BF16_NDHWC_HPS)
    x = _dev(x, "x", torch.bfloat16 if x_layout in bf_layouts else torch.float32)
    gpre = _dev(gpre, "gpre", torch.bfloat16)
    w = _dev(weight.detach(), "weight")
    Cout, Cin = w.shape[0], w.shape[1]
    B, Cx, D, W, H = _act_dims(x, x_layout)
    o = lambda n: (n - 1) // stride + 1
    if Cx != Cin or tuple(gpre.shape) != (B, o(D), o(W), o(H), Cout):
        raise ValueError("conv3d_bwd_bf16g: shape mismatch")
    lib, dev = _hip.lib(), x.device
    gx = None
    if mask_input_slope is not None:
        if x_layout not in bf_layouts:
            raise ValueError("the data gradient needs the bf16 saved input (mask source)")
        from .ops import conv3d_pack_weights_bf16
        packed_t = conv3d_pack_weights_bf16(w.transpose(0, 1).contiguous())     # rounds like the forward's pack
        gx = torch.empty((B, D, W, H, Cin), dtype=torch.bfloat16, device=dev)
        if x_sign4 is not None and (x_sign4.dtype != torch.uint8 or tuple(x_sign4.shape) != (B, D, W, H, Cin // 4) or
                                    not x_sign4.is_cuda or not x_sign4.is_contiguous()):
            raise ValueError(f"x_sign4 must be a contiguous uint8 GPU tensor of shape {(B, D, W, H, Cin // 4)}")
        with _timed(f"conv3d_dgrad_bf16_c{Cout}x{Cin}_{D}", flops=2.0 * 27 * Cin * Cout * gpre.numel() / Cout,
                    bytes=2 * (gpre.numel() + gx.numel()) + (x_sign4.numel() if x_sign4 is not None else 2 * gx.numel())):
            _hip.check(lib.lr_conv3d_dgrad_bf16(gpre.data_ptr(), packed_t.data_ptr(), gx.data_ptr(), B, Cout, Cin, D, W, H,
                                                x_sign4.data_ptr() if x_sign4 is not None else x.data_ptr(),
                                                _hip.LAYOUT_SIGN4 if x_sign4 is not None else x_layout,
                                                float(mask_input_slope), _stream()),
                       "lr_conv3d_dgrad_bf16")
    npart = lib.lr_conv3d_wgrad_partial_floats(Cin, Cout, x_layout, nblk)
    partial = torch.empty((npart,), dtype=torch.float32, device=dev)
    gw = torch.empty_like(w)
    gb = torch.empty((Cout,), dtype=torch.float32, device=dev)
    with _timed(f"conv3d_wgrad_c{Cin}x{Cout}_{D}_bf16g", flops=2.0 * 27 * Cin * Cout * gpre.numel() / Cout,
                bytes=x.numel() * x.element_size() + 2 * gpre.numel()):
        _hip.check(lib.lr_conv3d_wgrad_bf16g_f32(x.data_ptr(), x_layout, gpre.data_ptr(), partial.data_ptr(), gw.data_ptr(),
                                                 gb.data_ptr(), B, Cin, Cout, D, W, H, stride, nblk, _stream()),
                   "lr_conv3d_wgrad_bf16g_f32")
    return gx, gw, gb


def linear_bwd(x, weight, y, gy, negative_slope=1.0, need_gx=True):
    """Backward of ops.linear_lrelu: returns (gx or None, gw, gb)."""
    x, y, gy = _dev(x, "x"), _dev(y, "y"), _dev(gy, "gy")
    w = _dev(weight.detach(), "weight")
    B, K = x.shape
    O = w.shape[0]
    if B > 32:     # chunks of 32 batch rows: gx concatenates, gw / gb add
        parts = [linear_bwd(x[i:i + 32], weight, y[i:i + 32], gy[i:i + 32], negative_slope, need_gx) for i in range(0, B, 32)]
        gx = torch.cat([p[0] for p in parts], 0) if need_gx else None
        return gx, torch.stack([p[1] for p in parts]).sum(0), torch.stack([p[2] for p in parts]).sum(0)
    gx = torch.empty_like(x) if need_gx else None
    gw = torch.empty_like(w)
    gb = torch.empty((O,), dtype=torch.float32, device=x.device)
    with _timed(f"linear_bwd_{K}x{O}", bytes=4 * (2 * w.numel() + gw.numel())):
        _hip.check(_hip.lib().lr_linear_bwd_f32(x.data_ptr(), w.data_ptr(), y.data_ptr(), gy.data_ptr(), _ptr(gx),
                                                gw.data_ptr(), gb.data_ptr(), B, K, O, float(negative_slope), _stream()),
                   "lr_linear_bwd_f32")
    return gx, gw, gb


def disp_reg_bwd(disp, gout):
    """Gradient of ops.disp_reg w.r.t. disp (same assumed mermaid stencil; parity unpinned)."""
    disp = _dev(disp, "disp")
    gout = _dev(gout.reshape(()).to(torch.float32), "gout")
    B, _, D, W, H = disp.shape
    g = torch.empty_like(disp)
    with _timed("disp_reg_bwd", bytes=8 * disp.numel()):
        _hip.check(_hip.lib().lr_disp_reg_bwd_f32(disp.data_ptr(), gout.data_ptr(), g.data_ptr(), B, D, W, H, _stream()),
                   "lr_disp_reg_bwd_f32")
    return g
