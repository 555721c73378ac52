"""torch.autograd.Function wrappers: forward and backward both run on the HIP kernels
(liftreg_amd.ops / liftreg_amd.ops_bwd), so the reference's training step
(RegistrationNet.py:389-406: model(input) → loss(output) → total_loss.backward() → optimizer.step())
works unchanged on this package.  Autograd itself is PyTorch plumbing; no ATen compute kernel is used
for any op on the path.
"""
import os

import torch

from . import _hip, ops, ops_bwd


class ConvBlockFn(torch.autograd.Function):
    """LeakyReLU(Conv3d k3 p1 (x) + b) in any of the activation layouts.

    `premasked_grad`: the gradient this block receives already went through this block's LeakyReLU mask (the
    consumer block applied it in its data-gradient epilogue) and is plain NDHWC.  `mask_input_slope`: this block
    is such a consumer — its input is the producer block's LeakyReLU output with that slope.  The model sets the
    two consistently along the encoder chain; a stand-alone convBlock uses neither.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, stride, in_layout, out_layout, slope, packed, premasked_grad=False,
                mask_input_slope=None, mask_out=None, x_sign4=None):
        # mask_out: a uint8 (B,D,W,H,C/4) tensor this block's forward fills with its LeakyReLU sign mask (the encoder's
        # first block in training); x_sign4: the producer's such mask for THIS block's input — the data gradient then
        # takes the producer's mask from one byte per channel quad instead of re-reading 16 bytes
        extra = {} if mask_out is None else {"mask_out": mask_out}
        y = ops.conv3d_k3_lrelu(x, weight, bias, stride, in_layout=in_layout, out_layout=out_layout,
                                negative_slope=slope, packed=packed, **extra)
        ctx.save_for_backward(x, weight, y)
        ctx.x_sign4 = x_sign4
        ctx.cfg = (stride, in_layout, out_layout, slope, bias is not None, premasked_grad, mask_input_slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        stride, in_layout, out_layout, slope, has_bias, premasked, in_slope = ctx.cfg
        need_gx = ctx.needs_input_grad[0]
        if need_gx and (in_layout == _hip.LAYOUT_NCDHW or stride != 2):
            raise NotImplementedError("data gradient is built for the channels-last stride-2 blocks only "
                                      "(the encoder's first block receives data, not activations)")
        gx, gw, gb = ops_bwd.conv3d_bwd(x, in_layout, weight, y, out_layout, gy.contiguous(), out_layout, stride,
                                        slope, need_gx=need_gx, gy_is_gpre=premasked,
                                        mask_input_slope=in_slope if need_gx else None,
                                        x_sign4=ctx.x_sign4 if need_gx else None)
        return gx, gw, (gb if has_bias else None), None, None, None, None, None, None, None, None, None


class ConvPair01Fn(torch.autograd.Function):
    """The encoder's first two blocks (3 -> 16 stride 1, 16 -> 32 stride 2) as ONE autograd node (fp32 training).

    Forward: the two block kernels, the first also writing its LeakyReLU sign mask.  Backward: block 1's weight gradient,
    then ops_bwd.conv3d_dgrad_wgrad0 — block 1's data gradient, block 0's mask and block 0's weight / bias gradient in one
    kernel.  As two nodes (ConvBlockFn twice) the (B,D,W,H,16) gradient between the blocks is written by one kernel and read
    back by the next (8.6 GB each way at 256^3 x 8) although nothing else uses it: the encoder input is data.
    `premasked_grad`: as in ConvBlockFn, for block 1's output (block 2's data gradient applied block 1's mask already)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, slope0, slope1, mid_layout, out_layout, packed0, packed1, premasked_grad, packed_pair=None,
                rest=None):
        """`rest` (with `packed_pair`): `x` is the moving image (B,1,D,W,H) and `rest` the backprojected views (B,P,D,W,H) — the
        encoder input cat([moving, target_volume]) (…Backproj.py:95-98) is read from the two buffers by the forward and by the
        backward, never assembled."""
        B, _, D, W, H = x.shape
        if rest is not None and not (packed_pair is not None and ops.conv3d_pair01_train_supported(x, w0, w1, mid_layout, out_layout, rest) and
                                     ops_bwd.conv3d_dgrad_wgrad0_rest_ok(x, rest)):
            x, rest = torch.cat([x, rest], 1), None          # (shapes the fused kernel does not take: the concatenated input)
        if packed_pair is not None and ops.conv3d_pair01_train_supported(x, w0, w1, mid_layout, out_layout, rest):
            # the forward as the fused pair kernel (exact bf16 operand splits, csrc/conv01_fused.hip) that also writes y0 and mask0
            y1, y0, mask0 = ops.conv3d_pair01_train(x, w0, b0, w1, b1, mid_layout=mid_layout, out_layout=out_layout, slope0=slope0,
                                                    slope1=slope1, packed=packed_pair, rest=rest)
        else:
            mask0 = torch.empty((B, D, W, H, w0.shape[0] // 4), dtype=torch.uint8, device=x.device)
            y0 = ops.conv3d_k3_lrelu(x, w0, b0, 1, in_layout=_hip.LAYOUT_NCDHW, out_layout=mid_layout, negative_slope=slope0,
                                     packed=packed0, mask_out=mask0)
            y1 = ops.conv3d_k3_lrelu(y0, w1, b1, 2, in_layout=mid_layout, out_layout=out_layout, negative_slope=slope1,
                                     packed=packed1)
        ctx.save_for_backward(x, w0, w1, y0, y1, mask0, rest if rest is not None else x.new_empty(0))
        ctx.cfg = (slope0, slope1, mid_layout, out_layout, b0 is not None, b1 is not None, premasked_grad, rest is not None)
        return y1

    @staticmethod
    def backward(ctx, gy1):
        x, w0, w1, y0, y1, mask0, rest = ctx.saved_tensors
        slope0, slope1, mid_layout, out_layout, has_b0, has_b1, premasked, split = ctx.cfg
        gy1 = gy1.contiguous()
        gpre1 = gy1 if premasked else ops_bwd.lrelu_bwd(gy1, out_layout, y1, out_layout, slope1)
        _, gw1, gb1 = ops_bwd.conv3d_bwd(y0, mid_layout, w1, y1, out_layout, gpre1, _hip.LAYOUT_NDHWC, 2, slope1,
                                         need_gx=False, gy_is_gpre=True)
        gw0, gb0 = ops_bwd.conv3d_dgrad_wgrad0(gpre1, w1, mask0, slope0, x, rest if split else None)
        return (None, gw0, gb0 if has_b0 else None, gw1, gb1 if has_b1 else None, None, None, None, None, None, None, None, None, None)


class EncoderBf16Fn(torch.autograd.Function):
    """The six conv blocks of the bf16 variant (conv_dtype="bf16") as ONE autograd node: the activations between the
    blocks are bfloat16 tensors in private layouts, which autograd could not carry gradients for (a gradient must
    have its tensor's dtype), so the chain is walked here.  Forward: lr_conv3d_first_bf16 + lr_conv3d_k3_lrelu_bf16.
    Backward: fp32 gradient math on the bf16-rounded saved activations and weights — exactly the gradient of the
    forward's arithmetic (oracle: ATen autograd of ref_ops.encoder_bf16): LeakyReLU mask of the last block, then per
    block the weight/bias gradient and the data gradient fused with the producer's mask (ops_bwd.conv3d_bwd)."""

    @staticmethod
    def forward(ctx, x, layouts, slopes, strides, packed, grad_bf16, *wb):
        ws, bs = wb[0::2], wb[1::2]
        acts = [x]
        # the first block also leaves the LeakyReLU sign mask of its output (one byte per channel quad): block 1's data
        # gradient reads it instead of the 32-byte bf16 activation (7.2 GB -> 0.9 GB at C5)
        mask0 = None
        if x.shape[1] <= 3 and ws[0].shape[0] % 4 == 0 and not os.environ.get("LIFTREG_BF16_NO_SIGN4"):
            B_, _, D_, W_, H_ = x.shape
            mask0 = torch.empty((B_, D_, W_, H_, ws[0].shape[0] // 4), dtype=torch.uint8, device=x.device)
        for i in range(6):
            lin, lout = layouts[i]
            if i == 0:
                y = ops.conv3d_first_bf16(acts[-1], ws[0], bs[0], out_layout=lout, negative_slope=slopes[0], packed=packed[0],
                                          mask_out=mask0)
            else:
                y = ops.conv3d_k3_lrelu_bf16(acts[-1], ws[i], bs[i], strides[i], in_layout=lin, out_layout=lout,
                                             negative_slope=slopes[i], packed=packed[i])
            acts.append(y)
        ctx.save_for_backward(*acts, *ws, *([mask0] if mask0 is not None else []))
        ctx.cfg = (layouts, slopes, strides, [b is not None for b in bs], grad_bf16, mask0 is not None)
        return acts[-1]

    @staticmethod
    def backward(ctx, gfeat):
        layouts, slopes, strides, has_bias, grad_bf16, has_mask0 = ctx.cfg
        acts, ws = ctx.saved_tensors[:7], ctx.saved_tensors[7:13]
        mask0 = ctx.saved_tensors[13] if has_mask0 else None
        grads = [None] * 12
        g = gfeat.contiguous()
        if grad_bf16:
            # bf16-gradient variant (opt "grad_dtype": "bf16"): the pre-activation gradients between the blocks are
            # rounded to bf16 — the data gradient then runs on the bf16 MFMA and moves half the bytes
            g = ops.cast_bf16(ops_bwd.lrelu_bwd(g, layouts[5][1], acts[6], layouts[5][1], slopes[5]))
            for i in range(5, -1, -1):
                x_layout = _hip.LAYOUT_NCDHW_RBF16 if i == 0 else layouts[i][0]
                g, gw, gb = ops_bwd.conv3d_bwd_bf16g(acts[i], x_layout, ws[i], g, strides[i],
                                                     mask_input_slope=(slopes[i - 1] if i > 0 else None),
                                                     x_sign4=mask0 if i == 1 else None)
                grads[2 * i] = gw
                grads[2 * i + 1] = gb if has_bias[i] else None
            return (None, None, None, None, None, None, *grads)
        for i in range(5, -1, -1):
            lin, lout = layouts[i]
            x_layout = _hip.LAYOUT_NCDHW_RBF16 if i == 0 else lin
            gx, gw, gb = ops_bwd.conv3d_bwd(acts[i], x_layout, ws[i], acts[i + 1], lout, g, lout, strides[i], slopes[i],
                                            need_gx=(i > 0), gy_is_gpre=(i < 5),
                                            mask_input_slope=(slopes[i - 1] if i > 0 else None), round_weights=True,
                                            x_sign4=mask0 if i == 1 else None)
            grads[2 * i] = gw
            grads[2 * i + 1] = gb if has_bias[i] else None
            g = gx
        return (None, None, None, None, None, None, *grads)


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, slope):
        y = ops.linear_lrelu(x, weight, bias, slope)
        ctx.save_for_backward(x, weight, y)
        ctx.cfg = (slope, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        slope, has_bias = ctx.cfg
        gx, gw, gb = ops_bwd.linear_bwd(x, weight, y, gy.contiguous(), slope, need_gx=ctx.needs_input_grad[0])
        return gx, gw, (gb if has_bias else None), None


class PCAFn(torch.autograd.Function):
    """disp = coefs @ basis + mean; only the coefficients receive a gradient (the basis is data)."""

    @staticmethod
    def forward(ctx, coefs, basis_LxM, mean):
        ctx.save_for_backward(basis_LxM)
        return ops.pca_reconstruct(coefs, basis_LxM, mean)

    @staticmethod
    def backward(ctx, gdisp):
        (basis,) = ctx.saved_tensors
        return ops_bwd.pca_bwd_coef(gdisp.contiguous(), basis), None, None


class WarpFn(torch.autograd.Function):
    """(phi, warped) = warp(img, disp): gradient w.r.t. disp only (phi = disp + id passes its gradient through)."""

    @staticmethod
    def forward(ctx, img, disp, id0, id1, id2, seg, using_scale, zero_boundary):
        phi, warped = ops.warp(img, disp, (id0, id1, id2), seg, using_scale=using_scale, zero_boundary=zero_boundary)
        ctx.save_for_backward(img, disp, id0, id1, id2, seg if seg is not None else img.new_empty(0))
        ctx.cfg = (using_scale, zero_boundary, seg is not None)
        ctx.set_materialize_grads(False)
        return phi, warped

    @staticmethod
    def backward(ctx, gphi, gwarped):
        img, disp, id0, id1, id2, seg = ctx.saved_tensors
        using_scale, zero_boundary, has_seg = ctx.cfg
        g = None
        if gwarped is not None:
            g = ops_bwd.warp_bwd_disp(img, disp, (id0, id1, id2), seg if has_seg else None, gwarped.contiguous(),
                                      using_scale=using_scale, zero_boundary=zero_boundary)
        if gphi is not None:
            g = gphi if g is None else g + gphi
        return None, g, None, None, None, None, None, None


class DecodeFn(torch.autograd.Function):
    """The decode half as ONE autograd node: (disp, phi, warped) = warp(img, coefs @ basis + mean).

    Forward: the one-pass kernel (ops.pca_warp) where it applies, else PCA reconstruction + warp — the same bits.
    Backward: the displacement field receives gradient through `warped` (the similarity), through `disp` itself (the
    model returns it as `params`, which the regulariser reads) and possibly through `phi`; their sum is formed inside
    the warp-gradient kernel (`gadd`) instead of by a pass of autograd's own, then projected onto the basis."""

    @staticmethod
    def forward(ctx, coefs, basis_LxM, mean, img, id0, id1, id2, seg, using_scale, target=None):
        """`target` (B,1,D,W,H), single-channel images without a label mask: a 4th output = the similarity's five fp64
        moments of (warped, target) as a DIFFERENTIABLE result of this node — a loss that is a function of them (NCCFn) sends
        its gradient back as (B,5) numbers and the warp-gradient kernel forms d loss / d warped on the fly
        (ops_bwd.warp_bwd_disp_ncc) instead of reading it from a tensor another pass wrote."""
        B, _, D, W, H = img.shape
        if seg is None and ops.pca_warp_supported(coefs, basis_LxM, img):
            disp, phi, warped = ops.pca_warp(coefs, basis_LxM, mean, (id0, id1, id2), img, using_scale=using_scale)
        else:
            disp = ops.pca_reconstruct(coefs, basis_LxM, mean).view(B, 3, D, W, H)
            phi, warped = ops.warp(img, disp, (id0, id1, id2), seg, using_scale=using_scale, zero_boundary=True)
        with_m = target is not None
        ctx.save_for_backward(basis_LxM, img, disp, id0, id1, id2, seg if seg is not None else img.new_empty(0),
                              warped if with_m else img.new_empty(0), target if with_m else img.new_empty(0))
        ctx.cfg = (using_scale, seg is not None, with_m)
        ctx.set_materialize_grads(False)
        if with_m:
            return disp, phi, warped, ops.ncc_moments(warped, target, B)
        return disp, phi, warped

    @staticmethod
    def backward(ctx, gdisp, gphi, gwarped, gmoments=None):
        basis, img, disp, id0, id1, id2, seg, warped, target = ctx.saved_tensors
        using_scale, has_seg, with_m = ctx.cfg
        direct = gdisp if gphi is None else (gphi if gdisp is None else gdisp + gphi)   # phi = disp + id
        g = None if direct is None else direct.contiguous()
        if gwarped is not None:
            g = ops_bwd.warp_bwd_disp(img, disp, (id0, id1, id2), seg if has_seg else None, gwarped.contiguous(),
                                      using_scale=using_scale, zero_boundary=True, gadd=g)
        if with_m and gmoments is not None:        # the similarity's gradient arrives through the moments
            g = ops_bwd.warp_bwd_disp_ncc(img, disp, (id0, id1, id2), warped, target, gmoments, using_scale=using_scale, gadd=g)
        if g is None:
            return (None,) * 10
        return (ops_bwd.pca_bwd_coef(g.contiguous(), basis),) + (None,) * 9


class NCCFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, variant, moments=None):
        n_batch = x.shape[0]
        rows = n_batch if variant == _hip.NCC_CONFIGURED else n_batch * x.shape[1]
        # `moments`: the (rows,5) fp64 moments of exactly (x, y) that the one-pass decode's epilogue produced (model output
        # key "ncc_moments") — handed over explicitly by the caller, never looked up by tensor identity
        m = moments
        if m is None:
            m = ops.ncc_moments(x, y, rows)
        elif tuple(m.shape) != (rows, 5) or m.dtype != torch.float64 or m.device != x.device:
            raise ValueError(f"moments must be a float64 ({rows},5) tensor on {x.device}")
        loss, _ = ops.ncc_loss_from_moments(m, x.numel() // rows, n_batch, variant)
        # moments that require grad are a differentiable function of (x, y) computed by their producer (DecodeFn): the
        # gradient then goes back THROUGH them — (rows,5) numbers — and not to x directly (the producer carries it on)
        via_m = moments is not None and ctx.needs_input_grad[3]
        ctx.save_for_backward(*((m,) if via_m else (x, y, m)))
        ctx.cfg = (variant, x.numel() // rows, via_m)
        return loss

    @staticmethod
    def backward(ctx, gout):
        variant, n, via_m = ctx.cfg
        if via_m:
            (m,) = ctx.saved_tensors
            return None, None, None, ops_bwd.ncc_bwd_moments(m, gout, n, variant)
        x, y, m = ctx.saved_tensors
        return ops_bwd.ncc_bwd(x.contiguous(), y.contiguous(), m, gout, n, variant).view_as(x), None, None, None


class DispRegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp):
        ctx.save_for_backward(disp)
        return ops.disp_reg(disp)

    @staticmethod
    def backward(ctx, gout):
        (disp,) = ctx.saved_tensors
        return ops_bwd.disp_reg_bwd(disp.contiguous(), gout)


class SubspaceRegFn(torch.autograd.Function):
    """The regulariser on the PCA coefficients (ops.subspace_reg): forward and gradient come out of one tiny kernel."""

    @staticmethod
    def forward(ctx, coefs, gram, lin, r0):
        out, gc = ops.subspace_reg(coefs.contiguous(), gram, lin, r0, want_grad=True)
        ctx.save_for_backward(gc)
        return out

    @staticmethod
    def backward(ctx, gout):
        (gc,) = ctx.saved_tensors
        return gc * gout, None, None, None
