"""GPU: a seeded sweep of RANDOM small shapes through every kernel family against the oracles — the fixed-shape parity
tests pin known cases, this one looks for the shapes nobody thought of (odd / tiny / non-multiple-of-4 extents, ragged
tiles, batch and view counts).  Same bars as the parity tests: bit-exact for the gathers, 1e-4 for the fp32 GEMM-like ops."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import ref_ops as ro

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("LIFTREG_FUZZ_CASES", "10"))   # raise for a longer hunt
SEED = int(os.environ.get("LIFTREG_FUZZ_SEED", "0"))       # shifts every family's seed


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


LONG_H = int(os.environ.get("LIFTREG_FUZZ_LONG_H", "0"))   # > 0: rows up to this long (crosses the 64/256-voxel tile edges)


def _shape(rs, lo=2, hi=23):
    d, w, h = (int(v) for v in rs.randint(lo, hi, 3))
    if LONG_H:
        d, w, h = min(d, 6), min(w, 7), int(rs.randint(40, LONG_H + 1))
    return d, w, h


def test_fuzz_projector_and_backprojection(dev):
    from liftreg_amd import ops
    rs = np.random.RandomState(101 + SEED)
    for _ in range(N_CASES):
        D, W, H = _shape(rs, 2, 20)
        P, B = int(rs.randint(1, 5)), int(rs.randint(1, 4))
        Rd, Rh = int(rs.randint(2, 26)), int(rs.randint(2, 70))
        poses = ro.scan_poses(float(rs.uniform(10, 60)), P, W).astype(np.float32)
        sp = rs.uniform(1.0, 3.0, 3).astype(np.float32)
        vol = rs.uniform(0, 0.3, (D, W, H)).astype(np.float32)
        want = co.drr_forward(vol, poses, sp, (Rd, Rh))
        got = ops.drr_forward(T(vol, dev), poses, (Rd, Rh), sp, nseg=1).cpu().numpy()
        assert np.array_equal(got, want), ("drr", D, W, H, P, Rd, Rh)            # one run per ray: the oracle's sum order
        nseg = int(rs.choice([0, 2, 4, 8]))                                       # split rays: partial sums re-associate
        got = ops.drr_forward(T(vol, dev), poses, (Rd, Rh), sp, nseg=nseg).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6, err_msg=str(("drr nseg", nseg, D, W, H, P, Rd, Rh)))
        proj = rs.uniform(-1, 1, (B, P, Rd, Rh)).astype(np.float32)
        bp = ops.backproject(T(proj, dev), poses, (D, W, H)).cpu().numpy()
        assert np.array_equal(bp, co.backproject(proj, poses, (D, W, H))), ("backproject", D, W, H, P, Rd, Rh, B)


def test_fuzz_warp_pca_ncc_reg(dev):
    from liftreg_amd import ops
    rs = np.random.RandomState(102 + SEED)
    for _ in range(N_CASES):
        shape = _shape(rs, 2, 18)
        B, C = int(rs.randint(1, 4)), int(rs.randint(1, 3))
        img = rs.uniform(-1, 1, (B, C) + shape).astype(np.float32)
        disp = rs.normal(0, 0.3, (B, 3) + shape).astype(np.float32)
        tabs = ro.identity_axis_tables(shape)
        zb = bool(rs.randint(0, 2))
        phi, warped = ops.warp(T(img, dev), T(disp, dev), [T(t, dev) for t in tabs], None, using_scale=True, zero_boundary=zb)
        want = ro.warp(torch.from_numpy(img), torch.from_numpy(disp) + ro.identity_map(shape), zero_boundary=zb, using_scale=True)
        np.testing.assert_allclose(warped.cpu().numpy(), want.numpy(), rtol=1e-4, atol=2e-5, err_msg=str(("warp", shape, B, C, zb)))
        # PCA: M must be a multiple of 4 (the ABI's alignment rule)
        Lat, M = int(rs.randint(1, 20)), int(rs.randint(1, 300)) * 4
        basis, mean = rs.normal(0, 0.05, (Lat, M)).astype(np.float32), rs.normal(0, 0.01, M).astype(np.float32)
        coefs = rs.normal(0, 1, (B, Lat)).astype(np.float32)
        got = ops.pca_reconstruct(T(coefs, dev), T(basis, dev), T(mean, dev)).cpu().numpy()
        np.testing.assert_allclose(got, coefs.astype(np.float64) @ basis + mean, rtol=1e-4, atol=1e-5, err_msg=str(("pca", B, Lat, M)))
        # one-pass decode (PCA + identity + warp) == the two kernels, wherever it applies (rows of 4k voxels)
        V3 = 3 * int(np.prod(shape))
        Lw = int(rs.randint(1, 12))
        bw = T(rs.normal(0, 0.1, (Lw, V3)).astype(np.float32), dev)
        if rs.randint(0, 2):
            bw = bw.to(torch.bfloat16)
        mw, cw = T(rs.normal(0, 0.02, V3).astype(np.float32), dev), T(rs.normal(0, 1, (B, Lw)).astype(np.float32), dev)
        if ops.pca_warp_supported(cw, bw, T(img, dev)):
            d1, p1, w1 = ops.pca_warp(cw, bw, mw, [T(t, dev) for t in tabs], T(img, dev))
            d2 = ops.pca_reconstruct(cw, bw, mw).view(B, 3, *shape)
            p2, w2 = ops.warp(T(img, dev), d2, [T(t, dev) for t in tabs], None)
            assert torch.equal(d1, d2) and torch.equal(p1, p2) and torch.equal(w1, w2), ("pca_warp", shape, B, C, Lw, bw.dtype)
        else:
            assert shape[2] % 4 != 0
        # NCC (both variants) and the regulariser
        x, y = rs.uniform(-1, 1, (B, C) + shape).astype(np.float32), rs.uniform(-1, 1, (B, C) + shape).astype(np.float32)
        assert abs(float(ops.ncc_loss(T(x, dev), T(y, dev), 0)) - float(ro.ncc_loss(torch.from_numpy(x), torch.from_numpy(y)))) < 2e-5
        assert abs(float(ops.ncc_loss(T(x, dev), T(y, dev), 1)) - float(ro.ncc_loss_squared(torch.from_numpy(x), torch.from_numpy(y)))) < 2e-5
        r = float(ops.disp_reg(T(disp, dev)))
        rw = float(ro.disp_reg(torch.from_numpy(disp)))
        assert abs(r - rw) <= 1e-4 * max(1.0, abs(rw)), ("reg", shape, r, rw)


def test_fuzz_conv_forward_and_backward(dev):
    """Random extents for the first block (planar) and the stride-2 blocks (both channels-last layouts), forward and —
    through the chained backward — weight, bias and data gradients."""
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(103 + SEED)

    def to_layout(t_ncdhw, lay):
        cl = t_ncdhw.permute(0, 2, 3, 4, 1).contiguous()
        if lay == ops.LAYOUT_NDHWC_HPS:
            B, D, W, H, C = cl.shape
            h = torch.arange(H, device=cl.device)
            inv = torch.empty(H, dtype=torch.long, device=cl.device)
            inv[(h & 1) * (H // 2) + (h >> 1)] = h
            cl = cl.reshape(B, D, W, H, C // 16, 16)[:, :, :, inv].permute(0, 1, 2, 4, 3, 5).reshape(B, D, W, H, C).contiguous()
        return cl

    for case in range(N_CASES):
        first = case % 3 == 0
        shape, B = _shape(rs, 2, 15), int(rs.randint(1, 3))
        if first:
            cin, cout, s = int(rs.choice([1, 2, 3, 5, 12])), 16, 1
            xl = ops.LAYOUT_NCDHW
        else:
            cin, cout, s = int(rs.choice([16, 32])), int(rs.choice([16, 32])), 2
            xl = ops.LAYOUT_NDHWC_HPS if shape[2] % 2 == 0 and rs.randint(0, 2) else ops.LAYOUT_NDHWC
        ho = (shape[2] - 1) // s + 1
        yl = ops.LAYOUT_NDHWC_HPS if ho % 2 == 0 and rs.randint(0, 2) else ops.LAYOUT_NDHWC
        x = rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32)
        w = (rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)
        b = rs.uniform(-0.1, 0.1, cout).astype(np.float32)
        xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b))
        yref = ro.conv_block(xt, wt, bt, s)
        gy = rs.normal(0, 1, tuple(yref.shape)).astype(np.float32)
        xd = T(x, dev) if first else to_layout(T(x, dev), xl)
        yd = ops.conv3d_k3_lrelu(xd, T(w, dev), T(b, dev), s, in_layout=xl, out_layout=yl)
        yplain = ops.hps_to_ndhwc(yd) if yl == ops.LAYOUT_NDHWC_HPS else yd
        tag = str((cin, cout, s, shape, B, xl, yl))
        ygpu = yplain.permute(0, 4, 1, 2, 3).cpu()
        np.testing.assert_allclose(ygpu.numpy(), yref.detach().numpy(), rtol=1e-4, atol=1e-5, err_msg="fwd " + tag)
        if not first and xl == ops.LAYOUT_NDHWC_HPS:
            # the persistent Winograd rows kernel (default only for planes of >= 64 x 64 outputs) forced onto this small case,
            # every output layout it has, and as a z-slab with the other plane parity
            os.environ["LIFTREG_CONV_ROWS_ALWAYS"] = "1"
            try:
                for lay2 in (ops.LAYOUT_NDHWC, ops.LAYOUT_NCDHW) + ((ops.LAYOUT_NDHWC_HPS,) if ho % 2 == 0 else ()):
                    y2 = ops.conv3d_k3_lrelu(xd, T(w, dev), T(b, dev), s, in_layout=xl, out_layout=lay2)
                    y2 = y2 if lay2 == ops.LAYOUT_NCDHW else (ops.hps_to_ndhwc(y2) if lay2 == ops.LAYOUT_NDHWC_HPS else y2).permute(0, 4, 1, 2, 3)
                    np.testing.assert_allclose(y2.cpu().numpy(), yref.detach().numpy(), rtol=1e-4, atol=1e-5, err_msg="rows " + tag)
                if shape[0] >= 5:
                    full = ops.conv3d_k3_lrelu(xd, T(w, dev), T(b, dev), s, in_layout=xl, out_layout=ops.LAYOUT_NDHWC)
                    slab = ops.conv3d_k3_lrelu(xd[:, 2:].contiguous(), T(w, dev), T(b, dev), s, in_layout=xl,
                                               out_layout=ops.LAYOUT_NDHWC, z_phase=1)
                    assert torch.equal(slab[:, 1:], full[:, 2:]), "z_phase " + tag
            finally:
                del os.environ["LIFTREG_CONV_ROWS_ALWAYS"]
        # reference gradients through the LeakyReLU mask of the GPU's own output: a pre-activation within rounding
        # distance of 0 may sit on the other side on the CPU, and ONE flipped mask element moves the 27*Cin weight
        # gradients of its output channel by O(1) — that is a property of the comparison, not of the kernels
        mask = torch.where(ygpu > 0, torch.ones(()), torch.full((), 0.2))
        torch.nn.functional.conv3d(xt, wt, bt, stride=s, padding=1).backward(torch.from_numpy(gy) * mask)
        gx, gw, gb = ops_bwd.conv3d_bwd(xd, xl, T(w, dev), yd, yl, to_layout(T(gy, dev), ops.LAYOUT_NDHWC), ops.LAYOUT_NDHWC, s,
                                        need_gx=not first, nblk=int(rs.choice([1, 8, 64])))
        # sums over thousands of voxels: the absolute tolerance follows the gradient's own scale
        np.testing.assert_allclose(gw.cpu().numpy(), wt.grad.numpy(), rtol=3e-4, atol=3e-5 * max(1.0, float(wt.grad.abs().max())),
                                   err_msg="gw " + tag)
        np.testing.assert_allclose(gb.cpu().numpy(), bt.grad.numpy(), rtol=3e-4, atol=3e-5 * max(1.0, float(bt.grad.abs().max())),
                                   err_msg="gb " + tag)
        if not first:
            gxp = ops.hps_to_ndhwc(gx) if xl == ops.LAYOUT_NDHWC_HPS else gx
            np.testing.assert_allclose(gxp.permute(0, 4, 1, 2, 3).cpu().numpy(), xt.grad.numpy(), rtol=3e-4, atol=3e-5,
                                       err_msg="gx " + tag)


def test_fuzz_bf16_blocks(dev):
    from liftreg_amd import ops
    rs = np.random.RandomState(104 + SEED)
    for case in range(N_CASES):
        shape, B = _shape(rs, 2, 15), int(rs.randint(1, 3))
        if case % 3 == 0:
            cin, cout = int(rs.choice([1, 3, 4, 12])), int(rs.choice([16, 32]))
            x = torch.from_numpy(rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32))
            w = torch.from_numpy((rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
            b = torch.from_numpy(rs.uniform(-0.1, 0.1, cout).astype(np.float32))
            ol = ops.LAYOUT_BF16_NDHWC_HPS if shape[2] % 2 == 0 else ops.LAYOUT_BF16_NDHWC
            y = ops.conv3d_first_bf16(x.to(dev), w.to(dev), b.to(dev), out_layout=ol)
            want = ro.conv_block_bf16(x, w, b, 1)
        else:
            cin, cout = int(rs.choice([16, 32])), int(rs.choice([16, 32]))
            x = torch.from_numpy(rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32)).to(torch.bfloat16)
            w = torch.from_numpy((rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
            b = torch.from_numpy(rs.uniform(-0.1, 0.1, cout).astype(np.float32))
            il = ops.LAYOUT_BF16_NDHWC_HPS if shape[2] % 2 == 0 and rs.randint(0, 2) else ops.LAYOUT_BF16_NDHWC
            ho = (shape[2] - 1) // 2 + 1
            ol = ops.LAYOUT_BF16_NDHWC_HPS if ho % 2 == 0 else ops.LAYOUT_BF16_NDHWC
            xd = x.permute(0, 2, 3, 4, 1).contiguous().to(dev)
            if il == ops.LAYOUT_BF16_NDHWC_HPS:
                H = xd.shape[3]
                h = torch.arange(H, device=dev)
                inv = torch.empty(H, dtype=torch.long, device=dev)
                inv[(h & 1) * (H // 2) + (h >> 1)] = h
                xd = xd[:, :, :, inv].contiguous()
            y = ops.conv3d_k3_lrelu_bf16(xd, w.to(dev), b.to(dev), 2, in_layout=il, out_layout=ol)
            want = ro.conv_block_bf16(x.float(), w, b, 2)
        if ol == ops.LAYOUT_BF16_NDHWC_HPS:
            y = ops.bf16_hps_to_ndhwc(y)
        got = y.float().permute(0, 4, 1, 2, 3).cpu().numpy()
        flips = int((got != want.numpy()).sum())      # fp32 summation order may flip a final bf16 rounding here and there
        assert flips <= max(2, 0.01 * got.size), (cin, cout, shape, B, flips)
        np.testing.assert_allclose(got, want.numpy(), rtol=2.0 ** -7, atol=1e-6, err_msg=str((cin, cout, shape, B)))


def test_fuzz_bf16_gradient_backward(dev):
    """The bf16-gradient backward kernels (transposing LDS reads, shifted window copies) on random extents."""
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(105 + SEED)
    for case in range(N_CASES):
        B = int(rs.randint(1, 3))
        if case % 3 == 0:      # first block: planar fp32 input; the fast path needs H % 4 == 0
            cin, cout = int(rs.choice([1, 3, 5, 12])), 16
            shape = (int(rs.randint(2, 10)), int(rs.randint(2, 10)), int(rs.randint(1, 36)) * 4)
            x = torch.from_numpy(rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32))
            w = torch.from_numpy((rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
            g = torch.from_numpy(rs.normal(0, 1, (B, cout) + shape).astype(np.float32)).to(torch.bfloat16)
            wr, b0 = w.clone().requires_grad_(True), torch.zeros(cout, requires_grad=True)
            torch.nn.functional.conv3d(ro._bf16(x), ro._bf16(wr), b0, stride=1, padding=1).backward(g.float())
            gx, gw, gb = ops_bwd.conv3d_bwd_bf16g(x.to(dev), ops.LAYOUT_NCDHW_RBF16, w.to(dev),
                                                  g.permute(0, 2, 3, 4, 1).contiguous().to(dev), 1, nblk=int(rs.choice([1, 8, 64])))
            assert gx is None
        else:
            cin, cout = int(rs.choice([16, 32])), 32
            shape = _shape(rs, 2, 15)
            xl = ops.LAYOUT_BF16_NDHWC_HPS if shape[2] % 2 == 0 and rs.randint(0, 2) else ops.LAYOUT_BF16_NDHWC
            x = torch.from_numpy(rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32)).to(torch.bfloat16)
            w = torch.from_numpy((rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
            osz = tuple((n - 1) // 2 + 1 for n in shape)
            g = torch.from_numpy(rs.normal(0, 1, (B, cout) + osz).astype(np.float32)).to(torch.bfloat16)
            xr, wr, b0 = x.float().requires_grad_(True), w.clone().requires_grad_(True), torch.zeros(cout, requires_grad=True)
            torch.nn.functional.conv3d(xr, ro._bf16(wr), b0, stride=2, padding=1).backward(g.float())
            want_gx = ro._bf16(torch.where(x.float() > 0, xr.grad, 0.25 * xr.grad))
            xd = x.permute(0, 2, 3, 4, 1).contiguous().to(dev)
            if xl == ops.LAYOUT_BF16_NDHWC_HPS:
                H = xd.shape[3]
                h = torch.arange(H, device=dev)
                inv = torch.empty(H, dtype=torch.long, device=dev)
                inv[(h & 1) * (H // 2) + (h >> 1)] = h
                xd = xd[:, :, :, inv].contiguous()
            gx, gw, gb = ops_bwd.conv3d_bwd_bf16g(xd, xl, w.to(dev), g.permute(0, 2, 3, 4, 1).contiguous().to(dev), 2,
                                                  mask_input_slope=0.25, nblk=int(rs.choice([1, 8, 64])))
            got = gx.float().permute(0, 4, 1, 2, 3).cpu().numpy()
            flips = int((got != want_gx.numpy()).sum())
            assert flips <= max(2, 0.01 * got.size), (cin, shape, B, flips)
            np.testing.assert_allclose(got, want_gx.numpy(), rtol=2.0 ** -7, atol=1e-6, err_msg=str((cin, shape, B)))
        tag = str((cin, cout, shape, B))
        np.testing.assert_allclose(gw.cpu().numpy(), wr.grad.numpy(), rtol=3e-4, atol=4e-4 * max(1.0, float(wr.grad.abs().max()) / 10),
                                   err_msg="gw " + tag)
        np.testing.assert_allclose(gb.cpu().numpy(), b0.grad.numpy(), rtol=3e-4, atol=4e-4 * max(1.0, float(b0.grad.abs().max()) / 10),
                                   err_msg="gb " + tag)


def test_fuzz_whole_model(dev):
    """The plugin model end to end (fp32) on random non-cubic sizes, view counts, latent sizes, batch sizes, with and
    without label masks, against the CPU oracle — forward outputs and, every other case, all parameter gradients."""
    from liftreg_amd.losses.SubspaceLoss import loss as SubspaceLoss
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    rs = np.random.RandomState(106 + SEED)
    opt = {"initial_reg_factor": 0.01, "min_reg_factor": 0.01, "reg_factor_decay_from": 2}
    for case in range(max(2, N_CASES // 2)):
        shape = tuple(int(v) for v in rs.randint(17, 41, 3))
        P, L, B = int(rs.randint(1, 5)), int(rs.randint(2, 12)), int(rs.choice([1, 2, 3, 9, 13]))   # > 8: batch-chunked PCA paths
        Rd, Rh = int(rs.randint(8, 40)), int(rs.randint(8, 40))
        labels = bool(rs.randint(0, 2))
        torch.manual_seed(1000 + case)
        net = model(list(shape), {"drr_feature_num": P, "latent_dim": L, "pca_path": f"synthetic:{case}"}).to(dev)
        poses = ro.scan_poses(float(rs.uniform(15, 45)), P, shape[1]).astype(np.float32)
        inp = {"source": torch.from_numpy(rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32)),
               "target": torch.from_numpy(rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32)),
               "target_proj": torch.from_numpy(rs.uniform(-1, 1, (B, P, Rd, Rh)).astype(np.float32)),
               "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}
        if labels:
            inp["source_label"] = torch.from_numpy((rs.uniform(0, 1, (B, 1) + shape) > 0.3).astype(np.float32))
            inp["target_label"] = torch.from_numpy((rs.uniform(0, 1, (B, 1) + shape) > 0.3).astype(np.float32))
        dinp = {k: v.to(dev) if k != "target_poses" else v for k, v in inp.items()}
        train = case % 2 == 1
        tag = str((shape, P, L, B, Rd, Rh, labels, train))
        if train:
            net.train()
            out = net(dinp)
            out["epoch"] = 0
            SubspaceLoss(dict(opt))(out)["total_loss"].backward()
            params = {k: v.detach().cpu().clone().requires_grad_("gaussian" not in k) for k, v in net.state_dict().items()}
        else:
            net.eval()
            with torch.no_grad():
                out = net(dinp)
            params = {k: v.cpu() for k, v in net.state_dict().items()}
        ref = ro.model_forward(params, inp, net.pca_vectors_LxM.cpu(), net.pca_mean.cpu())
        for k in ("pca_coefs", "params", "phi", "warped", "target"):
            np.testing.assert_allclose(out[k].detach().cpu().numpy(), ref[k].detach().numpy(), rtol=1e-4, atol=6e-5, err_msg=k + tag)
        if train:
            ro.subspace_loss(ref, 0, **opt)["total_loss"].backward()
            for k, p in net.named_parameters():
                g, w = p.grad.cpu().numpy().astype(np.float64), params[k].grad.numpy().astype(np.float64)
                if np.abs(g - w).max() <= 2e-3 * max(np.abs(w).max(), 1e-12):
                    continue
                # A pre-activation within rounding distance of 0 can land on different sides of the LeakyReLU on the
                # two devices (a few of the millions of activations at the larger batches: measured, the GPU and the
                # CPU gradient are then equally far from a float64 run).  One flipped mask element shifts every
                # upstream weight gradient by up to a few % of its maximum — but not its direction or size, which is
                # what a structural error (a dropped sample, chunk, tap or channel block) changes.
                cos = float((g * w).sum() / max(np.sqrt((g * g).sum() * (w * w).sum()), 1e-300))
                ratio = float(np.sqrt((g * g).sum() / max((w * w).sum(), 1e-300)))
                assert cos >= 0.998 and abs(ratio - 1.0) <= 3e-2, (k, tag, cos, ratio)


def test_fuzz_bf16_model_modes(dev):
    """conv_dtype / grad_dtype / pca_dtype = bf16 on random (odd, non-multiple-of-4) sizes: the forward against the CPU
    restatement of the bf16 contract, and a training step in every mode runs and yields finite gradients close to the
    fp32-gradient mode's (the per-kernel contracts are pinned elsewhere; this hunts for unsupported-shape failures)."""
    from liftreg_amd.losses.SubspaceLoss import loss as SubspaceLoss
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    rs = np.random.RandomState(107 + SEED)
    opt = {"initial_reg_factor": 0.01, "min_reg_factor": 0.01, "reg_factor_decay_from": 2}
    for case in range(max(2, N_CASES // 2)):
        shape = tuple(int(v) for v in rs.randint(17, 37, 3))
        P, L, B = int(rs.randint(1, 4)), int(rs.randint(2, 9)), int(rs.randint(1, 3))
        poses = ro.scan_poses(30.0, P, shape[1]).astype(np.float32)
        inp = {"source": torch.from_numpy(rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32)).to(dev),
               "target": torch.from_numpy(rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32)).to(dev),
               "target_proj": torch.from_numpy(rs.uniform(-1, 1, (B, P, 24, 28)).astype(np.float32)).to(dev),
               "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}
        grads = {}
        for gd, pd in (("fp32", "fp32"), ("bf16", "fp32"), ("bf16", "bf16")):
            torch.manual_seed(500 + case)
            net = model(list(shape), {"drr_feature_num": P, "latent_dim": L, "pca_path": f"synthetic:{case}", "conv_dtype": "bf16",
                                      "grad_dtype": gd, "pca_dtype": pd}).to(dev).train()
            out = net(inp)
            out["epoch"] = 0
            SubspaceLoss(dict(opt))(out)["total_loss"].backward()
            g = torch.cat([p.grad.flatten() for p in net.parameters()])
            assert torch.isfinite(g).all(), (shape, gd, pd)
            grads[(gd, pd)] = g
            if (gd, pd) == ("fp32", "fp32"):       # forward against the restatement of the bf16 contract
                sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
                ref = ro.model_forward(sd, {k: v.cpu() for k, v in inp.items()}, net.pca_vectors_LxM.float().cpu(), net.pca_mean.cpu(),
                                       conv_dtype="bf16")
                scale = float(ref["pca_coefs"].abs().max())
                assert float((out["pca_coefs"].detach().cpu() - ref["pca_coefs"]).abs().max()) <= 5e-3 * scale, shape
        base = grads[("fp32", "fp32")]
        for key in (("bf16", "fp32"), ("bf16", "bf16")):
            cos = float(torch.dot(grads[key], base) / (grads[key].norm() * base.norm()))
            # the rounded basis is a (0.4 % per entry) different deformation model, not a rounding of the same one:
            # its gradient is compared for direction only (the kernels themselves are pinned bit-exact against the
            # fp32 kernels fed the rounded basis in test_gpu_bf16.py)
            assert cos > (0.85 if key[1] == "bf16" else 0.99), (shape, key, cos)


def test_fuzz_slab_sharded_forward(dev):
    """z-slab sharded forward == unsharded forward, bit for bit, on random W/H (odd, non-cubic), batch and view counts,
    fp32 and bf16 activations, with and without label masks (D must stay a multiple of 32·world)."""
    from liftreg_amd import parallel as par
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    rs = np.random.RandomState(108 + SEED)
    for case in range(max(2, N_CASES // 3)):
        world = int(rs.choice([2, 4]))
        D = 32 * world * int(rs.randint(1, 3)) if world == 2 else 128
        W, H = int(rs.randint(17, 40)), int(rs.randint(17, 40))
        P, L, B = int(rs.randint(1, 4)), int(rs.randint(2, 7)), int(rs.randint(1, 3))
        cd = str(rs.choice(["fp32", "bf16"]))
        torch.manual_seed(case)
        net = model([D, W, H], {"drr_feature_num": P, "latent_dim": L, "pca_path": f"synthetic:{case}", "conv_dtype": cd}).to(dev).eval()
        poses = ro.scan_poses(30.0, P, W).astype(np.float32)
        inp = {"source": torch.from_numpy(rs.uniform(-1, 1, (B, 1, D, W, H)).astype(np.float32)).to(dev),
               "target": torch.from_numpy(rs.uniform(-1, 1, (B, 1, D, W, H)).astype(np.float32)).to(dev),
               "target_proj": torch.from_numpy(rs.uniform(-1, 1, (B, P, 20, 24)).astype(np.float32)).to(dev),
               "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}
        if rs.randint(0, 2):
            inp["source_label"] = torch.from_numpy((rs.uniform(0, 1, (B, 1, D, W, H)) > 0.3).astype(np.float32)).to(dev)
            inp["target_label"] = torch.from_numpy((rs.uniform(0, 1, (B, 1, D, W, H)) > 0.3).astype(np.float32)).to(dev)
        with torch.no_grad():
            ref = net(inp)
            outs = par.SlabShardedRegistration(net, par.LocalComm(world)).forward([inp] * world)
        tag = str((world, D, W, H, P, L, B, cd, "source_label" in inp))
        for r, o in enumerate(outs):
            d0, d1 = par.slab_bounds(D, world, r)
            assert torch.equal(o["pca_coefs"], ref["pca_coefs"]), "coefs " + tag
            assert torch.equal(o["params"], ref["params"][:, :, d0:d1]), "params " + tag
            assert torch.equal(o["warped"], ref["warped"][:, :, d0:d1]), "warped " + tag


def test_fuzz_warp_variants_and_slabs(dev):
    """Warp modes (nearest / border / no scaling / label mask), slab rows [d0,d1) of warp and DRR, on random extents."""
    from liftreg_amd import ops
    rs = np.random.RandomState(109 + SEED)
    for _ in range(N_CASES):
        shape = _shape(rs, 2, 16)
        D = shape[0]
        B, C = int(rs.randint(1, 3)), int(rs.randint(1, 3))
        img = rs.uniform(-1, 1, (B, C) + shape).astype(np.float32)
        seg = (rs.uniform(0, 1, (B, C) + shape) > 0.4).astype(np.float32) if rs.randint(0, 2) else None
        disp = rs.normal(0, 0.35, (B, 3) + shape).astype(np.float32)
        tabs = ro.identity_axis_tables(shape)
        flags = (co.USING_SCALE if rs.randint(0, 2) else 0) | (co.BORDER if rs.randint(0, 2) else 0) | (co.NEAREST if rs.randint(0, 3) == 0 else 0)
        want_phi, want = co.warp(img, disp, tabs, seg, flags=flags)
        kw = dict(using_scale=bool(flags & co.USING_SCALE), zero_boundary=not (flags & co.BORDER),
                  mode="nearest" if flags & co.NEAREST else "bilinear")
        ids = [T(t, dev) for t in tabs]
        phi, warped = ops.warp(T(img, dev), T(disp, dev), ids, None if seg is None else T(seg, dev), **kw)
        assert np.array_equal(phi.cpu().numpy(), want_phi) and np.array_equal(warped.cpu().numpy(), want), ("warp", shape, flags)
        if D >= 3:       # a slab of rows is the same rows of the whole
            d0 = int(rs.randint(0, D - 1)); d1 = int(rs.randint(d0 + 1, D + 1))
            ids_s = [T(tabs[0][d0:d1], dev), ids[1], ids[2]]
            _, ws = ops.warp(T(img, dev), T(disp[:, :, d0:d1], dev), ids_s, None if seg is None else T(seg, dev), d0=d0, d1=d1, **kw)
            assert np.array_equal(ws.cpu().numpy(), want[:, :, d0:d1]), ("warp slab", shape, d0, d1, flags)
        # DRR: slab partial images add up to the whole
        P, Rd, Rh = int(rs.randint(1, 4)), int(rs.randint(3, 20)), int(rs.randint(3, 40))
        poses = ro.scan_poses(30.0, P, shape[1]).astype(np.float32)
        sp = np.array((2.2, 2.2, 2.2), np.float32)
        vol = rs.uniform(0, 0.3, shape).astype(np.float32)
        full = ops.drr_forward(T(vol, dev), poses, (Rd, Rh), sp, nseg=1)
        if D >= 2:
            cut = int(rs.randint(1, D))
            a = ops.drr_forward(T(vol[:cut], dev), poses, (Rd, Rh), sp, d0=0, d1=cut, full_D=D, nseg=1)
            b = ops.drr_forward(T(vol[cut:], dev), poses, (Rd, Rh), sp, d0=cut, d1=D, full_D=D, nseg=1)
            np.testing.assert_allclose((a + b).cpu().numpy(), full.cpu().numpy(), rtol=2e-5, atol=1e-6, err_msg=str(("drr slabs", shape, cut)))
        # HU input (the HU -> mu conversion folded into the tap loads), cut into slabs incl. ONE-plane slabs (the fast
        # kernel's z-edge test wrapped for a 1-plane slab and summed converted zeros, 0.2 each): every slab = the mu-input slab of
        # the converted volume bit for bit, and the slabs add up to the whole
        hu = rs.uniform(-1100, 400, shape).astype(np.float32)
        flip = bool(rs.randint(0, 2))
        mu = ops.hu_to_mu(T(hu, dev))
        full_hu = ops.drr_forward(T(hu, dev), poses, (Rd, Rh), sp, nseg=1, hu_input=True, flip_w=flip)
        assert torch.equal(full_hu, ops.drr_forward(mu, poses, (Rd, Rh), sp, nseg=1, flip_w=flip))
        tot = torch.zeros_like(full_hu)
        cuts = sorted({0, D, int(rs.randint(0, D + 1)), min(D, int(rs.randint(0, D + 1)) + 1)} | ({1, D - 1} if D >= 2 else set()))
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            part = ops.drr_forward(T(hu[lo:hi], dev), poses, (Rd, Rh), sp, d0=lo, d1=hi, full_D=D, nseg=1, hu_input=True, flip_w=flip)
            want_p = ops.drr_forward(mu[lo:hi].contiguous(), poses, (Rd, Rh), sp, d0=lo, d1=hi, full_D=D, nseg=1, flip_w=flip)
            assert torch.equal(part, want_p), ("drr HU slab", shape, lo, hi, flip)
            tot += part
        np.testing.assert_allclose(tot.cpu().numpy(), full_hu.cpu().numpy(), rtol=2e-5, atol=1e-6, err_msg=str(("drr HU slabs", shape, cuts)))


def test_fuzz_small_backward_ops(dev):
    """NCC / warp / PCA / Linear / regulariser gradients on random extents against ATen autograd of the oracle."""
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(110 + SEED)
    for _ in range(N_CASES):
        shape = _shape(rs, 2, 14)
        B, C = int(rs.randint(1, 3)), int(rs.randint(1, 3))
        # NCC
        for variant, f in ((0, ro.ncc_loss), (1, ro.ncc_loss_squared)):
            x = rs.uniform(-1, 1, (B, C) + shape).astype(np.float32)
            y = (0.5 * x + 0.5 * rs.uniform(-1, 1, x.shape)).astype(np.float32)
            xt = torch.from_numpy(x).requires_grad_(True)
            (f(xt, torch.from_numpy(y)) * 1.3).backward()
            R = B if variant == 0 else B * C
            m = ops.ncc_moments(T(x, dev), T(y, dev), R)
            gx = ops_bwd.ncc_bwd(T(x, dev), T(y, dev), m, torch.tensor(1.3, device=dev), x.size // R, variant)
            np.testing.assert_allclose(gx.cpu().numpy(), xt.grad.numpy(), rtol=3e-4, atol=1e-7, err_msg=str(("ncc", variant, shape)))
        # warp w.r.t. displacement
        img = rs.uniform(-1, 1, (B, C) + shape).astype(np.float32)
        disp = rs.normal(0, 0.25, (B, 3) + shape).astype(np.float32)
        gw = rs.normal(0, 1, (B, C) + shape).astype(np.float32)
        zb = bool(rs.randint(0, 2))
        d = torch.from_numpy(disp).requires_grad_(True)
        ro.warp(torch.from_numpy(img), d + ro.identity_map(shape), zero_boundary=zb, using_scale=True).backward(torch.from_numpy(gw))
        got = ops_bwd.warp_bwd_disp(T(img, dev), T(disp, dev), [T(t, dev) for t in ro.identity_axis_tables(shape)], None, T(gw, dev),
                                    using_scale=True, zero_boundary=zb)
        # the gradient carries the un-normalisation factor (size-1)/2 of its axis: fp32 cancellation scales with it
        np.testing.assert_allclose(got.cpu().numpy(), d.grad.numpy(), rtol=2e-4, atol=max(3e-5, 1.5e-6 * max(shape)),
                                   err_msg=str(("warp_bwd", shape, zb)))
        # regulariser
        d2 = torch.from_numpy(disp).requires_grad_(True)
        (ro.disp_reg(d2) * 0.7).backward()
        gr = ops_bwd.disp_reg_bwd(T(disp, dev), torch.tensor(0.7, device=dev))
        np.testing.assert_allclose(gr.cpu().numpy(), d2.grad.numpy(), rtol=2e-4, atol=1e-7, err_msg=str(("reg_bwd", shape)))
        # PCA coefficients (any M now) and Linear
        Lat, M = int(rs.randint(1, 15)), int(rs.randint(5, 900))
        gd, basis = rs.normal(0, 1, (B, M)).astype(np.float32), rs.normal(0, 0.05, (Lat, M)).astype(np.float32)
        np.testing.assert_allclose(ops_bwd.pca_bwd_coef(T(gd, dev), T(basis, dev)).cpu().numpy(), gd.astype(np.float64) @ basis.T,
                                   rtol=2e-4, atol=1e-5, err_msg=str(("pca_bwd", B, Lat, M)))
        K, O = int(rs.randint(3, 300)), int(rs.randint(1, 40))
        slope = float(rs.choice([0.2, 1.0]))
        xl = rs.uniform(-1, 1, (B, K)).astype(np.float32)
        wl = (rs.normal(0, 1, (O, K)) / np.sqrt(K)).astype(np.float32)
        bl = rs.uniform(-0.1, 0.1, O).astype(np.float32)
        gyl = rs.normal(0, 1, (B, O)).astype(np.float32)
        xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (xl, wl, bl))
        ro.fc_block(xt, wt, bt, None if slope == 1.0 else slope).backward(torch.from_numpy(gyl))
        yl = ops.linear_lrelu(T(xl, dev), T(wl, dev), T(bl, dev), slope)
        gxl, gwl, gbl = ops_bwd.linear_bwd(T(xl, dev), T(wl, dev), yl, T(gyl, dev), slope)
        for got_, want_, nm in ((gxl, xt.grad, "gx"), (gwl, wt.grad, "gw"), (gbl, bt.grad, "gb")):
            np.testing.assert_allclose(got_.cpu().numpy(), want_.numpy(), rtol=2e-4, atol=2e-5, err_msg=str(("linear", nm, B, K, O)))


def test_fuzz_round2_kernels(dev, monkeypatch):
    """Random shapes through the kernels added in round 2, each against the kernel pair / kernel it must equal bit for
    bit: the first block with the backprojection fused into it (conv0_pc.hip), the producer/consumer first block on plain
    inputs, the slab form of the one-pass decode with the NCC moments in its epilogue, and the sign-mask data gradient."""
    from liftreg_amd import ops, ops_bwd
    from liftreg_amd.utils import net_utils as N
    rs = np.random.RandomState(909 + SEED)
    for case in range(N_CASES):
        # ---- first block: fused backprojection == backproject + first_split; PC kernel == single-buffer kernel
        D, W = int(rs.randint(2, 14)), int(rs.randint(2, 14))
        H = 4 * int(rs.randint(1, 36 if not LONG_H else LONG_H // 4 + 1))
        B, P = int(rs.randint(1, 4)), int(rs.randint(1, 3))
        Pw, Ph = int(rs.randint(2, 40)), int(rs.randint(2, 90))
        moving = T(rs.uniform(-1, 1, (B, 1, D, W, H)).astype(np.float32), dev)
        proj = T(rs.uniform(-1, 1, (B, P, Pw, Ph)).astype(np.float32), dev)
        if rs.rand() < 0.5:
            poses = ro.scan_poses(float(rs.uniform(10, 60)), P, W).astype(np.float32)
        else:
            poses = (rs.uniform(-2.5, 2.5, (P, 3)) * np.array([1, 0, 1]) + np.array([0, rs.uniform(1.1, 4.0), 0])).astype(np.float32) * W
        w0 = T(rs.normal(0, 0.3, (16, P + 1, 3, 3, 3)).astype(np.float32), dev)
        b0 = T(rs.normal(0, 0.1, 16).astype(np.float32), dev)
        lay = ops.LAYOUT_NDHWC_HPS if (H % 2 == 0 and rs.rand() < 0.7) else ops.LAYOUT_NDHWC
        tv = ops.backproject(proj, poses, (D, W, H))
        want = ops.conv3d_first_split(moving, tv, w0, b0, out_layout=lay)
        from liftreg_amd import _hip
        if _hip.has_experimental():      # (conv0_pc.hip: the experimental build only — make exp + LIFTREG_HIP_LIB)
            got = ops.conv3d_first_fused_bp(moving, proj, poses, w0, b0, out_layout=lay)
            assert torch.equal(got, want), ("fused bp", case, D, W, H, B, P, Pw, Ph, lay)
            monkeypatch.setenv("LIFTREG_CONV0_PC", "1")
            got_pc = ops.conv3d_first_split(moving, tv, w0, b0, out_layout=lay)
            monkeypatch.delenv("LIFTREG_CONV0_PC")
            assert torch.equal(got_pc, want), ("pc kernel", case, D, W, H, B, P, lay)
        got_cat = ops.conv3d_k3_lrelu(torch.cat([moving, tv], 1), w0, b0, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=lay)
        assert torch.equal(got_cat, want), ("concatenated input", case, D, W, H, B, P, lay)
        # ---- sign mask of block 0 -> data gradient of block 1
        mask = torch.empty((B, D, W, H, 4), dtype=torch.uint8, device=dev)
        y0 = ops.conv3d_k3_lrelu(torch.cat([moving, tv], 1), w0, b0, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=lay, mask_out=mask)
        assert torch.equal(y0, want)
        Cout1 = int(rs.choice([16, 32]))
        w1 = T(rs.normal(0, 0.2, (Cout1, 16, 3, 3, 3)).astype(np.float32), dev)
        y1 = ops.conv3d_k3_lrelu(y0, w1, None, 2, in_layout=lay, out_layout=ops.LAYOUT_NDHWC)
        gpre = T(rs.normal(0, 1, tuple(y1.shape)).astype(np.float32), dev)
        g_act, gw_a, gb_a = ops_bwd.conv3d_bwd(y0, lay, w1, y1, ops.LAYOUT_NDHWC, gpre, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True, mask_input_slope=0.2)
        g_bit, gw_b, gb_b = ops_bwd.conv3d_bwd(y0, lay, w1, y1, ops.LAYOUT_NDHWC, gpre, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True, mask_input_slope=0.2,
                                               x_sign4=mask)
        assert torch.equal(g_act, g_bit) and torch.equal(gw_a, gw_b) and torch.equal(gb_a, gb_b), ("sign mask", case, D, W, H, Cout1, lay)
        monkeypatch.setenv("LIFTREG_DGRAD_OLD", "1")
        g_old, _, _ = ops_bwd.conv3d_bwd(y0, lay, w1, y1, ops.LAYOUT_NDHWC, gpre, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True, mask_input_slope=0.2)
        monkeypatch.delenv("LIFTREG_DGRAD_OLD")
        assert torch.equal(g_old, g_act), ("weights-in-LDS data gradient vs per-tile kernel", case, D, W, H, Cout1, lay)
        # the weight gradient's three kernels: bf16-split operands on the bf16 MFMA (the default since round 5), fp32 MFMA with
        # two-row bricks (LIFTREG_WGRAD_SPLIT=0) and with one-row bricks — other summation orders, equal to rounding
        monkeypatch.setenv("LIFTREG_WGRAD_SPLIT", "0")
        _, gw_2, gb_2 = ops_bwd.conv3d_bwd(y0, lay, w1, y1, ops.LAYOUT_NDHWC, gpre, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True, need_gx=False)
        monkeypatch.setenv("LIFTREG_WGRAD_ROWS", "1")
        _, gw_1, gb_1 = ops_bwd.conv3d_bwd(y0, lay, w1, y1, ops.LAYOUT_NDHWC, gpre, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True, need_gx=False)
        monkeypatch.delenv("LIFTREG_WGRAD_ROWS")
        monkeypatch.delenv("LIFTREG_WGRAD_SPLIT")
        scale = float(gw_1.abs().max()) + 1e-20
        bscale = float(gb_1.abs().max()) + 1e-20
        for gw_v, gb_v in ((gw_a, gb_a), (gw_2, gb_2)):
            assert float((gw_1 - gw_v).abs().max()) <= 2e-5 * scale and float((gb_1 - gb_v).abs().max()) <= 2e-5 * bscale
        # ---- slab decode with the NCC epilogue
        Bd, Lat = int(rs.randint(1, 10)), int(rs.randint(1, 9))
        V = D * W * H
        img = T(rs.uniform(-1, 1, (Bd, 1, D, W, H)).astype(np.float32), dev)
        tgt = T(rs.uniform(-1, 1, (Bd, 1, D, W, H)).astype(np.float32), dev)
        basis = T(rs.normal(0, 0.2, (Lat, 3 * V)).astype(np.float32), dev)
        if rs.rand() < 0.4:
            basis = basis.to(torch.bfloat16)
        mean = T(rs.normal(0, 0.02, 3 * V).astype(np.float32), dev)
        coefs = T(rs.normal(0, 1, (Bd, Lat)).astype(np.float32), dev)
        ids = [T(t, dev) for t in N.identity_axis_tables((D, W, H))]
        d_ref, p_ref, w_ref = ops.pca_warp(coefs, basis, mean, ids, img)
        m_want = ops.ncc_moments(w_ref, tgt, Bd).cpu().numpy()
        d0 = int(rs.randint(0, D))
        d1 = int(rs.randint(d0 + 1, D + 1))
        m_sum = np.zeros_like(m_want)
        for lo, hi in ((0, d0), (d0, d1), (d1, D)):
            if hi == lo:
                continue
            plane = W * H
            cols = torch.cat([torch.arange((c * D + lo) * plane, (c * D + hi) * plane, device=dev) for c in range(3)])
            bs, ms = (basis, mean) if rs.rand() < 0.5 else (basis[:, cols].contiguous(), mean[cols].contiguous())
            d_s, p_s, w_s, m_s = ops.pca_warp(coefs, bs, ms, ids, img, d0=lo, d1=hi, target=tgt[:, :, lo:hi].contiguous())
            assert torch.equal(d_s, d_ref[:, :, lo:hi]) and torch.equal(p_s, p_ref[:, :, lo:hi]) and torch.equal(w_s, w_ref[:, :, lo:hi]), \
                ("slab decode", case, D, W, H, Bd, Lat, lo, hi, bs.shape)
            m_sum += m_s.cpu().numpy()
        np.testing.assert_allclose(m_sum, m_want, rtol=1e-11, atol=1e-9, err_msg=str(("slab moments", case, D, W, H, Bd)))


def test_fuzz_fused_first_blocks_backward(dev):
    """lr_conv3d_dgrad_wgrad0_f32 on random extents (odd D / W, H a multiple of 4 up to several tiles, 2 to 5 input channels,
    batch 1-2) against the two kernels it replaces."""
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(733 + SEED)
    for case in range(max(6, N_CASES // 4)):
        D, W, H = int(rs.randint(2, 22)), int(rs.randint(2, 14)), 4 * int(rs.randint(2, 20))
        B, cin0 = int(rs.randint(1, 3)), int(rs.choice([2, 3, 4, 5]))
        x0 = T(rs.uniform(-1, 1, (B, cin0, D, W, H)).astype(np.float32), dev)
        w0 = T(rs.normal(0, 0.3, (16, cin0, 3, 3, 3)).astype(np.float32), dev)
        b0 = T(rs.normal(0, 0.1, 16).astype(np.float32), dev)
        w1 = T(rs.normal(0, 0.2, (32, 16, 3, 3, 3)).astype(np.float32), dev)
        lay = ops.LAYOUT_NDHWC_HPS
        if cin0 <= 3:
            mask = torch.empty((B, D, W, H, 4), dtype=torch.uint8, device=dev)
            y0 = ops.conv3d_k3_lrelu(x0, w0, b0, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=lay, mask_out=mask)
        else:    # (the generic first-block kernel writes no mask for 4 / 5 channels: bit r of byte q = "channel 4q + r > 0")
            y0 = ops.conv3d_k3_lrelu(x0, w0, b0, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=lay)
            bits = (ops.hps_to_ndhwc(y0) > 0).view(B, D, W, H, 4, 4).to(torch.uint8)
            mask = (bits[..., 0] | (bits[..., 1] << 1) | (bits[..., 2] << 2) | (bits[..., 3] << 3)).contiguous()
        y1 = ops.conv3d_k3_lrelu(y0, w1, None, 2, in_layout=lay, out_layout=ops.LAYOUT_NDHWC)
        gpre1 = T(rs.normal(0, 1, tuple(y1.shape)).astype(np.float32), dev)
        gw0, gb0 = ops_bwd.conv3d_dgrad_wgrad0(gpre1, w1, mask, 0.2, x0)
        gpre0, _, _ = ops_bwd.conv3d_bwd(y0, lay, w1, y1, ops.LAYOUT_NDHWC, gpre1, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True,
                                         mask_input_slope=0.2, x_sign4=mask)
        _, gw_ref, gb_ref = ops_bwd.conv3d_bwd(x0, ops.LAYOUT_NCDHW, w0, y0, lay, gpre0, ops.LAYOUT_NDHWC, 1, need_gx=False,
                                               gy_is_gpre=True)
        tag = str((D, W, H, B, cin0))
        assert float((gw0 - gw_ref).abs().max()) <= 3e-5 * max(1e-3, float(gw_ref.abs().max())), tag
        assert float((gb0 - gb_ref).abs().max()) <= 3e-5 * max(1e-3, float(gb_ref.abs().max())), tag


def test_fuzz_pair_kernel_shapes_channels_and_slabs(dev):
    """The fused blocks-0+1 kernel (csrc/conv01_fused.hip; layers.py:365-369 twice) on random shapes: every channel count it is built
    for (2..5: padded K, dense K for three channels, the five-channel form), odd depths, ragged 4 x 8 columns, several units per
    block (LIFTREG_PAIR01_BLOCKS), and a random z-slab — against an fp64 convolution (2e-6 of the scale) and, for the slab, the
    bits of the whole-volume launch."""
    import torch.nn.functional as F
    from liftreg_amd import ops
    rs = np.random.RandomState(1234 + SEED)
    for case in range(N_CASES):
        Cin = int(rs.randint(2, 6))
        B = int(rs.randint(1, 4))
        D, W = int(rs.randint(1, 19)), int(rs.randint(3, 41))
        H = 4 * int(rs.randint(1, 14))
        g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
        x = torch.randn(B, Cin, D, W, H, generator=g)
        w0 = torch.randn(16, Cin, 3, 3, 3, generator=g) * (2.0 / (27 * Cin)) ** 0.5
        b0 = torch.randn(16, generator=g) * 0.1
        w1 = torch.randn(32, 16, 3, 3, 3, generator=g) * (2.0 / 432) ** 0.5
        b1 = torch.randn(32, generator=g) * 0.1
        y = F.leaky_relu(F.conv3d(x.double(), w0.double(), b0.double(), padding=1), 0.2)
        ref = F.leaky_relu(F.conv3d(y, w1.double(), b1.double(), stride=2, padding=1), 0.2)
        xd, w0d, b0d, w1d, b1d = (t.to(dev) for t in (x, w0, b0, w1, b1))
        x0, rest = xd[:, 0:1].contiguous(), xd[:, 1:].contiguous()
        blocks = int(rs.choice([0, 1, 3]))
        if blocks:
            os.environ["LIFTREG_PAIR01_BLOCKS"] = str(blocks)
        try:
            got = ops.conv3d_pair01(x0, rest, w0d, b0d, w1d, b1d, out_layout=ops.LAYOUT_NDHWC)
        finally:
            os.environ.pop("LIFTREG_PAIR01_BLOCKS", None)
        err = float((got.permute(0, 4, 1, 2, 3).double().cpu() - ref).abs().max())
        assert err <= 2e-6 * float(ref.abs().max()), (case, Cin, B, D, W, H, blocks, err)
        if D >= 8:        # a slab [d0, d1) with even bounds: output planes [d0/2, d1/2) from input planes d0-2 .. d1
            d0 = 2 * int(rs.randint(0, D // 4 + 1))
            d1 = min(D - D % 2, d0 + 2 * int(rs.randint(1, 4)))
            if d1 > d0:
                lo, hi = max(d0 - 2, 0), min(d1 + 1, D)
                part = ops.conv3d_pair01(x0[:, :, lo:hi], rest[:, :, lo:hi].contiguous(), w0d, b0d, w1d, b1d, out_layout=ops.LAYOUT_NDHWC,
                                         slab=(D, lo, d0 // 2, (d1 - d0) // 2))
                assert torch.equal(part, got[:, d0 // 2:d1 // 2]), (case, Cin, D, d0, d1)
