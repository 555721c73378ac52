"""GPU: backward kernels of the training step against torch autograd of the CPU oracle
(oracle/ref_ops.py — the reference's own op sequence, differentiated by ATen exactly as the reference's
`loss.backward()` does, RegistrationNet.py:401)."""
import numpy as np
import pytest
import torch

from oracle import ref_ops as ro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_ncc_backward(dev):
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(1)
    for variant, f in ((0, ro.ncc_loss), (1, ro.ncc_loss_squared)):
        x = rs.uniform(-1, 1, (3, 2, 6, 7, 8)).astype(np.float32)
        y = (0.5 * x + 0.5 * rs.uniform(-1, 1, x.shape)).astype(np.float32)
        xt = torch.from_numpy(x).requires_grad_(True)
        (f(xt, torch.from_numpy(y)) * 1.7).backward()
        R = 3 if variant == 0 else 6
        m = ops.ncc_moments(T(x, dev), T(y, dev), R)
        gx = ops_bwd.ncc_bwd(T(x, dev), T(y, dev), m, torch.tensor(1.7, device=dev), x.size // R, variant)
        np.testing.assert_allclose(gx.cpu().numpy(), xt.grad.numpy(), rtol=2e-4, atol=1e-8)


def test_warp_backward_wrt_displacement(dev):
    from liftreg_amd import ops_bwd
    rs = np.random.RandomState(2)
    for shape, B, C, use_seg, zb in (((9, 11, 13), 2, 1, False, True), ((8, 8, 12), 1, 2, True, True), ((6, 7, 8), 1, 1, False, False)):
        img = rs.uniform(-1, 1, (B, C) + shape).astype(np.float32)
        seg = (rs.uniform(0, 1, (B, C) + shape) > 0.3).astype(np.float32) if use_seg else None
        disp = rs.normal(0, 0.25, (B, 3) + shape).astype(np.float32)
        gw = rs.normal(0, 1, (B, C) + shape).astype(np.float32)
        tabs = ro.identity_axis_tables(shape)
        d = torch.from_numpy(disp).requires_grad_(True)
        src = torch.from_numpy(img)
        if use_seg:
            src = (src + 1) * torch.from_numpy(seg) - 1
        out = ro.warp(src, d + ro.identity_map(shape), zero_boundary=zb, using_scale=True)
        out.backward(torch.from_numpy(gw))
        got = ops_bwd.warp_bwd_disp(T(img, dev), T(disp, dev), [T(t, dev) for t in tabs], None if seg is None else T(seg, dev),
                                    T(gw, dev), using_scale=True, zero_boundary=zb)
        np.testing.assert_allclose(got.cpu().numpy(), d.grad.numpy(), rtol=1e-4, atol=2e-5)


def test_warp_backward_fast_kernel_equals_general_kernel(dev, monkeypatch):
    """float4 rows without a mask take `warp_bwd_fast_kernel`; LIFTREG_WARP_GENERAL=1 forces the general kernel.
    Same bits, including samples outside every face."""
    from liftreg_amd import ops_bwd
    from liftreg_amd.utils import net_utils as N
    rs = np.random.RandomState(5)
    for shape, B, C in (((6, 7, 8), 2, 1), ((5, 9, 16), 1, 2), ((10, 6, 260), 1, 1)):
        img = rs.uniform(-1, 1, (B, C) + shape).astype(np.float32)
        disp = rs.normal(0, 0.7, (B, 3) + shape).astype(np.float32)
        gw = rs.normal(0, 1, (B, C) + shape).astype(np.float32)
        ids = [T(t, dev) for t in N.identity_axis_tables(shape)]
        for sc in (True, False):
            monkeypatch.delenv("LIFTREG_WARP_GENERAL", raising=False)
            fast = ops_bwd.warp_bwd_disp(T(img, dev), T(disp, dev), ids, None, T(gw, dev), using_scale=sc).cpu().numpy()
            monkeypatch.setenv("LIFTREG_WARP_GENERAL", "1")
            gen = ops_bwd.warp_bwd_disp(T(img, dev), T(disp, dev), ids, None, T(gw, dev), using_scale=sc).cpu().numpy()
            monkeypatch.delenv("LIFTREG_WARP_GENERAL", raising=False)
            assert np.array_equal(fast, gen), (shape, sc)
            assert np.abs(fast).max() > 0
            # the accumulate form (gadd: the regulariser's gradient) == the plain gradient + gadd, both kernels
            ga = rs.normal(0, 1, disp.shape).astype(np.float32)
            for env in (None, "1"):
                if env:
                    monkeypatch.setenv("LIFTREG_WARP_GENERAL", env)
                acc = ops_bwd.warp_bwd_disp(T(img, dev), T(disp, dev), ids, None, T(gw, dev), using_scale=sc, gadd=T(ga, dev))
                monkeypatch.delenv("LIFTREG_WARP_GENERAL", raising=False)
                assert np.array_equal(acc.cpu().numpy(), fast + ga), (shape, sc, env)


def test_pca_backward_wrt_coefficients(dev):
    from liftreg_amd import ops_bwd
    rs = np.random.RandomState(3)
    for B, L, M in ((8, 56, 3 * 24 ** 3), (3, 5, 4096), (1, 9, 3 * 8 * 8 * 12)):
        g = rs.normal(0, 1, (B, M)).astype(np.float32)
        basis = rs.normal(0, 0.02, (L, M)).astype(np.float32)
        want = g.astype(np.float64) @ basis.astype(np.float64).T
        got = ops_bwd.pca_bwd_coef(T(g, dev), T(basis, dev)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)
    # batches above 8 rows (the reference's shipped batch is 30): row chunks of ONE launch — the bits of chunk-by-chunk launches
    for B, L, M in ((30, 56, 3 * 20 ** 3), (9, 13, 4096), (64, 8, 2048), (17, 56, 3 * 16 * 16 * 12)):
        g = T(rs.normal(0, 1, (B, M)).astype(np.float32), dev)
        basis = T(rs.normal(0, 0.02, (L, M)).astype(np.float32), dev)
        got = ops_bwd.pca_bwd_coef(g, basis)
        chunks = torch.cat([ops_bwd.pca_bwd_coef(g[i:i + 8].contiguous(), basis) for i in range(0, B, 8)], 0)
        assert torch.equal(got, chunks), (B, L, M)
        np.testing.assert_allclose(got.cpu().numpy(), g.cpu().double().numpy() @ basis.cpu().double().numpy().T, rtol=1e-4, atol=1e-5)
    # ... an unaligned leading dimension keeps the chunk-by-chunk route (the scalar kernel holds 8 rows)
    g = T(rs.normal(0, 1, (11, 1001)).astype(np.float32), dev)
    basis = T(rs.normal(0, 0.02, (5, 1001)).astype(np.float32), dev)
    np.testing.assert_allclose(ops_bwd.pca_bwd_coef(g, basis).cpu().numpy(), g.cpu().double().numpy() @ basis.cpu().double().numpy().T, rtol=1e-4, atol=1e-5)


def test_linear_backward(dev):
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(4)
    for B, K, O, slope in ((8, 16384, 800, 0.2), (8, 800, 256, 0.2), (5, 256, 56, 1.0), (2, 37, 5, 0.2)):
        x = rs.uniform(-1, 1, (B, K)).astype(np.float32)
        w = (rs.normal(0, 1, (O, K)) / np.sqrt(K)).astype(np.float32)
        b = rs.uniform(-0.1, 0.1, O).astype(np.float32)
        gy = rs.normal(0, 1, (B, O)).astype(np.float32)
        xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b))
        ro.fc_block(xt, wt, bt, None if slope == 1.0 else slope).backward(torch.from_numpy(gy))
        y = ops.linear_lrelu(T(x, dev), T(w, dev), T(b, dev), slope)
        gx, gw, gb = ops_bwd.linear_bwd(T(x, dev), T(w, dev), y, T(gy, dev), slope)
        np.testing.assert_allclose(gx.cpu().numpy(), xt.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gw.cpu().numpy(), wt.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(gb.cpu().numpy(), bt.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_conv_block_backward_all_layouts(dev):
    """dgrad (stride-2 blocks), wgrad and bias gradient of convBlock against torch autograd, for every
    activation layout the encoder uses (planar first block; NDHWC / parity-split in between; NCDHW at the end)."""
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(6)
    L = ops
    cases = [  # cin, cout, stride, shape, B, x_layout, y_layout, gy_layout
        (3, 16, 1, (6, 7, 20), 2, L.LAYOUT_NCDHW, L.LAYOUT_NDHWC_HPS, L.LAYOUT_NDHWC),
        (5, 16, 1, (5, 6, 9), 1, L.LAYOUT_NCDHW, L.LAYOUT_NDHWC, L.LAYOUT_NDHWC),
        (12, 16, 1, (4, 5, 8), 2, L.LAYOUT_NCDHW, L.LAYOUT_NDHWC_HPS, L.LAYOUT_NDHWC),
        (3, 16, 1, (9, 13, 72), 1, L.LAYOUT_NCDHW, L.LAYOUT_NDHWC_HPS, L.LAYOUT_NDHWC),
        (16, 32, 2, (9, 12, 70), 1, L.LAYOUT_NDHWC_HPS, L.LAYOUT_NDHWC, L.LAYOUT_NDHWC),
        (16, 32, 2, (8, 10, 20), 2, L.LAYOUT_NDHWC_HPS, L.LAYOUT_NDHWC_HPS, L.LAYOUT_NDHWC),
        (16, 32, 2, (7, 9, 11), 1, L.LAYOUT_NDHWC, L.LAYOUT_NDHWC, L.LAYOUT_NDHWC),
        (32, 32, 2, (8, 8, 16), 2, L.LAYOUT_NDHWC_HPS, L.LAYOUT_NCDHW, L.LAYOUT_NCDHW),
        (32, 32, 2, (4, 6, 6), 3, L.LAYOUT_NDHWC, L.LAYOUT_NDHWC, L.LAYOUT_NDHWC),
        # 8 and 20 input channels: not a multiple of 16 -> the generic weight-gradient kernel (no data gradient there)
        (8, 16, 2, (6, 7, 9), 2, L.LAYOUT_NDHWC, L.LAYOUT_NDHWC, L.LAYOUT_NDHWC),
        (20, 32, 1, (4, 5, 6), 1, L.LAYOUT_NDHWC, L.LAYOUT_NDHWC, L.LAYOUT_NDHWC),
    ]

    def to_layout(t_ncdhw, lay):
        if lay == L.LAYOUT_NCDHW:
            return t_ncdhw.contiguous()
        cl = t_ncdhw.permute(0, 2, 3, 4, 1).contiguous()
        if lay == L.LAYOUT_NDHWC_HPS:
            B, D, W, H, C = cl.shape
            h = torch.arange(H, device=cl.device)
            inv = torch.empty(H, dtype=torch.long, device=cl.device)
            inv[(h & 1) * (H // 2) + (h >> 1)] = h
            cl = cl.reshape(B, D, W, H, C // 16, 16)[:, :, :, inv].permute(0, 1, 2, 4, 3, 5).reshape(B, D, W, H, C).contiguous()
        return cl

    for cin, cout, s, shape, B, xl, yl, gl in cases:
        x = rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32)
        w = (rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)
        b = rs.uniform(-0.1, 0.1, cout).astype(np.float32)
        xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b))
        yref = ro.conv_block(xt, wt, bt, s)
        gy = rs.normal(0, 1, tuple(yref.shape)).astype(np.float32)
        yref.backward(torch.from_numpy(gy))
        xd = to_layout(T(x, dev), xl)
        yd = ops.conv3d_k3_lrelu(xd, T(w, dev), T(b, dev), s, in_layout=xl, out_layout=yl)
        gx, gw, gb = ops_bwd.conv3d_bwd(xd, xl, T(w, dev), yd, yl, to_layout(T(gy, dev), gl), gl, s, need_gx=(s == 2 and cin % 16 == 0), nblk=8)
        tag = str((cin, cout, s, shape))
        np.testing.assert_allclose(gw.cpu().numpy(), wt.grad.numpy(), rtol=2e-4, atol=2e-5, err_msg="gw " + tag)
        np.testing.assert_allclose(gb.cpu().numpy(), bt.grad.numpy(), rtol=2e-4, atol=2e-5, err_msg="gb " + tag)
        if s == 2 and cin % 16 == 0:
            gx_plain = ops.hps_to_ndhwc(gx) if xl == L.LAYOUT_NDHWC_HPS else gx        # grad comes in x's own layout
            np.testing.assert_allclose(gx_plain.permute(0, 4, 1, 2, 3).cpu().numpy(), xt.grad.numpy(), rtol=2e-4, atol=2e-5,
                                       err_msg="gx " + tag)
            # chained form the model uses: the incoming gradient is already this block's gpre (plain NDHWC), and the
            # data-gradient epilogue applies the PRODUCER's LeakyReLU mask (sign of this block's saved input)
            gpre = torch.where(T(yref.detach().numpy(), dev) > 0, T(gy, dev), 0.2 * T(gy, dev))
            gx2, gw2, gb2 = ops_bwd.conv3d_bwd(xd, xl, T(w, dev), yd, yl, to_layout(gpre, L.LAYOUT_NDHWC), None, s,
                                               nblk=8, gy_is_gpre=True, mask_input_slope=0.3)
            want = np.where(x > 0, xt.grad.numpy(), 0.3 * xt.grad.numpy())
            np.testing.assert_allclose(gx2.permute(0, 4, 1, 2, 3).cpu().numpy(), want, rtol=2e-4, atol=2e-5, err_msg="gx2 " + tag)
            np.testing.assert_allclose(gw2.cpu().numpy(), wt.grad.numpy(), rtol=2e-4, atol=2e-5, err_msg="gw2 " + tag)
            np.testing.assert_allclose(gb2.cpu().numpy(), bt.grad.numpy(), rtol=2e-4, atol=2e-5, err_msg="gb2 " + tag)


def test_disp_reg_backward(dev):
    from liftreg_amd import ops_bwd
    rs = np.random.RandomState(7)
    for shape, B in (((6, 7, 9), 2), ((2, 3, 2), 1), ((12, 5, 8), 1), ((3, 2, 16), 2), ((5, 9, 12), 1), ((2, 4, 8), 1), ((3, 4, 20), 1)):
        disp = rs.normal(0, 0.1, (B, 3) + shape).astype(np.float32)
        d = torch.from_numpy(disp).requires_grad_(True)
        (ro.disp_reg(d) * 0.3).backward()
        got = ops_bwd.disp_reg_bwd(T(disp, dev), torch.tensor(0.3, device=dev))
        np.testing.assert_allclose(got.cpu().numpy(), d.grad.numpy(), rtol=1e-4, atol=1e-7)


def test_large_batches_are_chunked(dev):
    """Batch sizes past the kernels' per-block limits (PCA gradient 8, Linear 32) go through in chunks."""
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(8)
    B, K, O = 70, 96, 24
    x = rs.uniform(-1, 1, (B, K)).astype(np.float32)
    w = (rs.normal(0, 1, (O, K)) / np.sqrt(K)).astype(np.float32)
    b = rs.uniform(-0.1, 0.1, O).astype(np.float32)
    gy = rs.normal(0, 1, (B, O)).astype(np.float32)
    xt, wt, bt = (torch.from_numpy(a).requires_grad_(True) for a in (x, w, b))
    yref = ro.fc_block(xt, wt, bt, 0.2)
    yref.backward(torch.from_numpy(gy))
    y = ops.linear_lrelu(T(x, dev), T(w, dev), T(b, dev), 0.2)
    np.testing.assert_allclose(y.cpu().numpy(), yref.detach().numpy(), rtol=1e-4, atol=1e-5)
    gx, gw, gb = ops_bwd.linear_bwd(T(x, dev), T(w, dev), y, T(gy, dev), 0.2)
    np.testing.assert_allclose(gx.cpu().numpy(), xt.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gw.cpu().numpy(), wt.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gb.cpu().numpy(), bt.grad.numpy(), rtol=1e-4, atol=1e-5)
    Lat, M = 7, 1000
    g = rs.normal(0, 1, (19, M)).astype(np.float32)
    basis = rs.normal(0, 0.05, (Lat, M)).astype(np.float32)
    np.testing.assert_allclose(ops_bwd.pca_bwd_coef(T(g, dev), T(basis, dev)).cpu().numpy(), g.astype(np.float64) @ basis.T, rtol=1e-4, atol=1e-5)
    coefs = rs.normal(0, 1, (40, Lat)).astype(np.float32)          # forward PCA: the C entry point tiles up to 32 rows
    got = ops.pca_reconstruct(T(coefs, dev), T(basis, dev), T(np.zeros(M, np.float32), dev))
    np.testing.assert_allclose(got.cpu().numpy(), coefs.astype(np.float64) @ basis, rtol=1e-4, atol=1e-5)


def test_sign_mask_from_the_first_block_replaces_the_activation_re_read(dev, monkeypatch):
    """Training chain block 0 -> block 1: block 0's forward also writes one byte per channel quad (bit r = channel 4q+r > 0) and
    block 1's data gradient takes the producer's LeakyReLU mask from it instead of re-reading the 64-byte activation.
    The mask has exactly the signs of the activation, and the masked data gradient has the same bits either way — through
    the weights-in-LDS kernel and the older per-tile kernel (LIFTREG_DGRAD_OLD), plain and parity-split layouts, ragged sizes."""
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(8)
    for shape, B in (((8, 8, 64), 2), ((6, 10, 36), 1), ((16, 4, 32), 1)):
        D, W, H = shape
        x0 = T(rs.uniform(-1, 1, (B, 3) + shape).astype(np.float32), dev)
        w0 = T(rs.normal(0, 0.3, (16, 3, 3, 3, 3)).astype(np.float32), dev)
        b0 = T(rs.normal(0, 0.1, 16).astype(np.float32), dev)
        w1 = T(rs.normal(0, 0.2, (32, 16, 3, 3, 3)).astype(np.float32), dev)
        for lay in (ops.LAYOUT_NDHWC_HPS, ops.LAYOUT_NDHWC):
            mask = torch.empty((B, D, W, H, 4), dtype=torch.uint8, device=dev)
            assert ops.conv3d_mask_supported(x0, w0, 1, ops.LAYOUT_NCDHW, lay)
            y0 = ops.conv3d_k3_lrelu(x0, w0, b0, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=lay, mask_out=mask)
            assert torch.equal(y0, ops.conv3d_k3_lrelu(x0, w0, b0, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=lay))
            plain = ops.hps_to_ndhwc(y0) if lay == ops.LAYOUT_NDHWC_HPS else y0           # (B,D,W,H,16)
            want_bits = ((plain > 0).to(torch.int32).reshape(B, D, W, H, 4, 4) << torch.arange(4, device=dev, dtype=torch.int32)).sum(-1)
            assert torch.equal(mask.to(torch.int32), want_bits)
            y1 = ops.conv3d_k3_lrelu(y0, w1, None, 2, in_layout=lay, out_layout=ops.LAYOUT_NDHWC)
            gpre = T(rs.normal(0, 1, tuple(y1.shape)).astype(np.float32), dev)
            for old in (False, True):
                if old:
                    monkeypatch.setenv("LIFTREG_DGRAD_OLD", "1")
                else:
                    monkeypatch.delenv("LIFTREG_DGRAD_OLD", raising=False)
                g_act, gw_a, _ = ops_bwd.conv3d_bwd(y0, lay, w1, y1, ops.LAYOUT_NDHWC, gpre, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True,
                                                    mask_input_slope=0.2)
                g_bit, gw_b, _ = ops_bwd.conv3d_bwd(y0, lay, w1, y1, ops.LAYOUT_NDHWC, gpre, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True,
                                                    mask_input_slope=0.2, x_sign4=mask)
                assert torch.equal(g_act, g_bit), (shape, lay, old)
                assert torch.equal(gw_a, gw_b)
            monkeypatch.delenv("LIFTREG_DGRAD_OLD", raising=False)


@pytest.mark.gpu
def test_fused_first_blocks_backward_matches_two_kernels(dev):
    """lr_conv3d_dgrad_wgrad0_f32 (block 1's data gradient + block 0's LeakyReLU mask + block 0's weight / bias gradient in one
    kernel, the 16-channel gradient between them never written) against the two-kernel path it replaces
    (conv3d_bwd of block 1 with the sign mask -> gpre0, conv3d_bwd of block 0 on it) and against float64 torch autograd of the
    same two blocks: ragged sizes (odd D / W, H not a multiple of 32, several tiles per axis), 2..5 input channels (4, 5: the
    4-plane tile form with two waves per quotient plane — the reference's shipped 4-view configuration has 5)."""
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(18)
    for shape, B, cin0 in (((8, 8, 64), 2, 3), ((7, 9, 36), 1, 3), ((18, 6, 40), 1, 2), ((34, 10, 68), 1, 3),
                           ((8, 8, 64), 2, 5), ((7, 9, 36), 1, 4), ((18, 6, 40), 1, 5), ((34, 10, 68), 1, 5), ((11, 7, 36), 2, 4), ((3, 3, 8), 1, 5)):
        D, W, H = shape
        x0 = T(rs.uniform(-1, 1, (B, cin0) + shape).astype(np.float32), dev)
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
        assert ops_bwd.conv3d_dgrad_wgrad0_supported(x0, mask, w1)
        gw0, gb0 = ops_bwd.conv3d_dgrad_wgrad0(gpre1, w1, mask, 0.2, x0)
        gpre0, _, _ = ops_bwd.conv3d_bwd(y0, lay, w1, y1, ops.LAYOUT_NDHWC, gpre1, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True,
                                         mask_input_slope=0.2, x_sign4=mask)
        _, gw_ref, gb_ref = ops_bwd.conv3d_bwd(x0, ops.LAYOUT_NCDHW, w0, y0, lay, gpre0, ops.LAYOUT_NDHWC, 1, need_gx=False,
                                               gy_is_gpre=True)
        sw, sb = float(gw_ref.abs().max()), float(gb_ref.abs().max())
        assert float((gw0 - gw_ref).abs().max()) <= 2e-5 * sw, (shape, cin0, float((gw0 - gw_ref).abs().max()), sw)
        assert float((gb0 - gb_ref).abs().max()) <= 2e-5 * sb, (shape, cin0)
        # float64 autograd of the two blocks (CPU): d/dw0, d/db0 of <conv1(lrelu(conv0(x0))), gpre1>
        xd = x0.double().cpu()
        w0d, b0d = w0.double().cpu().requires_grad_(True), b0.double().cpu().requires_grad_(True)
        y0d = torch.nn.functional.leaky_relu(torch.nn.functional.conv3d(xd, w0d, b0d, padding=1), 0.2)
        y1d = torch.nn.functional.conv3d(y0d, w1.double().cpu(), None, stride=2, padding=1)
        (y1d * gpre1.permute(0, 4, 1, 2, 3).double().cpu()).sum().backward()
        assert float((gw0.cpu().double() - w0d.grad).abs().max()) <= 1e-4 * float(w0d.grad.abs().max()), (shape, cin0)
        assert float((gb0.cpu().double() - b0d.grad).abs().max()) <= 1e-4 * float(b0d.grad.abs().max()), (shape, cin0)
