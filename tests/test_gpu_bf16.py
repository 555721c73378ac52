"""GPU: the bf16-storage variant of the encoder (BASELINE configs C4/C5: "bf16 convs") against its CPU restatement
(oracle/ref_ops.py: conv_block_bf16 / encoder_bf16 — bf16-rounded operands, exact products, fp32 accumulation).
The two differ only by fp32 summation order, which can flip a final bf16 rounding (one bf16 ulp = 2^-8 relative), so
the bar is: ≥99.5 % of the values identical, the rest within one bf16 ulp."""
import numpy as np
import pytest
import torch

from oracle import ref_ops as ro

pytestmark = pytest.mark.gpu
ULP = 2.0 ** -7   # spacing of bf16 relative to the value (8 significant bits)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _close_bf16(got, want, tag):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    same = got == want
    assert same.mean() >= 0.995, (tag, same.mean())
    np.testing.assert_allclose(got, want, rtol=ULP, atol=1e-6, err_msg=tag)


def _to_hps(t):       # plain (B,D,W,H,C) → [parity][H/2][C] rows
    H = t.shape[3]
    h = torch.arange(H, device=t.device)
    inv = torch.empty(H, dtype=torch.long, device=t.device)
    inv[(h & 1) * (H // 2) + (h >> 1)] = h
    return t[:, :, :, inv].contiguous()


def test_bf16_blocks_all_layouts(dev):
    from liftreg_amd import ops
    rs = np.random.RandomState(31)
    L = ops
    cases = [  # cin, cout, shape, B, in_layout, out_layout
        (16, 32, (8, 10, 20), 2, L.LAYOUT_BF16_NDHWC_HPS, L.LAYOUT_BF16_NDHWC_HPS),
        (16, 32, (7, 9, 11), 1, L.LAYOUT_BF16_NDHWC, L.LAYOUT_BF16_NDHWC),
        (32, 32, (8, 8, 16), 2, L.LAYOUT_BF16_NDHWC_HPS, L.LAYOUT_NCDHW),
        (32, 32, (5, 6, 7), 3, L.LAYOUT_BF16_NDHWC, L.LAYOUT_BF16_NDHWC_HPS),
        (32, 16, (6, 6, 36), 1, L.LAYOUT_BF16_NDHWC_HPS, L.LAYOUT_BF16_NDHWC),
        (16, 16, (4, 5, 34), 1, L.LAYOUT_BF16_NDHWC_HPS, L.LAYOUT_BF16_NDHWC),
    ]
    for cin, cout, shape, B, il, ol in cases:
        x = torch.from_numpy(rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32)).to(torch.bfloat16)
        w = torch.from_numpy((rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
        b = torch.from_numpy(rs.uniform(-0.1, 0.1, cout).astype(np.float32))
        want = ro.conv_block_bf16(x.float(), w, b, 2, round_out=(ol != L.LAYOUT_NCDHW))
        xd = x.permute(0, 2, 3, 4, 1).contiguous().to(dev)
        if il == L.LAYOUT_BF16_NDHWC_HPS:
            xd = _to_hps(xd)
        y = ops.conv3d_k3_lrelu_bf16(xd, w.to(dev), b.to(dev), 2, in_layout=il, out_layout=ol)
        tag = str((cin, cout, shape, il, ol))
        if ol == L.LAYOUT_NCDHW:
            assert y.dtype == torch.float32
            np.testing.assert_allclose(y.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-6, err_msg=tag)
        else:
            assert y.dtype == torch.bfloat16
            if ol == L.LAYOUT_BF16_NDHWC_HPS:
                y = ops.bf16_hps_to_ndhwc(y)
            _close_bf16(y.float().permute(0, 4, 1, 2, 3).cpu().numpy(), want.numpy(), tag)


@pytest.mark.parametrize("cin", [16, 32, "16-ty8"])
@pytest.mark.parametrize("zc", [None, 1, 3])
def test_bf16_z_march_vs_oracle_and_row_kernel(dev, zc, cin, monkeypatch):
    """The 16->32 and 32->32 blocks on parity-split rows run as z-marching kernels (conv3d_bf16.hip: conv3d_march_s2_*):
    ragged tiles (Wo % 4, Ho % 16 != 0), odd depths, every z-chunk boundary (LIFTREG_BF16_MARCH_ZC), all three output
    layouts — against the CPU restatement and against the row kernel (LIFTREG_BF16_NO_MARCH) it replaces."""
    from liftreg_amd import ops
    L = ops
    rs = np.random.RandomState(77)
    if cin == "16-ty8":      # the 8 x 16-output columns (512 threads, one block per CU: LIFTREG_BF16_MARCH_TY8)
        cin = 16
        monkeypatch.setenv("LIFTREG_BF16_MARCH_TY8", "1")
    if zc is not None:
        monkeypatch.setenv("LIFTREG_BF16_MARCH_ZC", str(zc))
    cases = [  # shape (D, W, H), B, out_layout
        ((9, 10, 36), 2, L.LAYOUT_BF16_NDHWC_HPS),
        ((7, 9, 34), 1, L.LAYOUT_BF16_NDHWC),
        ((2, 3, 2), 1, L.LAYOUT_BF16_NDHWC),
        ((5, 18, 70), 2, L.LAYOUT_NCDHW),
        ((12, 33, 68), 1, L.LAYOUT_BF16_NDHWC_HPS),
    ]
    for shape, B, ol in cases:
        x = torch.from_numpy(rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32)).to(torch.bfloat16)
        w = torch.from_numpy((rs.normal(0, 1, (32, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
        b = torch.from_numpy(rs.uniform(-0.1, 0.1, 32).astype(np.float32))
        want = ro.conv_block_bf16(x.float(), w, b, 2, round_out=(ol != L.LAYOUT_NCDHW)).numpy()
        xd = _to_hps(x.permute(0, 2, 3, 4, 1).contiguous().to(dev))

        def run():
            y = ops.conv3d_k3_lrelu_bf16(xd, w.to(dev), b.to(dev), 2, in_layout=L.LAYOUT_BF16_NDHWC_HPS, out_layout=ol)
            if ol == L.LAYOUT_NCDHW:
                return y.cpu().numpy()
            if ol == L.LAYOUT_BF16_NDHWC_HPS:
                y = ops.bf16_hps_to_ndhwc(y)
            return y.float().permute(0, 4, 1, 2, 3).cpu().numpy()

        got = run()
        again = run()
        np.testing.assert_array_equal(got, again)   # deterministic
        monkeypatch.setenv("LIFTREG_BF16_NO_MARCH", "1")
        rows = run()
        monkeypatch.delenv("LIFTREG_BF16_NO_MARCH")
        tag = str((cin, shape, B, ol, zc))
        if ol == L.LAYOUT_NCDHW:
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6, err_msg=tag)
            np.testing.assert_allclose(got, rows, rtol=2e-5, atol=2e-6, err_msg=tag)
        else:
            _close_bf16(got, want, tag)
            _close_bf16(got, rows, tag + " vs rows")


def test_first_block_bf16_store_and_cast(dev):
    """Block 0 (fp32 MFMA) writing bf16 rows, and the stand-alone cast: both round to nearest even like torch."""
    from liftreg_amd import ops
    rs = np.random.RandomState(32)
    x = torch.from_numpy(rs.uniform(-1, 1, (2, 3, 6, 7, 20)).astype(np.float32))
    w = torch.from_numpy((rs.normal(0, 1, (16, 3, 3, 3, 3)) / 9).astype(np.float32))
    b = torch.from_numpy(rs.uniform(-0.1, 0.1, 16).astype(np.float32))
    want = ro._bf16(ro.conv_block(x, w, b, 1))
    for ol in (ops.LAYOUT_BF16_NDHWC, ops.LAYOUT_BF16_NDHWC_HPS):
        y = ops.conv3d_k3_lrelu(x.to(dev), w.to(dev), b.to(dev), 1, out_layout=ol)
        assert y.dtype == torch.bfloat16
        if ol == ops.LAYOUT_BF16_NDHWC_HPS:
            y = ops.bf16_hps_to_ndhwc(y)
        _close_bf16(y.float().permute(0, 4, 1, 2, 3).cpu().numpy(), want.numpy(), f"conv0 {ol}")
    # the bf16-MFMA first block: inputs rounded to bf16 too; 3 channels (one pass), 12 channels (4 passes, C4's
    # 11 views + CT), a 5-channel ragged last pass, H not a multiple of 4 (scalar staging) and Cout = 32
    for cin, cout, shape, B in ((3, 16, (6, 7, 20), 2), (12, 16, (5, 4, 72), 1), (5, 32, (4, 9, 13), 1), (3, 16, (9, 6, 130), 1)):
        x = torch.from_numpy(rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32))
        w = torch.from_numpy((rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
        b = torch.from_numpy(rs.uniform(-0.1, 0.1, cout).astype(np.float32))
        want = ro.conv_block_bf16(x, w, b, 1)
        for ol in (ops.LAYOUT_BF16_NDHWC, ops.LAYOUT_BF16_NDHWC_HPS):
            if ol == ops.LAYOUT_BF16_NDHWC_HPS and shape[2] % 2:
                continue
            y = ops.conv3d_first_bf16(x.to(dev), w.to(dev), b.to(dev), out_layout=ol)
            if ol == ops.LAYOUT_BF16_NDHWC_HPS:
                y = ops.bf16_hps_to_ndhwc(y)
            _close_bf16(y.float().permute(0, 4, 1, 2, 3).cpu().numpy(), want.numpy(), f"first_bf16 {cin} {cout} {shape} {ol}")
    v = torch.from_numpy(rs.normal(0, 3, 100003).astype(np.float32))
    assert torch.equal(ops.cast_bf16(v.to(dev)).cpu(), v.to(torch.bfloat16))


def test_model_bf16_encoder_vs_oracle_and_fp32(dev):
    """conv_dtype="bf16": PCA coefficients against the CPU restatement of the same bf16 contract, and the distance
    to the fp32 model (the precision price of bf16 storage, reported not hidden)."""
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    from liftreg_amd import ops
    n, P, Lat, B = 64, 2, 8, 2
    torch.manual_seed(7)
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": Lat, "pca_path": "synthetic:7", "conv_dtype": "bf16"}).to(dev).eval()
    ref32 = model([n, n, n], {"drr_feature_num": P, "latent_dim": Lat, "pca_path": "synthetic:7"}).to(dev).eval()
    ref32.load_state_dict(net.state_dict())
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    poses = ro.scan_poses(30, P, n).astype(np.float32)
    inp = {"source": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
           "target": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
           "target_proj": torch.rand((B, P, n, n), generator=g, device=dev) * 2 - 1,
           "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}
    with torch.no_grad():
        out = net(inp)
        out32 = ref32(inp)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    tv = ops.backproject(inp["target_proj"], poses, (n, n, n)).cpu()
    feat = ro.encoder_bf16(sd, torch.cat([inp["source"].cpu(), tv], 1)).flatten(1)
    h = ro.fc_block(feat, sd["encoders.6.1.fc.weight"], sd["encoders.6.1.fc.bias"])
    h = ro.fc_block(h, sd["encoders.6.2.fc.weight"], sd["encoders.6.2.fc.bias"])
    want = ro.fc_block(h, sd["encoders.6.3.fc.weight"], sd["encoders.6.3.fc.bias"], slope=None)
    got = out["pca_coefs"].cpu().numpy()
    scale = np.abs(want.numpy()).max()
    assert np.abs(got - want.numpy()).max() <= 2e-3 * scale          # rounding flips only
    d32 = np.abs(got - out32["pca_coefs"].cpu().numpy()).max() / scale
    assert d32 < 5e-2, d32                                             # bf16 storage vs fp32: a few 1e-3 in practice
    assert out["warped"].dtype == torch.float32                         # fp32 warp, as configs C4/C5 state


@pytest.mark.parametrize("grad_dtype", ["fp32", "bf16"])
def test_bf16_forward_training_gradients(dev, grad_dtype):
    """C5: "bf16 convs + fp32 warp, training loop with NCC loss backward".  conv_dtype="bf16" under autograd: bf16
    forward, fp32 gradient math on the bf16-rounded activations and weights.  Oracle: ATen autograd of the CPU
    restatement of the same arithmetic (casts are straight-through).  The two forwards agree up to rare one-ulp bf16
    rounding flips, so gradients agree to a few 1e-3 of their scale, not to fp32 round-off."""
    from liftreg_amd.losses.SubspaceLoss import loss as SubspaceLoss
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    shape, P, L, B = (32, 32, 32), 2, 8, 2
    torch.manual_seed(11)
    net = model(list(shape), {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:11", "conv_dtype": "bf16",
                              "grad_dtype": grad_dtype}).to(dev).train()
    tol = 2e-2 if grad_dtype == "fp32" else 4e-2    # bf16 gradient storage: one more rounding per block on both sides
    rs = np.random.RandomState(11)
    poses = ro.scan_poses(30, P, shape[0]).astype(np.float32)
    inp = {"source": torch.from_numpy(rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32)),
           "target": torch.from_numpy(rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32)),
           "target_proj": torch.from_numpy(rs.uniform(-1, 1, (B, P, 32, 32)).astype(np.float32)),
           "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}
    opt = {"initial_reg_factor": 0.01, "min_reg_factor": 0.01, "reg_factor_decay_from": 2}
    out = net({k: v.to(dev) if k != "target_poses" else v for k, v in inp.items()})
    out["epoch"] = 0
    got = SubspaceLoss(dict(opt))(out)
    got["total_loss"].backward()

    params = {k: v.detach().cpu().clone().requires_grad_("gaussian" not in k) for k, v in net.state_dict().items()}
    ref = ro.model_forward(params, inp, net.pca_vectors_LxM.cpu(), net.pca_mean.cpu(), conv_dtype="bf16",
                           grad_dtype=grad_dtype)
    want = ro.subspace_loss(ref, 0, **opt)
    want["total_loss"].backward()
    assert abs(float(got["total_loss"].detach()) - float(want["total_loss"].detach())) < 1e-4
    for k, p in net.named_parameters():
        g, w = p.grad.cpu().numpy().ravel(), params[k].grad.numpy().ravel()
        scale = np.abs(w).max()
        assert scale > 0, k
        assert np.abs(g - w).max() <= tol * scale, (k, np.abs(g - w).max() / scale)
        assert float(np.dot(g, w) / (np.linalg.norm(g) * np.linalg.norm(w))) > (0.9995 if grad_dtype == "fp32" else 0.999), k
    # and a few optimizer steps reduce the loss
    optim = torch.optim.Adam(net.parameters(), lr=2e-4)
    dinp = {k: v.to(dev) if k != "target_poses" else v for k, v in inp.items()}
    net.set_pca(net.pca_vectors_LxM * 40, net.pca_mean)
    losses = []
    for step in range(6):
        optim.zero_grad()
        o = net(dinp)
        o["epoch"] = step
        l = SubspaceLoss(dict(opt))(o)["total_loss"]
        l.backward()
        optim.step()
        losses.append(float(l.detach()))
    assert losses[-1] < losses[0]


def test_bf16_gradient_block_backward(dev):
    """lr_conv3d_dgrad_bf16 + lr_conv3d_wgrad_bf16g_f32 on single blocks (even/odd extents, both mask-source layouts,
    16 and 32 input channels) against ATen autograd of the same contract: bf16 gpre in, the producer's mask applied,
    bf16 gradient out; weight/bias gradients in fp32."""
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(41)
    L = ops
    for cin, shape, B, xl in ((16, (8, 10, 20), 2, L.LAYOUT_BF16_NDHWC_HPS), (32, (7, 9, 11), 1, L.LAYOUT_BF16_NDHWC),
                              (32, (8, 8, 34), 1, L.LAYOUT_BF16_NDHWC_HPS), (16, (5, 6, 7), 3, L.LAYOUT_BF16_NDHWC)):
        cout = 32
        x = torch.from_numpy(rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32)).to(torch.bfloat16)
        w = torch.from_numpy((rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
        osz = tuple((n - 1) // 2 + 1 for n in shape)
        g = torch.from_numpy(rs.normal(0, 1, (B, cout) + osz).astype(np.float32)).to(torch.bfloat16)
        # oracle: x is the producer's LeakyReLU output (slope 0.3), so d/d(pre_prev) = mask(x) * conv_transpose
        xr = x.float().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        b0 = torch.zeros(cout, requires_grad=True)
        pre = torch.nn.functional.conv3d(xr, ro._bf16(wr), b0, stride=2, padding=1)
        pre.backward(g.float())
        want_gx = ro._bf16(torch.where(x.float() > 0, xr.grad, 0.3 * xr.grad))
        xd = x.permute(0, 2, 3, 4, 1).contiguous().to(dev)
        if xl == L.LAYOUT_BF16_NDHWC_HPS:
            xd = _to_hps(xd)
        gd = g.permute(0, 2, 3, 4, 1).contiguous().to(dev)
        gx, gw, gb = ops_bwd.conv3d_bwd_bf16g(xd, xl, w.to(dev), gd, 2, mask_input_slope=0.3, nblk=8)
        tag = str((cin, shape, xl))
        assert gx.dtype == torch.bfloat16
        _close_bf16(gx.float().permute(0, 4, 1, 2, 3).cpu().numpy(), want_gx.numpy(), "gx " + tag)
        np.testing.assert_allclose(gw.cpu().numpy(), wr.grad.numpy(), rtol=2e-4, atol=2e-4, err_msg="gw " + tag)
        np.testing.assert_allclose(gb.cpu().numpy(), b0.grad.numpy(), rtol=2e-4, atol=2e-4, err_msg="gb " + tag)


def test_pca_with_bf16_stored_basis(dev):
    """pca_dtype="bf16": the basis is rounded to bf16 ONCE (storage); reconstruction and its coefficient gradient are
    then the fp32 ops on those values — identical to the fp32 kernels fed the rounded basis."""
    from liftreg_amd import ops, ops_bwd
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    rs = np.random.RandomState(51)
    for B, Lat, M in ((8, 56, 3 * 20 ** 3), (3, 5, 4096), (1, 9, 3 * 8 * 8 * 12)):
        basis = torch.from_numpy(rs.normal(0, 0.02, (Lat, M)).astype(np.float32)).to(dev)
        mean = torch.from_numpy(rs.normal(0, 0.01, M).astype(np.float32)).to(dev)
        coefs = torch.from_numpy(rs.normal(0, 1, (B, Lat)).astype(np.float32)).to(dev)
        g = torch.from_numpy(rs.normal(0, 1, (B, M)).astype(np.float32)).to(dev)
        bq = basis.to(torch.bfloat16)
        assert torch.equal(ops.pca_reconstruct(coefs, bq, mean), ops.pca_reconstruct(coefs, bq.float(), mean))
        assert torch.equal(ops_bwd.pca_bwd_coef(g, bq), ops_bwd.pca_bwd_coef(g, bq.float()))
        want = coefs.double().cpu() @ bq.double().cpu() + mean.double().cpu()
        np.testing.assert_allclose(ops.pca_reconstruct(coefs, bq, mean).cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-6)
    net = model([32, 32, 32], {"drr_feature_num": 2, "latent_dim": 8, "pca_path": "synthetic:5", "pca_dtype": "bf16"}).to(dev).eval()
    net._ensure_pca(dev)
    assert net.pca_vectors_LxM.dtype == torch.bfloat16 and net.pca_mean.dtype == torch.float32
    poses = ro.scan_poses(30, 2, 32).astype(np.float32)
    inp = {"source": torch.rand((1, 1, 32, 32, 32), device=dev) * 2 - 1, "target": torch.rand((1, 1, 32, 32, 32), device=dev),
           "target_proj": torch.rand((1, 2, 32, 32), device=dev), "target_poses": torch.from_numpy(poses[None].copy())}
    with torch.no_grad():
        out = net(inp)
    ref = out["pca_coefs"].double().cpu() @ net.pca_vectors_LxM.double().cpu()
    np.testing.assert_allclose(out["params"].reshape(1, -1).cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-7)


def test_bf16_gradient_first_block_weight_gradient(dev):
    """Block 0 in the bf16-gradient variant: fp32 planar input (rounded to bf16 like the forward did), bf16 gradient;
    3 input channels, 12 (C4) and a ragged 5; rows longer and shorter than one 64-voxel brick."""
    from liftreg_amd import ops, ops_bwd
    rs = np.random.RandomState(43)
    for cin, shape, B in ((3, (6, 7, 72), 2), (12, (4, 5, 8), 1), (5, (3, 9, 132), 1), (3, (9, 4, 64), 1)):
        cout = 16
        x = torch.from_numpy(rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32))
        w = torch.from_numpy((rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32))
        g = torch.from_numpy(rs.normal(0, 1, (B, cout) + shape).astype(np.float32)).to(torch.bfloat16)
        wr, b0 = w.clone().requires_grad_(True), torch.zeros(cout, requires_grad=True)
        torch.nn.functional.conv3d(ro._bf16(x), ro._bf16(wr), b0, stride=1, padding=1).backward(g.float())
        gd = g.permute(0, 2, 3, 4, 1).contiguous().to(dev)
        gx, gw, gb = ops_bwd.conv3d_bwd_bf16g(x.to(dev), ops.LAYOUT_NCDHW_RBF16, w.to(dev), gd, 1, nblk=8)
        assert gx is None
        tag = str((cin, shape))
        np.testing.assert_allclose(gw.cpu().numpy(), wr.grad.numpy(), rtol=2e-4, atol=3e-4, err_msg="gw " + tag)
        np.testing.assert_allclose(gb.cpu().numpy(), b0.grad.numpy(), rtol=2e-4, atol=3e-4, err_msg="gb " + tag)
