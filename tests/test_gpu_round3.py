"""GPU: round-3 additions outside the big-size files — the model's non-reference option keys end to end, the sharded
basis slab built from a LOADED basis (column runs only), the compulsory-bytes model of the PCA gradient."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(n, P, B, dev, seed):
    from liftreg_amd.utils.sdct_projection_utils import scan_poses
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    poses = scan_poses(30, P, n).astype(np.float32)
    return {"source": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
            "target": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
            "target_proj": torch.rand((B, P, n, n), generator=g, device=dev) * 2 - 1,
            "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}


def test_fuse_ncc_key_hands_the_moments_over_explicitly():
    """opt key fuse_ncc: the output dict carries "ncc_moments" (+ "ncc_moments_of": non-reference keys), NCCLoss / SubspaceLoss take them
    explicitly (no pass over the volumes, same value); without the key the output has exactly the reference's 7 keys.
    A later in-place change of `warped` cannot meet stale moments: nothing is cached by tensor identity any more."""
    from liftreg_amd import ops
    from liftreg_amd.layers.losses import NCCLoss
    from liftreg_amd.losses.SubspaceLoss import loss as SubspaceLoss
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    dev = torch.device("cuda:0")
    n, P, L, B = 64, 2, 8, 2
    torch.manual_seed(11)
    plain = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:3"}).to(dev).eval()
    fused = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:3", "fuse_ncc": True}).to(dev).eval()
    fused.load_state_dict(plain.state_dict())
    inp = _inputs(n, P, B, dev, 11)
    sim = NCCLoss(check_nan=False)
    with torch.no_grad():
        o0, o1 = plain(inp), fused(inp)
        assert sorted(o0) == sorted(["warped", "phi", "params", "target", "pca_coefs", "target_proj", "warped_proj"])
        assert sorted(o1) == sorted(list(o0) + ["ncc_moments", "ncc_moments_of"])      # the moments and the tensors they describe
        for k in ("warped", "phi", "params", "pca_coefs"):
            assert torch.equal(o0[k], o1[k]), k
        want = float(sim(o0["warped"], o0["target"]))
        with ops.kernel_timer() as kt:
            got = float(sim(o1["warped"], o1["target"], moments=o1["ncc_moments"]))
            crit = SubspaceLoss({"initial_reg_factor": 0.01, "min_reg_factor": 0.01, "reg_factor_decay_from": 2})
            crit.sim.check_nan = False
            tot = crit({**o1, "epoch": 0})
            names = set(kt.summary())
        assert "ncc_moments" not in names, names
        assert abs(got - want) < 1e-7 and abs(tot["sim_loss"] - want) < 1e-7
        # after an in-place change the caller simply does not pass the (now stale) moments: the loss is recomputed
        o1["warped"].mul_(0.5)
        assert abs(float(sim(o1["warped"], o1["target"])) - float(sim(o0["warped"] * 0.5, o0["target"]))) < 1e-7


def test_pca_slab_of_a_loaded_basis_copies_column_runs_only(tmp_path):
    """A basis loaded from pca_path, offloaded to the host on a sharded rank: pca_slab copies the three column runs of the
    rank's rows to the device — equal to the columns of the full basis, the full array never reaches the device again, and
    the sharded forward built on those slabs equals the unsharded one."""
    from liftreg_amd import parallel as par
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    dev = torch.device("cuda:0")
    n, P, L, B = 64, 2, 6, 1
    rs = np.random.RandomState(5)
    vec = (rs.standard_normal((L, 3 * n ** 3)) * 0.01).astype(np.float32)
    mean = (rs.standard_normal((3 * n ** 3,)) * 0.001).astype(np.float32)
    np.save(os.path.join(tmp_path, "pca_vectors.npy"), vec)
    np.save(os.path.join(tmp_path, "pca_mean.npy"), mean)
    torch.manual_seed(5)
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": str(tmp_path)}).to(dev).eval()
    inp = _inputs(n, P, B, dev, 5)
    with torch.no_grad():
        ref = net(inp)
    net.offload_full_basis()
    assert not net.pca_vectors_LxM.is_cuda
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    before = torch.cuda.memory_allocated()
    plane = n * n
    for world in (2,):
        for r in range(world):
            d0, d1 = par.slab_bounds(n, world, r)
            bs, ms = net.pca_slab(d0, d1, dev)
            cols = np.concatenate([np.arange((c * n + d0) * plane, (c * n + d1) * plane) for c in range(3)])
            assert bs.is_cuda and tuple(bs.shape) == (L, 3 * (d1 - d0) * plane)
            assert np.array_equal(bs.cpu().numpy(), vec[:, cols]) and np.array_equal(ms.cpu().numpy(), mean[cols])
    # both slabs together are one basis worth of bytes; a full device copy or an int64 index would show in the peak
    assert torch.cuda.max_memory_allocated() - before < 1.2 * (vec.nbytes + mean.nbytes), (torch.cuda.max_memory_allocated() - before, vec.nbytes)
    with torch.no_grad():
        outs = par.SlabShardedRegistration(net, par.LocalComm(2)).forward([inp] * 2)
    for r, o in enumerate(outs):
        d0, d1 = par.slab_bounds(n, 2, r)
        assert torch.equal(o["params"], ref["params"][:, :, d0:d1]) and torch.equal(o["warped"], ref["warped"][:, :, d0:d1])


def test_pca_gradient_byte_model_is_compulsory_traffic():
    """ops_bwd.pca_bwd_coef reports basis + ONE read of the gradient (the l-groups' re-reads come from L2): no table entry
    can exceed what HBM delivers."""
    from liftreg_amd import ops, ops_bwd
    dev = torch.device("cuda:0")
    B, L, M = 4, 56, 3 * 32 ** 3
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    basis = torch.randn((L, M), generator=g, device=dev)
    gd = torch.randn((B, M), generator=g, device=dev)
    with ops.kernel_timer() as kt:
        gc = ops_bwd.pca_bwd_coef(gd, basis)
        rec = kt.summary()["pca_bwd_coef"]
    assert rec["info"]["bytes"] == 4 * L * M + 4 * B * M
    want = gd.double() @ basis.double().T
    assert float((gc.double() - want).abs().max() / want.abs().max()) < 1e-5


def test_encoder_input_bf16_records_equal_feature_volume_path():
    """Many views (C4's 11), bf16 variant: the channels-last bf16 encoder input (lr_backproject_encin_bf16) holds
    bf16(moving) | bf16(lr_backproject_f32's samples) | zeros, for the whole volume and for a z-slab written into a strided
    batch; the first block on it (lr_conv3d_first_clin_bf16) equals lr_conv3d_first_bf16 on the fp32 concatenation bit for
    bit — ragged sizes (D, W not multiples of the bundle / step) and an oblique geometry (the slow path) included."""
    from liftreg_amd import ops
    from liftreg_amd.utils.sdct_projection_utils import scan_poses
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(21)
    for (D, W, H), P, B, Pw, Ph, oblique in (((22, 18, 64), 11, 3, 40, 64, False), ((9, 7, 128), 8, 1, 64, 128, False),
                                             ((16, 16, 64), 9, 2, 24, 64, True), ((40, 36, 256), 11, 2, 64, 256, False), ((12, 10, 32), 15, 1, 32, 32, False)):
        mv = torch.rand((B, 1, D, W, H), generator=g, device=dev) * 2 - 1
        proj = torch.rand((B, P, Pw, Ph), generator=g, device=dev) * 2 - 1
        poses = scan_poses(30, P, W).astype(np.float32)
        if oblique:
            poses[:, 1] = W * 0.55          # emitter almost inside the volume: shadows fan out over many detector rows
        assert ops.encoder_input_bf16_supported(mv, proj)
        e = ops.backproject_encoder_input_bf16(mv, proj, poses)
        tv = ops.backproject(proj, poses, (D, W, H))
        assert torch.equal(e[..., 0], mv[:, 0].to(torch.bfloat16))
        assert torch.equal(e[..., 1:P + 1], tv.permute(0, 2, 3, 4, 1).to(torch.bfloat16)), (D, W, H, P, oblique)
        assert bool((e[..., P + 1:] == 0).all())
        # a z-slab into a strided batch (the sharded model's buffers)
        d0, d1 = 3, D - 2
        big = torch.zeros((B, D + 3, W, H, 16), dtype=torch.bfloat16, device=dev)
        ops.backproject_encoder_input_bf16(mv, proj, poses, d0=d0, d1=d1, out=big[:, 2:2 + d1 - d0])
        assert torch.equal(big[:, 2:2 + d1 - d0], e[:, d0:d1]) and bool((big[:, :2] == 0).all()) and bool((big[:, 2 + d1 - d0:] == 0).all())
        if H % 4 == 0:
            w = torch.randn((16, P + 1, 3, 3, 3), generator=g, device=dev) * 0.05
            bias = torch.randn((16,), generator=g, device=dev) * 0.1
            x = torch.cat([mv, tv], 1)
            for lay in (ops.LAYOUT_BF16_NDHWC, ops.LAYOUT_BF16_NDHWC_HPS):
                want = ops.conv3d_first_bf16(x, w, bias, out_layout=lay)
                got = ops.conv3d_first_clin_bf16(e, w, bias, out_layout=lay)
                assert torch.equal(got, want), (D, W, H, P, lay)


def test_first_block_winograd_sweep_vs_c_oracle_and_direct(monkeypatch):
    """The DEFAULT first block (conv3d_planar_kernel, Winograd F(2,3) along H, channels-last output) against the scalar C
    oracle (the reference's fmaf chain) and against the direct sweep (LIFTREG_CONV0_DIRECT=1): same fp32 arithmetic, another
    summation tree -> within 2e-5 of the activation scale, EVERY element (no outlier allowance).  Ragged sizes (D, W, H not
    multiples of the 4 x 4 x 64 brick), 1-3 input channels, both channels-last layouts, the strided-batch output."""
    import oracle.c_oracle as co
    from liftreg_amd import ops
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(5)
    for (D, W, H), B, ci in (((6, 9, 64), 2, 3), ((5, 4, 132), 1, 2), ((9, 11, 72), 1, 3), ((4, 4, 256), 1, 1), ((13, 6, 68), 2, 3)):
        x = torch.from_numpy(rs.uniform(-1, 1, (B, ci, D, W, H)).astype(np.float32)).to(dev)
        w = torch.from_numpy((rs.normal(0, 1, (16, ci, 3, 3, 3)) / (27 * ci) ** 0.5).astype(np.float32)).to(dev)
        b = torch.from_numpy(rs.uniform(-0.1, 0.1, 16).astype(np.float32)).to(dev)
        ref = co.conv3d_k3_lrelu(x.cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), 1, 0.2)          # (B,16,D,W,H)
        scale = float(np.abs(ref).max())
        got = ops.conv3d_k3_lrelu(x, w, b, 1, out_layout=ops.LAYOUT_NDHWC)                            # Winograd sweep
        err = np.abs(got.permute(0, 4, 1, 2, 3).cpu().numpy() - ref).max()
        assert err <= 2e-5 * scale, ((D, W, H), ci, err, scale)
        monkeypatch.setenv("LIFTREG_CONV0_DIRECT", "1")
        direct = ops.conv3d_k3_lrelu(x, w, b, 1, out_layout=ops.LAYOUT_NDHWC)
        monkeypatch.delenv("LIFTREG_CONV0_DIRECT")
        assert np.array_equal(direct.permute(0, 4, 1, 2, 3).cpu().numpy(), ref), "the direct sweep IS the oracle's chain"
        assert not torch.equal(direct, got)                                                          # … and the default is the other kernel
        if H % 2 == 0:
            hps = ops.conv3d_k3_lrelu(x, w, b, 1, out_layout=ops.LAYOUT_NDHWC_HPS)
            assert torch.equal(ops.hps_to_ndhwc(hps), got)
        # strided-batch output (the sharded model's halo-padded buffers): same bits, the padding planes untouched
        big = torch.full((B, D + 3, W, H, 16), 7.0, dtype=torch.float32, device=dev)
        ops.conv3d_k3_lrelu(x, w, b, 1, out_layout=ops.LAYOUT_NDHWC, out=big[:, 2:2 + D])
        assert torch.equal(big[:, 2:2 + D], got) and bool((big[:, :2] == 7.0).all()) and bool((big[:, 2 + D:] == 7.0).all())


def test_stride2_blocks_strided_batch_output_equals_dense(monkeypatch):
    """lr_conv3d_k3_lrelu_obs_f32 / _obs_bf16: the stride-2 blocks writing into a plane range of per-sample padded buffers (one
    launch for the whole batch) give the bits of the dense launch — Winograd rows kernel, direct rows kernel and the bf16 rows
    kernel, with z_phase, every output layout; the planes around the range stay untouched."""
    from liftreg_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    for (D, W, H), B, ci, co_, force_rows in (((10, 12, 32), 3, 16, 32, False), ((6, 128, 128), 2, 16, 32, False), ((8, 10, 16), 2, 32, 32, True)):
        if force_rows:
            monkeypatch.setenv("LIFTREG_CONV_ROWS_ALWAYS", "1")
        else:
            monkeypatch.delenv("LIFTREG_CONV_ROWS_ALWAYS", raising=False)
        x = torch.rand((B, D, W, H, ci), generator=g, device=dev) * 2 - 1
        w = torch.randn((co_, ci, 3, 3, 3), generator=g, device=dev) / (27 * ci) ** 0.5
        b = torch.randn((co_,), generator=g, device=dev) * 0.1
        Do, Wo, Ho = (D - 1) // 2 + 1, (W - 1) // 2 + 1, (H - 1) // 2 + 1
        for zp in (0, 1):
            for lay in (ops.LAYOUT_NDHWC, ops.LAYOUT_NDHWC_HPS):
                want = ops.conv3d_k3_lrelu(x, w, b, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=lay, z_phase=zp)
                big = torch.full((B, Do + 2, Wo, Ho, co_), 3.0, dtype=torch.float32, device=dev)
                ops.conv3d_k3_lrelu(x, w, b, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=lay, z_phase=zp, out=big[:, 1:1 + Do])
                assert torch.equal(big[:, 1:1 + Do], want) and bool((big[:, 0] == 3.0).all()) and bool((big[:, 1 + Do] == 3.0).all())
        xb = x.to(torch.bfloat16)
        for lay in (ops.LAYOUT_BF16_NDHWC, ops.LAYOUT_BF16_NDHWC_HPS):
            want = ops.conv3d_k3_lrelu_bf16(xb, w, b, 2, in_layout=ops.LAYOUT_BF16_NDHWC_HPS, out_layout=lay)
            big = torch.full((B, Do + 2, Wo, Ho, co_), 3.0, dtype=torch.bfloat16, device=dev)
            ops.conv3d_k3_lrelu_bf16(xb, w, b, 2, in_layout=ops.LAYOUT_BF16_NDHWC_HPS, out_layout=lay, out=big[:, 1:1 + Do])
            assert torch.equal(big[:, 1:1 + Do], want) and bool((big[:, 0] == 3.0).all()) and bool((big[:, 1 + Do] == 3.0).all())
    monkeypatch.delenv("LIFTREG_CONV_ROWS_ALWAYS", raising=False)


def test_hu_fold_in_fast_projector_equals_prologue_and_c_oracle():
    """calc_relative_atten_coef folded into the fast projector's tap loads (division by 1000 = multiplication + one
    exact correction step; out-of-volume taps removed through their axis weight) gives the bits of the one-pass prologue
    (IEEE divide per voxel) and of the C oracle — on phantom-like HU, on values around the -1000 clamp, on huge / tiny /
    signed-zero values, with the axis-1 flip, for a z-slab, and through the general kernel as well."""
    from liftreg_amd import ops
    from liftreg_amd.utils.sdct_projection_utils import scan_poses
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    n, P = 48, 3
    p32 = scan_poses(30, P, n).astype(np.float32)
    sp = np.array((2.2, 2.2, 2.2), np.float32)
    hu = torch.rand((n, n, n), generator=g, device=dev) * 2300 - 1150
    hu.view(-1)[:8] = torch.tensor([-1000.0, -999.99994, -1000.0001, 0.0, -0.0, 1e-30, 3e38, -3e38], device=dev)
    for flip in (False, True):
        a = ops.drr_forward(hu, p32, (n + 8, n), sp, hu_input=True, flip_w=flip, fold_hu=False)
        b = ops.drr_forward(hu, p32, (n + 8, n), sp, hu_input=True, flip_w=flip, fold_hu=True)
        assert torch.equal(a, b), flip
    want = co.drr_forward(hu.cpu().numpy(), p32, sp, (n, n), flags=1)
    got = ops.drr_forward(hu, p32, (n, n), sp, hu_input=True, nseg=1)
    assert np.array_equal(got.cpu().numpy(), want)
    # a z-slab: partial DRRs of HU slabs (taps of the planes outside the slab are dropped through their weights)
    d0, d1 = 10, 30
    s1 = ops.drr_forward(hu[d0:d1].contiguous(), p32, (n, n), sp, d0=d0, d1=d1, full_D=n, hu_input=True, nseg=1)
    s2 = ops.drr_forward(ops.hu_to_mu(hu[d0:d1].contiguous()), p32, (n, n), sp, d0=d0, d1=d1, full_D=n, nseg=1)
    assert torch.equal(s1, s2)


@pytest.mark.parametrize("grad_dtype", ["fp32", "bf16"])
def test_bf16_training_sign_mask_of_first_block_gives_the_same_gradients(grad_dtype, monkeypatch):
    """bf16-forward training: the first block's forward also writes the LeakyReLU sign mask of its stored bf16 output
    (lr_conv3d_first_mask_bf16, LR_LAYOUT_SIGN4) and block 1's data gradient reads that byte per channel quad instead of the
    32-byte activation: every parameter gradient keeps its bits (the mask is the same predicate, bf16 bits > 0), for fp32 and
    bf16 gradient storage; the mask equals the sign of the stored output."""
    from liftreg_amd import ops
    from liftreg_amd.losses.SubspaceLoss import loss as SubspaceLoss
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    dev = torch.device("cuda:0")
    n, P, L, B = 64, 2, 8, 2
    inp = _inputs(n, P, B, dev, 17)

    def grads(no_sign4):
        if no_sign4:
            monkeypatch.setenv("LIFTREG_BF16_NO_SIGN4", "1")
        else:
            monkeypatch.delenv("LIFTREG_BF16_NO_SIGN4", raising=False)
        torch.manual_seed(17)
        net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:4", "conv_dtype": "bf16",
                                "grad_dtype": grad_dtype}).to(dev).train()
        crit = SubspaceLoss({"initial_reg_factor": 0.01, "min_reg_factor": 0.01, "reg_factor_decay_from": 2})
        crit.sim.check_nan = False
        with ops.kernel_timer() as kt:
            out = net(inp)
            out["epoch"] = 0
            crit(out)["total_loss"].backward()
            names = set(kt.summary())
        return {k: p.grad.detach().clone() for k, p in net.named_parameters()}, names

    g_old, _ = grads(True)
    g_new, _ = grads(False)
    for k in g_old:
        assert torch.equal(g_old[k], g_new[k]), k
    # the mask itself
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    x = torch.rand((B, 3, 12, 10, 64), generator=g, device=dev) * 2 - 1
    w = torch.randn((16, 3, 3, 3, 3), generator=g, device=dev) * 0.2
    b = torch.randn((16,), generator=g, device=dev) * 0.1
    for lay in (ops.LAYOUT_BF16_NDHWC, ops.LAYOUT_BF16_NDHWC_HPS):
        m = torch.zeros((B, 12, 10, 64, 4), dtype=torch.uint8, device=dev)
        y = ops.conv3d_first_bf16(x, w, b, out_layout=lay, mask_out=m)
        assert torch.equal(y, ops.conv3d_first_bf16(x, w, b, out_layout=lay))
        yp = ops.bf16_hps_to_ndhwc(y) if lay == ops.LAYOUT_BF16_NDHWC_HPS else y
        pos = (yp.float() > 0).reshape(B, 12, 10, 64, 4, 4).to(torch.uint8)
        want = pos[..., 0] | (pos[..., 1] << 1) | (pos[..., 2] << 2) | (pos[..., 3] << 3)
        assert torch.equal(m, want)


def test_wgrad_split_operands_accuracy(monkeypatch):
    """Block 1's fp32 weight gradient on the bf16 MFMA with exact three-way bf16 splits of both operands
    (conv3d_wgrad_cl_split_kernel, the default; LIFTREG_WGRAD_SPLIT=0 = the fp32-MFMA kernel; 6 of the 9 partial products): against an fp64 reference it is at
    least as accurate as the default fp32-MFMA kernel, for plain and parity-split x, ragged row ends and
    both output widths."""
    from liftreg_amd import ops, ops_bwd
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(31)
    for cout, shape, B, hps in ((32, (10, 12, 72), 2, True), (32, (7, 9, 38), 1, False), (16, (6, 8, 130), 1, True)):
        D, W, H = shape
        x = rs.normal(0, 1, (B, 16, D, W, H)).astype(np.float32)
        o = lambda n: (n - 1) // 2 + 1
        g = rs.normal(0, 1, (B, cout, o(D), o(W), o(H))).astype(np.float32)
        xt = torch.from_numpy(x).double()
        wt = torch.zeros((cout, 16, 3, 3, 3), dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv3d(xt, wt, None, stride=2, padding=1).backward(torch.from_numpy(g).double())
        want = wt.grad.numpy()
        want_b = g.astype(np.float64).sum(axis=(0, 2, 3, 4))
        xd = torch.from_numpy(x).to(dev).permute(0, 2, 3, 4, 1).contiguous()
        lay = ops.LAYOUT_NDHWC
        if hps:
            h = torch.arange(H, device=dev)
            inv = torch.empty(H, dtype=torch.long, device=dev)
            inv[(h & 1) * (H // 2) + (h >> 1)] = h
            xd = xd[:, :, :, inv].contiguous()
            lay = ops.LAYOUT_NDHWC_HPS
        gd = torch.from_numpy(g).to(dev).permute(0, 2, 3, 4, 1).contiguous()
        w = torch.zeros((cout, 16, 3, 3, 3), device=dev)
        y = torch.ones_like(gd)

        def run():
            _, gw, gb = ops_bwd.conv3d_bwd(xd, lay, w, y, ops.LAYOUT_NDHWC, gd, ops.LAYOUT_NDHWC, 2, gy_is_gpre=True, need_gx=False, nblk=16)
            return gw.cpu().numpy().astype(np.float64), gb.cpu().numpy().astype(np.float64)

        monkeypatch.setenv("LIFTREG_WGRAD_SPLIT", "0")
        gw_f, gb_f = run()
        monkeypatch.delenv("LIFTREG_WGRAD_SPLIT")
        gw_s, gb_s = run()          # (the default since round 5)
        scale = np.abs(want).max()
        es, ef = np.abs(gw_s - want).max() / scale, np.abs(gw_f - want).max() / scale
        assert es <= 3e-6 and es <= 1.5 * ef + 2e-7, (cout, shape, es, ef)
        assert np.abs(gb_s - want_b).max() <= 3e-6 * np.abs(want_b).max() + 1e-4, (cout, shape)
        assert not np.array_equal(gw_s, gw_f) or scale == 0     # the two kernels really are different code paths
