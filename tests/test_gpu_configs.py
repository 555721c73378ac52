"""GPU: the shapes of BASELINE.json's other configurations as parity cases (fp32):
  C2  128^3, 2x128^2, batch 4                         — whole forward vs the torch-CPU oracle
  C4  11-view limited-angle DRR, z-slab sharded       — 128^3 stand-in (4 virtual ranks) vs the unsharded forward,
                                                        and its 12-channel first block vs the CPU oracle
  C5  384^3 volume with a 2x512^2 detector            — projector, backprojection, first conv block and warp on
                                                        a non-256 volume whose detector differs from the volume
                                                        (crops checked against the CPU oracles; full-size oracle
                                                        runs would take minutes)
"""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import ref_ops as ro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _net(n, P, L, dev, seed):
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    torch.manual_seed(seed)
    return model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": f"synthetic:{seed}"}).to(dev).eval()


def _inputs(n, P, R, B, dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    poses = ro.scan_poses(30, P, n).astype(np.float32)
    return {"source": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
            "target": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
            "target_proj": torch.rand((B, P, R, R), generator=g, device=dev) * 2 - 1,
            "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}


def test_c2_forward_vs_oracle(dev):
    from liftreg_amd.layers.losses import NCCLoss
    n, P, L, B = 128, 2, 16, 4
    net = _net(n, P, L, dev, 12)
    inp = _inputs(n, P, n, B, dev, 12)
    with torch.no_grad():
        out = net(inp)
        loss = NCCLoss()(out["warped"], out["target"])
        ref = ro.model_forward({k: v.cpu() for k, v in net.state_dict().items()}, {k: v.cpu() for k, v in inp.items()},
                               net.pca_vectors_LxM.cpu(), net.pca_mean.cpu())
    np.testing.assert_allclose(out["pca_coefs"].cpu().numpy(), ref["pca_coefs"].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["params"].cpu().numpy(), ref["params"].numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out["warped"].cpu().numpy(), ref["warped"].numpy(), rtol=1e-4, atol=2e-5)
    assert abs(float(loss) - float(ro.ncc_loss(ref["warped"], ref["target"]))) < 1e-5


def test_c4_eleven_views_sharded(dev):
    from liftreg_amd import ops, parallel as par
    n, P, L, B = 128, 11, 8, 1
    net = _net(n, P, L, dev, 4)
    inp = _inputs(n, P, n, B, dev, 4)
    with torch.no_grad():
        ref = net(inp)
        outs = par.SlabShardedRegistration(net, par.LocalComm(4)).forward([inp] * 4)
        for r, o in enumerate(outs):
            d0, d1 = par.slab_bounds(n, 4, r)
            assert torch.equal(o["pca_coefs"], ref["pca_coefs"])
            assert torch.equal(o["warped"], ref["warped"][:, :, d0:d1])
        # the 12-channel first block (4 channel passes per brick) against the CPU oracle on a crop
        blk = net.encoders[0]
        x = torch.cat([inp["source"], ops.backproject(inp["target_proj"], inp["target_poses"][0].numpy(), (n, n, n))], 1)
        y = ops.conv3d_k3_lrelu(x, blk.conv.weight, blk.conv.bias, 1)
        crop = x[:, :, 40:72, 50:82, 0:40].cpu()
        want = ro.conv_block(crop, blk.conv.weight.cpu(), blk.conv.bias.cpu(), 1)
    np.testing.assert_allclose(y[:, :, 41:71, 51:81, 0:39].cpu().numpy(), want[:, :, 1:-1, 1:-1, 0:39].numpy(), rtol=1e-4, atol=1e-5)


def test_c5_shapes_384_volume_512_detector(dev):
    from liftreg_amd import ops
    from liftreg_amd.utils.net_utils import identity_axis_tables
    n, R, P = 384, 512, 2
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    poses64 = ro.scan_poses(30, P, n)
    poses = poses64.astype(np.float32)
    sp = np.array((2.2, 2.2, 2.2), np.float32)
    # projector: coordinates bit-exact against the C oracle on a detector patch, DRR on whole rows
    vol = torch.rand((n, n, n), generator=g, device=dev) * 0.3
    drr = ops.drr_forward(vol, poses, (R, R), sp, nseg=1)
    pix, dx = ops.drr_sample_coords(poses, sp, (n, n, n), (R, R), dev)
    cpix, cdx = co.drr_sample_coords(poses, sp, (n, n, n), (R, R))
    assert np.array_equal(dx.cpu().numpy(), cdx)
    assert np.array_equal(pix[:, 200:216].cpu().numpy(), cpix[:, 200:216])
    want = co.drr_forward(vol.cpu().numpy(), poses, sp, (R, R))
    assert np.array_equal(drr.cpu().numpy(), want)
    # backprojection of 512^2 views into 384^3
    proj = torch.rand((1, P, R, R), generator=g, device=dev) * 2 - 1
    tv = ops.backproject(proj, poses, (n, n, n))
    bpix = ops.backproject_coords(poses, (n, n, n), (R, R), dev)
    assert np.array_equal(bpix[:, 100:104].cpu().numpy(), co.backproject_coords(poses, (n, n, n), (R, R))[:, 100:104])
    assert np.array_equal(tv[:, :, 300:308].cpu().numpy(), co.backproject(proj.cpu().numpy(), poses, (n, n, n), d0=300, d1=308))
    # first conv block on the 384^3 encoder input and the warp of a 384^3 volume: crops against the oracles
    moving = torch.rand((1, 1, n, n, n), generator=g, device=dev) * 2 - 1
    x = torch.cat([moving, tv], 1)
    w = torch.randn((16, 3, 3, 3, 3), generator=g, device=dev) / 9
    b = torch.randn((16,), generator=g, device=dev) * 0.1
    y = ops.conv3d_k3_lrelu(x, w, b, 1)
    wantc = ro.conv_block(x[:, :, 350:384, 0:34, 330:384].cpu(), w.cpu(), b.cpu(), 1)
    np.testing.assert_allclose(y[:, :, 351:384, 0:33, 331:384].cpu().numpy(), wantc[:, :, 1:, :-1, 1:].numpy(), rtol=1e-4, atol=1e-5)
    disp = (torch.rand((1, 3, n, n, n), generator=g, device=dev) - 0.5) * 0.04
    tabs = identity_axis_tables((n, n, n))
    ids = [torch.from_numpy(t).to(dev) for t in tabs]
    phi, warped = ops.warp(moving, disp, ids, None)
    cphi, cw = co.warp(moving.cpu().numpy(), disp[:, :, 190:198].cpu().numpy(), ids=(tabs[0][190:198], tabs[1], tabs[2]),
                       d0=190, d1=198)
    assert np.array_equal(phi[:, :, 190:198].cpu().numpy(), cphi) and np.array_equal(warped[:, :, 190:198].cpu().numpy(), cw)


def test_c1_graph_replay_matches_eager(dev):
    """C1 shape (64³, 2×64², B=1): the forward + NCC captured in one HIP graph gives the eager results bit for bit,
    for the captured batch and for a second batch copied into the static inputs."""
    from liftreg_amd.layers.losses import NCCLoss
    from liftreg_amd.pipeline import GraphedRegistrar
    n, P, L, B = 64, 2, 56, 1
    net = _net(n, P, L, dev, 21)
    sim = NCCLoss(check_nan=False)
    b0, b1 = _inputs(n, P, n, B, dev, 21), _inputs(n, P, n, B, dev, 22)
    reg = GraphedRegistrar(net, b0, sim=sim)
    for batch in (b0, b1, b0):
        with torch.no_grad():
            want = net(batch)
            want_loss = sim(want["warped"], want["target"])
        out, loss = reg(batch)
        torch.cuda.synchronize()
        for k in ("pca_coefs", "params", "phi", "warped"):
            assert torch.equal(out[k], want[k]), k
        assert torch.equal(loss, want_loss)


def test_c5_conv_backward_at_384(dev):
    """C5 is the training configuration (384³): block 1's weight and data gradient at that size, where the saved
    input (16 ch x 384³ x 4 B = 3.6 GB) is past 31-bit byte offsets.  A sparse pre-activation gradient (a few
    single voxels, including both corners) makes the exact answer a handful of gathered input values."""
    from liftreg_amd import ops, ops_bwd
    n, Cin, Cout = 384, 16, 32
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    x = torch.rand((1, n, n, n, Cin), generator=g, device=dev) * 2 - 1            # plain NDHWC
    w = (torch.rand((Cout, Cin, 3, 3, 3), generator=g, device=dev) - 0.5) * 0.2
    no = n // 2
    pts = [((0, 0, 0), 3, 1.5), ((no - 1, no - 1, no - 1), 31, -2.0), ((100, 7, 150), 0, 0.75), ((191, 0, 64), 17, 1.25),
           ((5, 190, 3), 9, -0.5)]
    gpre = torch.zeros((1, no, no, no, Cout), device=dev)
    for (z, y, xx), co, val in pts:
        gpre[0, z, y, xx, co] = val
    ydummy = torch.empty_like(gpre)
    gx, gw, gb = ops_bwd.conv3d_bwd(x, ops.LAYOUT_NDHWC, w, ydummy, ops.LAYOUT_NDHWC, gpre, ops.LAYOUT_NDHWC, 2,
                                    gy_is_gpre=True)
    want_gw = torch.zeros_like(w)
    want_gb = torch.zeros(Cout, device=dev)
    abs_sum = 0.0
    for (z, y, xx), co, val in pts:
        want_gb[co] += val
        for tz in range(3):
            for ty in range(3):
                for tx in range(3):
                    zi, yi, xi = 2 * z + tz - 1, 2 * y + ty - 1, 2 * xx + tx - 1
                    if min(zi, yi, xi) < 0 or max(zi, yi, xi) >= n:
                        continue
                    want_gw[co, :, tz, ty, tx] += val * x[0, zi, yi, xi]
                    # data gradient at that input voxel = val * W[co, :, tap] (the points are far apart)
                    np.testing.assert_allclose(gx[0, zi, yi, xi].cpu().numpy(), (val * w[co, :, tz, ty, tx]).cpu().numpy(),
                                               rtol=1e-5, atol=1e-7)
                    abs_sum += float((val * w[co, :, tz, ty, tx]).abs().sum())
    np.testing.assert_allclose(gw.cpu().numpy(), want_gw.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gb.cpu().numpy(), want_gb.cpu().numpy(), rtol=1e-6, atol=1e-6)
    assert abs(float(gx.abs().sum(dtype=torch.float64)) - abs_sum) < 1e-3 * abs_sum      # and nothing anywhere else


@pytest.mark.parametrize("B", [1, 9])
def test_reference_native_size_160(dev, B):
    """The reference's own configuration (cur_task_setting.json: 160³ volumes, drr_feature_num 4, latent_dim 56;
    the hard-coded Linear(4000, 800) of …Backproj.py:36): whole forward + configured NCC against the CPU oracle.
    160 → 80 → 40 → 20 → 10 → 5: both channels-last layouts (parity-split and plain) are on the path.  The fast path must be
    the one that runs: encoder blocks 0 + 1 as the five-channel pair kernel (no 16-channel activation in memory) and — B = 9 —
    the decode of a batch above 8 as ONE launch."""
    from liftreg_amd import ops
    from liftreg_amd.layers.losses import NCCLoss
    n, P, L = 160, 4, 56
    net = _net(n, P, L, dev, 33)
    assert net.encoders[6][1].fc.in_features == 4000
    inp = _inputs(n, P, 240, B, dev, 33)                       # 1.5x receptor, the reference's default detector size
    with torch.no_grad():
        with ops.kernel_timer() as kt:
            out = net(inp)
            torch.cuda.synchronize()
        names = {k: len(v["ms"]) for k, v in kt.summary().items()}
        assert names.get("conv3d_pair01_c5x16x32_160") == 1 and names.get("pca_warp") == 1, names
        assert not any(k.startswith("conv3d_c5x16") or k.startswith("conv3d_c16x32") for k in names), names
        loss = NCCLoss()(out["warped"], out["target"])
        ref = ro.model_forward({k: v.cpu() for k, v in net.state_dict().items()}, {k: v.cpu() for k, v in inp.items()},
                               net.pca_vectors_LxM.cpu(), net.pca_mean.cpu())
    np.testing.assert_allclose(out["pca_coefs"].cpu().numpy(), ref["pca_coefs"].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["params"].cpu().numpy(), ref["params"].numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out["phi"].cpu().numpy(), ref["phi"].numpy(), rtol=1e-4, atol=1e-6)
    # white-noise moving image: one grey level per voxel of slope, so a 1e-6 difference in phi (≈1e-4 voxel) shows up
    # as up to a few 1e-5 in a handful of the 4 M warped values
    np.testing.assert_allclose(out["warped"].cpu().numpy(), ref["warped"].numpy(), rtol=1e-4, atol=6e-5)
    assert abs(float(loss) - float(ro.ncc_loss(ref["warped"], ref["target"]))) < 1e-5


def test_first_block_full_size_store_paths_agree(dev, monkeypatch):
    """Block 0 at 256^3 (three co-resident persistent blocks per CU).  With the DIRECT sweep (LIFTREG_CONV0_DIRECT=1) the
    channels-last outputs (unconditional bounds-checked buffer stores) and the NCDHW output (plain stores) hold the same
    bits, and the split-input entry point equals the concatenated one.  The default Winograd F(2,3)-along-H sweep of the
    channels-last paths gives the same bits on both of its layouts and entry points, and differs from the direct sum by
    rounding only (every one of the 256^3 x 16 outputs within 4e-6 absolute at |y| <= 4)."""
    from liftreg_amd import ops
    n = 256
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    x0 = torch.rand((1, 1, n, n, n), device=dev, generator=g) * 2 - 1
    rest = torch.rand((1, 2, n, n, n), device=dev, generator=g) * 2 - 1
    w = torch.randn((16, 3, 3, 3, 3), device=dev, generator=g) / 9
    b = torch.randn(16, device=dev, generator=g) * 0.1
    x = torch.cat([x0, rest], 1)
    y_nc = ops.conv3d_k3_lrelu(x, w, b, 1, out_layout=ops.LAYOUT_NCDHW)          # always the direct sweep
    monkeypatch.setenv("LIFTREG_CONV0_DIRECT", "1")
    y_cl = ops.conv3d_k3_lrelu(x, w, b, 1, out_layout=ops.LAYOUT_NDHWC)
    assert torch.equal(y_cl.permute(0, 4, 1, 2, 3), y_nc)
    del y_cl
    y_hps = ops.conv3d_first_split(x0, rest, w, b, out_layout=ops.LAYOUT_NDHWC_HPS)
    assert torch.equal(ops.hps_to_ndhwc(y_hps).permute(0, 4, 1, 2, 3), y_nc)
    del y_hps
    monkeypatch.delenv("LIFTREG_CONV0_DIRECT")
    w_cl = ops.conv3d_k3_lrelu(x, w, b, 1, out_layout=ops.LAYOUT_NDHWC)          # Winograd sweep
    w_hps = ops.conv3d_first_split(x0, rest, w, b, out_layout=ops.LAYOUT_NDHWC_HPS)
    assert torch.equal(ops.hps_to_ndhwc(w_hps), w_cl)
    del w_hps
    err = float((w_cl.permute(0, 4, 1, 2, 3) - y_nc).abs().max())
    assert err <= 4e-6 and float(y_nc.abs().max()) <= 4.0, err


@pytest.mark.parametrize("grad_dtype,B", [("fp32", 1), ("bf16", 4)])
def test_c5_whole_bf16_training_step_at_384(dev, grad_dtype, B):
    """BASELINE configs[4] as a CONFIGURATION, not as parts: 384^3 CT, 2 x 512^2 DRR, bf16 convs + fp32 warp, a training
    loop with the NCC loss backward — model(input) -> SubspaceLoss -> backward -> Adam.step (RegistrationNet.py:389-406)
    — at the configuration's 4 registrations per GPU (batch 32 over 8 GPUs) with bf16 gradients, and at batch 1 with fp32
    gradients.  Every loss is finite, every parameter receives a finite gradient, and Adam reduces the loss.  (What the
    gradients ARE is pinned on the 96^3 twin below.)"""
    from liftreg_amd.losses.SubspaceLoss import loss as SubspaceLoss
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    n, R, P, L = 384, 512, 2, 56
    torch.manual_seed(5)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:5", "conv_dtype": "bf16",
                            "grad_dtype": grad_dtype}).to(dev).train()
    crit = SubspaceLoss({"sim_class": "liftreg_amd.layers.losses.NCCLoss", "initial_reg_factor": 0.01, "min_reg_factor": 0.01,
                         "reg_factor_decay_from": 2})
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, eps=1e-5)            # RegistrationNet.py:245
    ax = torch.linspace(-1, 1, n, device=dev)
    blob = torch.exp(-4 * (ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2))
    target = (blob * 2 - 1)[None, None].expand(B, 1, n, n, n).contiguous()
    moving = torch.roll(target, shifts=(6, -4, 5), dims=(2, 3, 4)).contiguous()      # a displaced copy: NCC can improve
    poses = ro.scan_poses(30, P, n).astype(np.float32)
    inp = {"source": moving, "target": target, "target_proj": torch.rand((B, P, R, R), generator=g, device=dev) * 2 - 1,
           "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}
    losses = []
    for it in range(4):
        opt.zero_grad(set_to_none=True)
        out = net(inp)
        out["epoch"] = it
        res = crit(out)
        assert np.isfinite(res["sim_loss"]) and np.isfinite(res["reg_loss"])
        res["total_loss"].backward()
        for name, p in net.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), name
        opt.step()
        losses.append(float(res["total_loss"]))
    assert out["warped"].shape == (B, 1, n, n, n) and out["params"].dtype == torch.float32      # "fp32 warp"
    assert losses[-1] < losses[0], losses
    del net, opt, out
    torch.cuda.empty_cache()


@pytest.mark.parametrize("grad_dtype", ["bf16", "fp32"])
def test_c5_twin_first_step_gradients_vs_aten_autograd(dev, grad_dtype):
    """C5's training step on a 96^3 twin (same model family and options: bf16 convs + fp32 warp, 2 views, B = 4 per GPU,
    latent 56): the loss and EVERY parameter gradient of the first step against ATen autograd of the CPU restatement of the
    bf16 contract (oracle/ref_ops.model_forward(conv_dtype="bf16", grad_dtype=...) + subspace_loss) — what the reference's
    own loss.backward() would compute under that storage contract.  The two forwards differ by rare one-ulp bf16 rounding
    flips, so gradients agree to a few 1e-3 of their scale and in direction to 0.999."""
    from liftreg_amd.losses.SubspaceLoss import loss as SubspaceLoss
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    n, R, P, L, B = 96, 128, 2, 56, 4
    torch.manual_seed(7)
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:7", "conv_dtype": "bf16",
                            "grad_dtype": grad_dtype}).to(dev).train()
    rs = np.random.RandomState(7)
    ax = np.linspace(-1, 1, n, dtype=np.float32)
    blob = np.exp(-4 * (ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)).astype(np.float32)
    tgt = np.stack([np.roll(blob, s, axis=0) for s in range(B)])[:, None] * 2 - 1 + rs.normal(0, 0.02, (B, 1, n, n, n)).astype(np.float32)
    mov = np.roll(tgt, (3, -2, 2), axis=(2, 3, 4))
    poses = ro.scan_poses(30, P, n).astype(np.float32)
    inp = {"source": torch.from_numpy(mov.astype(np.float32)).contiguous(), "target": torch.from_numpy(tgt.astype(np.float32)).contiguous(),
           "target_proj": torch.from_numpy(rs.uniform(-1, 1, (B, P, R, R)).astype(np.float32)),
           "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}
    opt = {"initial_reg_factor": 0.01, "min_reg_factor": 0.01, "reg_factor_decay_from": 2}
    out = net({k: v.to(dev) if k != "target_poses" else v for k, v in inp.items()})
    out["epoch"] = 0
    got = SubspaceLoss({**opt, "sim_class": "liftreg_amd.layers.losses.NCCLoss"})(out)
    got["total_loss"].backward()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    params = {k: v.detach().cpu().clone().requires_grad_("gaussian" not in k) for k, v in net.state_dict().items()}
    ref = ro.model_forward(params, inp, net.pca_vectors_LxM.cpu(), net.pca_mean.cpu(), conv_dtype="bf16", grad_dtype=grad_dtype)
    want = ro.subspace_loss(ref, 0, **opt)
    want["total_loss"].backward()
    assert abs(float(got["total_loss"].detach()) - float(want["total_loss"].detach())) < 1e-4
    tol = 2e-2 if grad_dtype == "fp32" else 4e-2
    worst = 0.0
    for k, p in net.named_parameters():
        g, w = p.grad.cpu().numpy().ravel().astype(np.float64), params[k].grad.numpy().ravel().astype(np.float64)
        scale = np.abs(w).max()
        assert scale > 0, k
        rel = np.abs(g - w).max() / scale
        worst = max(worst, rel)
        assert rel <= tol, (k, rel)
        assert float(np.dot(g, w) / (np.linalg.norm(g) * np.linalg.norm(w))) > 0.999, k
    print(f"C5 twin (96^3, B=4, grad_dtype {grad_dtype}): worst parameter-gradient distance {worst:.2e} of the gradient's scale")


@pytest.mark.gpu
@pytest.mark.parametrize("B,n,L,C", [(19, 40, 11, 1), (30, 32, 56, 1), (9, 36, 6, 2)])
def test_pca_warp_batches_above_eight_in_one_launch(B, n, L, C):
    """ops.pca_warp with B > 8: ONE launch, chunks of 8 batch rows per (tile, chunk) block with the chunks of a tile dealt to one
    XCD (the basis leaves HBM once per batch — the reference's shipped batch size is 30, cur_task_setting.json:57): the bits of
    the launch-per-chunk form (…Backproj.py:102 + :68-69)."""
    import torch
    from liftreg_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(B * 100 + n)
    V = n * n * n
    basis = torch.randn(L, 3 * V, device=dev, generator=g) * (0.02 / L ** 0.5)
    mean = torch.randn(3 * V, device=dev, generator=g) * 0.01
    coefs = torch.randn(B, L, device=dev, generator=g)
    img = torch.rand(B, C, n, n, n, device=dev, generator=g) * 2 - 1
    ids = tuple(torch.linspace(-1, 1, n, device=dev) for _ in range(3))
    got = ops.pca_warp(coefs, basis, mean, ids, img)
    for lo in range(0, B, 8):
        hi = min(B, lo + 8)
        want = ops.pca_warp(coefs[lo:hi].contiguous(), basis, mean, ids, img[lo:hi].contiguous())
        for gt, wt in zip(got, want):
            assert torch.equal(gt[lo:hi], wt), (lo, hi)
