"""GPU: BASELINE configs[3] = C4 AT ITS OWN SIZE — 256^3 CT, 11-view limited-angle DRR, bf16 conv blocks, batch 4 per
GPU (batch 16 over 4 GPUs), 4-way z-slab sharding — through the HIP path, on bench.py's own C4 inputs (seed 2021):

  * the whole forward, sample 0, against oracle/ref_ops.model_forward(conv_dtype="bf16") — the CPU restatement of the
    bf16-storage contract (inputs and weights rounded to bf16, exact products, fp32 accumulation / bias / LeakyReLU, bf16
    activations between the blocks; FC head, PCA, warp and NCC in fp32).  GPU and CPU differ only where fp32 summation
    order flips a final bf16 rounding, so the bars are those of tests/test_gpu_bf16.py: coefficients within 2e-3 of
    their scale, displacement within 2e-3 of ITS scale, NCC within 1e-4;
  * 4 virtual ranks (LocalComm(4): 64-row slabs, bf16 halo planes, all-gathered features, all-reduced moments) equal the
    unsharded model bit for bit at that size;
  * the 12-channel first block (the kernel C4 spends most of its time in) against the CPU restatement on crops at the
    volume's corners and centre, and the bf16 stride-2 block behind it on a crop.
"""
import numpy as np
import pytest
import torch

from oracle import ref_ops as ro

pytestmark = pytest.mark.gpu

ULP = 2.0 ** -7


@pytest.fixture(scope="module")
def c4(request):
    import bench
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    dev = torch.device("cuda:0")
    cfg = bench.CONFIGS["c4"]
    n, P, L = cfg["n"], cfg["P"], cfg["L"]
    assert (n, P, cfg["B"]) == (256, 11, 4)
    torch.manual_seed(2021)
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:2021", "conv_dtype": "bf16"}).to(dev).eval()
    inp = bench.synth_inputs(cfg, dev, seed=2021)
    with torch.no_grad():
        out = net(inp)
    return net, inp, out


def test_c4_full_size_forward_vs_bf16_cpu_restatement(c4):
    import bench
    net, inp, out = c4
    par = bench.parity_vs_cpu(net, inp, out)
    print("C4 parity vs bf16 CPU forward:", par)
    ref = bench._cpu_forward_sample0(net, inp)
    coef_scale = float(ref["pca_coefs"].abs().max())
    disp_scale = float(ref["params"].abs().max())
    assert par["max_rel_coefs"] <= 2e-3, par
    assert par["max_abs_disp"] <= 2e-3 * disp_scale, (par, disp_scale)
    assert par["ncc_abs"] <= 1e-4, par
    assert coef_scale > 0 and out["warped"].dtype == torch.float32 and out["params"].dtype == torch.float32


def test_c4_full_size_distance_from_the_fp32_reference(c4):
    """the bf16 configurations are pinned to a bf16 contract the builder wrote — so state, at C4's own size, how
    far the bf16 path is from the REFERENCE's fp32 arithmetic (oracle/ref_ops.model_forward, conv_dtype fp32; the bench line
    prints the same record as `vs_fp32_reference`).  Measured on bench.py's C4 inputs (round 4): displacement max 3.4e-7
    absolute = 8.5e-5 of the field's scale, coefficients 1.0e-4 of theirs — bf16 activations are averaged over 16 K-wide
    sums before they reach a coefficient.  Bars: north_star's 1e-4 on the displacement in absolute units (2.5 % of this synthetic
    field's scale — weak), 5e-4 of the field's scale on the displacement (6 x the measured value) and 2e-3 on the coefficients."""
    import bench
    net, inp, out = c4
    rec = bench.vs_fp32_reference(net, inp, out)
    print("C4 bf16 path vs the fp32 reference forward:", rec)
    assert rec["max_abs_disp"] <= 1e-4, rec                  # north_star's bar, in the field's own units
    assert rec["max_rel_coefs"] <= 2e-3, rec
    assert rec["max_rel_disp"] <= 5e-4, rec                  # relative to the field's scale: the meaningful bar (measured 8.5e-5)
    assert rec["mean_abs_disp"] <= 5e-4 * rec["disp_scale"], rec
    assert rec["max_rel_coefs"] > 1e-6          # it IS another arithmetic: the record must not silently compare bf16 with bf16


def test_c4_full_size_four_virtual_ranks_equal_unsharded(c4):
    from liftreg_amd import parallel as par
    from liftreg_amd.layers.losses import NCCLoss
    net, inp, ref = c4
    n = net.img_sz[0]
    with torch.no_grad():
        ref_loss = NCCLoss(check_nan=False)(ref["warped"], ref["target"])
        outs = par.SlabShardedRegistration(net, par.LocalComm(4)).forward([inp] * 4)
    for r, o in enumerate(outs):
        d0, d1 = par.slab_bounds(n, 4, r)
        assert (d0, d1) == (64 * r, 64 * (r + 1))
        assert torch.equal(o["pca_coefs"], ref["pca_coefs"]), f"rank {r}: coefficients differ"
        assert torch.equal(o["params"], ref["params"][:, :, d0:d1]), f"rank {r}: displacement slab differs"
        assert torch.equal(o["phi"], ref["phi"][:, :, d0:d1])
        assert torch.equal(o["warped"], ref["warped"][:, :, d0:d1])
        assert abs(float(o["sim_loss"]) - float(ref_loss)) < 1e-6


def test_c4_full_size_first_blocks_on_crops_vs_cpu(c4):
    """Block 0 (12 input channels -> 16, fp32 in / bf16 out) and block 1 (bf16 16 -> 32, stride 2) at 256^3, B = 4:
    crops around two corners and the centre against the CPU restatement; >= 99.5 % identical, the rest one bf16 ulp."""
    from liftreg_amd import ops
    net, inp, _ = c4
    n, P = net.img_sz[0], net.drr_feature_num
    b0, b1 = net.encoders[0], net.encoders[1]
    with torch.no_grad():
        tv = ops.backproject(inp["target_proj"], inp["target_poses"][0].numpy(), (n, n, n))
        x = torch.cat([inp["source"], tv], 1)
        del tv
        lin0, lout0 = net._bf16_layouts[0]
        y0 = ops.conv3d_first_bf16(x, b0.conv.weight, b0.conv.bias, out_layout=lout0, negative_slope=b0._slope)
        y0p = ops.bf16_hps_to_ndhwc(y0) if lout0 == ops.LAYOUT_BF16_NDHWC_HPS else y0        # (B,D,W,H,16) bf16
        lin1, lout1 = net._bf16_layouts[1]
        y1 = ops.conv3d_k3_lrelu_bf16(y0, b1.conv.weight, b1.conv.bias, 2, in_layout=lin1, out_layout=lout1, negative_slope=b1._slope)
        y1p = ops.bf16_hps_to_ndhwc(y1) if lout1 == ops.LAYOUT_BF16_NDHWC_HPS else y1
    w0, bb0 = b0.conv.weight.detach().cpu(), b0.conv.bias.detach().cpu()
    w1, bb1 = b1.conv.weight.detach().cpu(), b1.conv.bias.detach().cpu()

    def check(got, want, tag):
        got, want = got.float().numpy(), want.numpy()
        same = got == want
        assert same.mean() >= 0.995, (tag, same.mean())
        # one bf16 ulp where a final rounding flipped; 2e-6 absolute for values near 0 (fp32 summation order over 324 products)
        np.testing.assert_allclose(got, want, rtol=ULP, atol=2e-6, err_msg=str(tag))

    for (z, y_, h) in ((0, 0, 0), (n - 20, n - 20, n - 36), (118, 120, 100)):
        for b in (0, 3):
            zs, ys, hs = slice(z, z + 20), slice(y_, y_ + 20), slice(h, h + 36)
            crop = x[b:b + 1, :, zs, ys, hs].cpu()
            want = ro.conv_block_bf16(crop, w0, bb0, 1)                                       # (1,16,20,20,36), bf16-rounded
            # interior of the crop only where the crop cut the volume (the volume's own faces ARE the conv's zero padding)
            lo = lambda a: 0 if a == 0 else 1
            hi = lambda a, ln: ln if a + ln == n else ln - 1
            sl = (slice(lo(z), hi(z, 20)), slice(lo(y_), hi(y_, 20)), slice(lo(h), hi(h, 36)))
            got = y0p[b, z + sl[0].start:z + sl[0].stop, y_ + sl[1].start:y_ + sl[1].stop, h + sl[2].start:h + sl[2].stop].permute(3, 0, 1, 2).cpu()
            check(got, want[0][:, sl[0], sl[1], sl[2]], ("block0", z, y_, h, b))
    # block 1 on a centre crop of block 0's GPU output (its input is then identical on both sides)
    z, y_, h = 100, 96, 64                                                                     # even starts: stride-2 phase kept
    crop = y0p[1:2, z:z + 22, y_:y_ + 22, h:h + 38].permute(0, 4, 1, 2, 3).float().cpu()
    want = ro.conv_block_bf16(crop, w1, bb1, 2)                                               # (1,32,11,11,19)
    got = y1p[1, z // 2 + 1:z // 2 + 10, y_ // 2 + 1:y_ // 2 + 10, h // 2 + 1:h // 2 + 18].permute(3, 0, 1, 2).cpu()
    check(got, want[0][:, 1:10, 1:10, 1:18], "block1")
