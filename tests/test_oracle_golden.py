"""CPU: pin BOTH oracles (oracle/ref_ops.py torch restatement, oracle/liftreg_oracle.c plain C)
against the golden vectors produced by importing the reference (tests/golden/make_golden.py).

Bars: sampling grids / pixel coordinates / floor indices bit-exact; fp32 values within 1e-5
(the reference's own reduction order inside ATen differs from a scalar loop by a few ulp)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import ref_ops as ro
from util import pca_basis_32

RTOL, ATOL = 1e-5, 1e-6


def unnorm(g, size):
    g = g.astype(np.float32)
    return ((g + np.float32(1)) / np.float32(2)) * np.float32(size - 1)


@pytest.mark.parametrize("tag", ["drr_a", "drr_b", "drr_c"])
def test_drr_grid_bit_exact(golden, tag):
    g = golden(tag)
    shape = g["hu"].shape
    res = tuple(int(v) for v in g["resolution"])
    grid, dx = ro.project_grid(g["poses"], res, shape, g["spacing"])
    assert np.array_equal(grid.numpy(), g["grid"])           # torch restatement: every bit
    assert np.array_equal(dx.numpy(), g["dx"])
    pix, dxc = co.drr_sample_coords(g["poses"].astype(np.float32), g["spacing"], shape, res)
    D, W, H = shape
    want = np.stack([unnorm(g["grid"][..., 0], D), unnorm(g["grid"][..., 1], W), unnorm(g["grid"][..., 2], H)], -1)
    assert np.array_equal(pix, want)                          # C restatement: every bit
    assert np.array_equal(np.floor(pix), np.floor(want))
    assert np.array_equal(dxc, g["dx"])


@pytest.mark.parametrize("tag", ["drr_a", "drr_b", "drr_c"])
def test_drr_forward(golden, tag):
    g = golden(tag)
    res = tuple(int(v) for v in g["resolution"])
    assert np.array_equal(ro.calc_relative_atten_coef(g["hu"]), g["mu"])
    assert np.array_equal(co.calc_relative_atten_coef(g["hu"]), g["mu"])
    assert np.array_equal(ro.drr_forward(g["mu"], g["poses"], res, g["spacing"]), g["proj"])
    c = co.drr_forward(g["mu"], g["poses"].astype(np.float32), g["spacing"], res)
    np.testing.assert_allclose(c, g["proj"], rtol=RTOL, atol=ATOL)
    c_hu = co.drr_forward(g["hu"], g["poses"].astype(np.float32), g["spacing"], res, flags=1)
    assert np.array_equal(c_hu, c)                            # HU fold == explicit conversion
    c_flip = co.drr_forward(np.flip(g["mu"], 1).copy(), g["poses"].astype(np.float32), g["spacing"], res, flags=2)
    assert np.array_equal(c_flip, c)                          # flip fold == explicit flip


def test_drr_default_receptor_and_poses(golden):
    g = golden("drr_default_receptor")
    D, W, H = g["hu"].shape
    poses = ro.scan_poses(30, 4, W)
    assert np.array_equal(poses, g["poses"])
    res = [int(D * 1.5), int(H * 1.5)]
    assert g["proj"].shape == (4, res[0], res[1])
    got = ro.drr_forward(ro.calc_relative_atten_coef(g["hu"]), poses, res, (2.2, 2.2, 2.2))
    assert np.array_equal(got, g["proj"])


def test_drr_slab_partials_sum(golden):
    g = golden("drr_b")
    res = tuple(int(v) for v in g["resolution"])
    poses = g["poses"].astype(np.float32)
    full = co.drr_forward(g["mu"], poses, g["spacing"], res)
    D = g["mu"].shape[0]
    parts = [co.drr_forward(g["mu"][a:b], poses, g["spacing"], res, d0=a, d1=b, full_D=D)
             for a, b in ((0, 5), (5, 11), (11, D))]
    np.testing.assert_allclose(sum(parts), full, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["bp_a", "bp_b", "bp_c"])
def test_backproject(golden, tag):
    g = golden(tag)
    shape = tuple(int(v) for v in g["shape"])
    pshape = g["proj"].shape[2:]
    grid = ro.backproj_grid(g["poses"][0:1], shape, pshape)
    assert np.array_equal(grid.numpy(), g["grid"])
    tv = ro.backproject(torch.from_numpy(g["proj"]), g["poses"], shape)
    assert np.array_equal(tv.numpy(), g["volume"])
    pix = co.backproject_coords(g["poses"][0], shape, pshape)       # (P,D,W,H,2): (Pw axis, Ph axis)
    want_pw = unnorm(g["grid"][0, :, 1], pshape[0])                   # grid ch1 <-> Pw
    want_ph = unnorm(g["grid"][0, :, 0], pshape[1])                   # grid ch0 <-> Ph
    assert np.array_equal(pix[..., 0], want_pw) and np.array_equal(pix[..., 1], want_ph)
    c = co.backproject(g["proj"], g["poses"][0], shape)
    np.testing.assert_allclose(c, g["volume"], rtol=RTOL, atol=ATOL)
    a, b = 3, shape[0] - 2                                             # slab == rows of the whole
    assert np.array_equal(co.backproject(g["proj"], g["poses"][0], shape, d0=a, d1=b), c[:, :, a:b])


@pytest.mark.parametrize("tag", ["warp_a", "warp_b"])
def test_identity_and_warp(golden, tag):
    g = golden(tag)
    shape = g["img"].shape[2:]
    assert np.array_equal(ro.identity_map(shape).numpy(), g["identity"])
    t0, t1, t2 = ro.identity_axis_tables(shape)
    idm = np.stack(np.broadcast_arrays(t0[:, None, None], t1[None, :, None], t2[None, None, :]))
    assert np.array_equal(idm, g["identity"])
    img, phi = torch.from_numpy(g["img"]), torch.from_numpy(g["phi"])
    cases = {"warped_zeros_scale": (True, True, "bilinear", co.USING_SCALE),
             "warped_border_scale": (False, True, "bilinear", co.USING_SCALE | co.BORDER),
             "warped_zeros_noscale": (True, False, "bilinear", 0),
             "warped_nearest": (True, True, "nearest", co.USING_SCALE | co.NEAREST)}
    for key, (zb, sc, mode, flags) in cases.items():
        assert np.array_equal(ro.warp(img, phi, zb, sc, mode).numpy(), g[key]), key
        cphi, cw = co.warp(g["img"], g["disp"], ids=(t0, t1, t2), flags=flags)
        assert np.array_equal(cphi, g["phi"]), key
        np.testing.assert_allclose(cw, g[key], rtol=RTOL, atol=ATOL, err_msg=key)
        _, cw2 = co.warp(g["img"], g["phi"], ids=None, flags=flags)   # phi passed directly
        assert np.array_equal(cw2, cw)
    cphi, cw = co.warp(g["img"], g["disp"], ids=(t0, t1, t2), flags=co.USING_SCALE)
    a, b = 2, shape[0] - 3
    sphi, sw = co.warp(g["img"], g["disp"][:, :, a:b], ids=(t0[a:b], t1, t2), flags=co.USING_SCALE, d0=a, d1=b)
    assert np.array_equal(sw, cw[:, :, a:b]) and np.array_equal(sphi, cphi[:, :, a:b])


def test_ncc(golden):
    g = golden("ncc")
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    assert np.array_equal(ro.ncc_loss(x, y).numpy(), g["loss_configured"])
    assert np.array_equal(ro.ncc_loss_squared(x, y).numpy(), g["loss_squared"])
    lc, _ = co.ncc_loss(g["x"], g["y"], 0)
    ls, _ = co.ncc_loss(g["x"], g["y"], 1)
    assert abs(lc - float(g["loss_configured"])) < 2e-6
    assert abs(ls - float(g["loss_squared"])) < 2e-6


@pytest.mark.parametrize("tag", ["conv_a", "conv_b", "conv_c"])
def test_conv_block(golden, tag):
    g = golden(tag)
    s = int(g["stride"])
    y = ro.conv_block(torch.from_numpy(g["x"]), torch.from_numpy(g["weight"]), torch.from_numpy(g["bias"]), s)
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(co.conv3d_k3_lrelu(g["x"], g["weight"], g["bias"], s), g["y"], rtol=1e-5, atol=2e-6)


def test_fc(golden):
    g = golden("fc")
    h1 = ro.fc_block(torch.from_numpy(g["x"]), torch.from_numpy(g["w1"]), torch.from_numpy(g["b1"]))
    h2 = ro.fc_block(h1, torch.from_numpy(g["w2"]), torch.from_numpy(g["b2"]), slope=None)
    np.testing.assert_allclose(h1.numpy(), g["h1"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(h2.numpy(), g["h2"], rtol=1e-6, atol=1e-6)
    c1 = co.linear_lrelu(g["x"], g["w1"], g["b1"], 0.2)
    c2 = co.linear_lrelu(c1, g["w2"], g["b2"], 1.0)
    np.testing.assert_allclose(c1, g["h1"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c2, g["h2"], rtol=1e-5, atol=2e-6)


def test_model_forward(golden):
    g = golden("model_32")
    vec, mean = pca_basis_32(int(g["latent_dim"]), 32, int(g["pca_seed"]))
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    assert sorted(sd) == sorted(str(k) for k in g["state_keys"]) and len(sd) == 19
    inp = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in::")}
    with torch.no_grad():
        out = ro.model_forward(sd, inp, torch.from_numpy(vec), torch.from_numpy(mean))
    for k in ("warped", "phi", "params", "target", "pca_coefs"):
        np.testing.assert_allclose(out[k].numpy(), g["out::" + k], rtol=1e-5, atol=1e-6, err_msg=k)
    # C restatement of the PCA reconstruction against the reference's displacement field
    disp = co.pca_reconstruct(g["out::pca_coefs"], vec, mean)
    np.testing.assert_allclose(disp.reshape(g["out::params"].shape), g["out::params"], rtol=1e-5, atol=1e-7)
    # mask compose (…Backproj.py:57-58)
    assert np.array_equal(co.mask_compose(g["in::target"], g["in::target_label"]), g["out::target"])


def test_prologue_and_overlap_metric(golden):
    """f3/f4: the oracle's intensity normalisation and overlap metric against the imported reference."""
    g = golden("metrics")
    assert np.array_equal(ro.normalize_clip(g["hu"], -1000, 0), g["hu_norm"])
    assert np.array_equal(ro.normalize_clip(g["drr"], 0, 6), g["drr_norm"])
    for k in range(int(g["n_cases"])):
        r = ro.cal_metric(g[f"pred{k}"], g[f"gt{k}"])
        assert [r["iou"], r["dice"], r["recall"], r["precision"]] == list(g[f"res{k}"])
