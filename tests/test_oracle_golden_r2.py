"""CPU: pin the oracle restatements of the rows closed in round 2 against golden vectors produced by importing the
reference (tests/golden/make_golden_r2.py): forward_grids / forward_grids_with_poses (a5), the pose-less float64
backproj_grids (a6'), the CSV geometry wrapper (a2') and the DirLab landmark sampler / TRE (f4)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import ref_ops as ro


def test_forward_grids_bit_exact(golden):
    g = golden("fwd_grids")
    shape, spacing = tuple(int(v) for v in g["shape"]), tuple(float(v) for v in g["spacing"])
    g1, dx1 = ro.forward_grids(30, 3, spacing, shape, receptor_size=(7, 9))
    assert np.array_equal(g1.numpy(), g["g1"]) and np.array_equal(dx1.numpy(), g["dx1"])
    g2, dx2 = ro.forward_grids(24, 2, spacing, shape)                       # default receptor [int(1.5 D), int(1.5 H)]
    assert g2.shape == (2, 15, 12, shape[1], 3)
    assert np.array_equal(g2.numpy(), g["g2"]) and np.array_equal(dx2.numpy(), g["dx2"])
    g3, dx3 = ro.forward_grids_with_poses(g["poses"], spacing, shape, receptor_size=(6, 11))
    assert np.array_equal(g3.numpy(), g["g3"]) and np.array_equal(dx3.numpy(), g["dx3"])


def test_backproj_grids_poseless_float64_bit_exact(golden):
    g = golden("bp_poseless")
    for tag in ("a", "b"):
        shp, pshape = tuple(int(v) for v in g[f"{tag}_shape"]), tuple(int(v) for v in g[f"{tag}_pshape"])
        got = ro.backproj_grid_poseless(int(g[f"{tag}_range"]), int(g[f"{tag}_P"]), shp, pshape)
        assert got.dtype == torch.float64 and got.shape == (int(g[f"{tag}_P"]), 2) + shp
        assert np.array_equal(got.numpy(), g[f"{tag}_grid"])


def test_csv_geometry(golden):
    g = golden("csv_geo")
    sp = tuple(float(v) for v in g["img_spacing"])      # a tuple of Python floats, as the reference's call sites pass it
    poses = ro.csv_geometry_poses(g["geo_mm"], sp)
    assert poses.dtype == np.float64 and np.array_equal(poses, g["poses"])
    assert np.array_equal(ro.calc_relative_atten_coef(g["hu"]), g["mu"])
    got = ro.drr_forward(g["mu"], poses, (9, 13), sp)
    assert np.array_equal(got, g["proj"])
    D, _, H = g["hu"].shape
    got = ro.drr_forward(g["mu"], poses, (int(D * 1.5), int(H * 1.5)), sp)
    assert np.array_equal(got, g["proj_default"])


def test_landmark_tre(golden):
    g = golden("tre")
    tre, xyz, warped, s_norm = ro.landmark_tre(g["source"], g["target"], g["phi"][None], g["dim"], g["spacing"], g["origin"],
                                               g["phi_spacing"])
    assert np.array_equal(s_norm.numpy(), g["source_norm"])
    assert np.array_equal(warped.numpy(), g["warped"])
    assert tre == float(g["tre"]) and np.array_equal(np.array(xyz), g["tre_xyz"])
    assert (g["warped"][:4] == 0).any()          # the fixture does hold landmarks outside the map (zeros padding)


def test_regulariser_is_a_quadratic_form_of_the_pca_coefficients():
    """The identity behind lr_subspace_reg_f32 / ops.subspace_reg_gram, on the oracle alone (fp64): the field is affine in
    the coefficients (…Backproj.py:102) and compute_reg_loss (losses/SubspaceLoss.py:51-67) is a quadratic form of the
    field, so R(mean + c·basis) = r0 + mean_b(2 lin·c_b + c_b^T G c_b) with G, lin, r0 the form's values on the basis."""
    import torch
    rs = np.random.RandomState(4)
    L, B, shape = 4, 3, (6, 5, 8)
    basis = torch.from_numpy(rs.normal(0, 1, (L, 3) + shape))
    mean = torch.from_numpy(rs.normal(0, 1, (3,) + shape))
    coefs = torch.from_numpy(rs.normal(0, 1, (B, L)))
    r = lambda f: ro.disp_reg(f[None])                                   # one field
    q = lambda u, w: (r(u + w) - r(u - w)) / 4                          # polarisation
    G = torch.stack([torch.stack([q(basis[k], basis[j]) for j in range(L)]) for k in range(L)])
    lin = torch.stack([q(mean, basis[k]) for k in range(L)])
    r0 = r(mean)
    disp = mean[None] + torch.einsum("bl,lcdwh->bcdwh", coefs, basis)
    want = ro.disp_reg(disp)
    got = r0 + (2 * coefs @ lin + ((coefs @ G) * coefs).sum(1)).mean()
    assert abs(float(got) - float(want)) <= 1e-12 * abs(float(want))
    # and its gradient w.r.t. the coefficients
    c = coefs.clone().requires_grad_(True)
    ro.disp_reg(mean[None] + torch.einsum("bl,lcdwh->bcdwh", c, basis)).backward()
    gc = (2 * lin[None] + 2 * coefs @ G) / B
    assert float((c.grad - gc).abs().max()) <= 1e-12 * float(gc.abs().max())


def test_projector_reciprocal_division_is_the_ieee_division_on_its_whitelist():
    """liftreg_amd/csrc/drr_forward.hip normalises the sample coordinates (sdct_projection_utils.py:54-56: x / D, y / (W - 1),
    z / H) with q = x r, e = fma(-q, d, x), q' = fma(e, r, q) for the divisors on its launcher's whitelist (fastdiv_ok): here
    EVERY float 2^-20 <= |x| < 2^12 is divided both ways on the CPU — zero mismatches, so the three-operation sequence is the
    IEEE division there (coordinates are 0 or at least an ulp of the emitter distance, > 2^-15, and below 2^12 voxels)."""
    import re
    from oracle import c_oracle as co
    src = open(os.path.join(ROOT, "liftreg_amd", "csrc", "drr_forward.hip")).read()
    listed = re.search(r"static const int ok\[\] = \{([0-9, ]+)\};", src)
    assert listed, "whitelist not found in drr_forward.hip"
    divisors = [int(v) for v in listed.group(1).split(",")]
    assert len(divisors) >= 8
    for d in divisors:
        assert co.fastdiv_mismatches(d) == 0, d
