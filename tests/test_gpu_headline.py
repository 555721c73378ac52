"""GPU: the headline configuration (BASELINE.json configs[2] = C3: 256^3 CT, 2x256^2 DRR, batch 8, latent 56, fp32)
compared with the oracle AT ITS OWN SIZE — the kernel instances bench.py times (`pca_warp_kernel<false,true,8>`,
the persistent block-0 kernel, the row-major channels-last kernels at 256^3) on the inputs bench.py times them on.

  * one-pass decode == PCA reconstruction + warp kernels, every bit, at 256^3 / B=8 / L=56;
  * D-slab crops of the decode (first rows, a middle band, last rows) against the C oracle, every bit;
  * one whole registration of the timed workload (bench.synth_inputs, sample 0) against the torch-CPU restatement of
    the reference's forward (oracle/ref_ops.model_forward — ≈10 s of host time): displacement field within the north
    star's 1e-4, PCA coefficients 1e-4 relative, warped image and NCC.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import ref_ops as ro



def _need_experimental():
    """These paths live in the experimental build only (include/liftreg_hip.h, last section): `make -C liftreg_amd/csrc exp`,
    then LIFTREG_HIP_LIB=liftreg_amd/csrc/libliftreg_hip_exp.so python -m pytest -m gpu -k 'fused_backprojection or conv0_split'."""
    from liftreg_amd import _hip
    if not _hip.has_experimental():
        pytest.skip("experimental kernels are not in the product library (make exp + LIFTREG_HIP_LIB)")

pytestmark = pytest.mark.gpu

N, P, B, L = 256, 2, 8, 56


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def decode_case(dev):
    """C3-sized decode inputs: basis randn·(0.02/√L) as bench.py's synthetic basis, a small non-zero mean, a smooth-ish
    moving image, coefficients of the size the FC head produces."""
    from liftreg_amd.utils.net_utils import identity_axis_tables
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    V = N ** 3
    basis = torch.empty((L, 3 * V), dtype=torch.float32, device=dev)
    for l in range(L):
        basis[l].normal_(0.0, 0.02 / float(np.sqrt(L)), generator=g)
    mean = torch.empty((3 * V,), dtype=torch.float32, device=dev).normal_(0.0, 0.002, generator=g)
    img = torch.rand((B, 1, N, N, N), generator=g, device=dev) * 2 - 1
    coefs = torch.randn((B, L), generator=g, device=dev)
    coefs[1] *= 8.0          # one sample whose displacements leave the volume in places (zeros-padding taps)
    ids = [torch.from_numpy(t).to(dev) for t in identity_axis_tables((N, N, N))]
    return basis, mean, img, coefs, ids


def test_c3_one_pass_decode_equals_two_kernels_every_bit(decode_case):
    """lr_pca_warp_f32 at the headline size (the instance bench.py times, 21.9 % of the step in round 1) writes the
    same params, phi and warped as lr_pca_reconstruct_f32 + lr_warp_trilinear_f32 (…Backproj.py:102, :68-69)."""
    from liftreg_amd import ops
    basis, mean, img, coefs, ids = decode_case
    assert ops.pca_warp_supported(coefs, basis, img)
    d1, p1, w1 = ops.pca_warp(coefs, basis, mean, ids, img)
    d2 = ops.pca_reconstruct(coefs, basis, mean).view(B, 3, N, N, N)
    p2, w2 = ops.warp(img, d2, ids, None)
    assert torch.equal(d1, d2), "params differ"
    assert torch.equal(p1, p2), "phi differs"
    assert torch.equal(w1, w2), "warped differs"
    assert float((w1 == -1.0).float().mean()) > 1e-4          # the out-of-volume sample did exercise zeros padding
    torch.cuda.synchronize()


def test_c3_decode_slab_crops_equal_c_oracle(decode_case):
    """Rows [0,2), [127,130) and [254,256) of the 256^3 decode against the scalar C restatement (fmaf chain of the PCA
    reconstruction, ATen grid_sample arithmetic of the warp): every bit of params, phi and warped."""
    from liftreg_amd import ops
    basis, mean, img, coefs, ids = decode_case
    disp, phi, warped = ops.pca_warp(coefs, basis, mean, ids, img)
    img_h = img.cpu().numpy()
    coefs_h = coefs.cpu().numpy()
    tabs = [t.cpu().numpy() for t in ids]
    plane, V = N * N, N ** 3
    for d0, d1 in ((0, 2), (127, 130), (254, 256)):
        cols = np.concatenate([np.arange((c * N + d0) * plane, (c * N + d1) * plane) for c in range(3)])
        cols_t = torch.from_numpy(cols).to(basis.device)
        bs = basis[:, cols_t].cpu().numpy()                    # the (L, 3·Dn·W·H) column slab of the basis
        ms = mean[cols_t].cpu().numpy()
        want_disp = co.pca_reconstruct(coefs_h, bs, ms).reshape(B, 3, d1 - d0, N, N)
        assert np.array_equal(disp[:, :, d0:d1].cpu().numpy(), want_disp), (d0, d1, "params")
        want_phi, want_w = co.warp(img_h, want_disp, ids=(tabs[0][d0:d1], tabs[1], tabs[2]), flags=co.USING_SCALE, d0=d0, d1=d1)
        assert np.array_equal(phi[:, :, d0:d1].cpu().numpy(), want_phi), (d0, d1, "phi")
        assert np.array_equal(warped[:, :, d0:d1].cpu().numpy(), want_w), (d0, d1, "warped")


def test_c3_one_registration_of_the_timed_workload_vs_cpu_forward(dev):
    """bench.py's own C3 input and model (seed 2021), B=8 on the GPU; sample 0 through the torch-CPU restatement of
    model.forward (…Backproj.py:49-104, net_utils.py:26-52).  Bars: displacement ≤ 1e-4 (north star), coefficients
    1e-4 relative, warped ≤ 1e-4 absolute (the north star's bar for fp32 warps; measured 3e-5: a ~1e-9 coordinate
    difference times the phantom's edges), NCC ≤ 1e-5."""
    import bench
    from liftreg_amd.layers.losses import NCCLoss
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    cfg = bench.CONFIGS["c3"]
    torch.manual_seed(2021)
    net = model([N, N, N], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:2021"}).to(dev).eval()
    inp = bench.synth_inputs(cfg, dev, seed=2021)
    with torch.no_grad():
        out = net(inp)
        loss = NCCLoss(check_nan=False)(out["warped"][:1], out["target"][:1])
    par = bench.parity_vs_cpu(net, inp, out)
    print("C3 parity vs CPU forward:", par)
    assert par["max_abs_disp"] <= 1e-4, par
    assert par["max_rel_coefs"] <= 1e-4, par
    assert par["max_abs_warped"] <= 1e-4, par
    assert abs(par["ncc_gpu"] - float(loss)) < 1e-6 and par["ncc_abs"] <= 1e-5, par


def test_c3_ncc_moments_in_the_decode_epilogue(decode_case):
    """the one-pass decode with a target accumulates the similarity's five fp64 moments while `warped` is
    still in registers.  At the headline size: params/phi/warped keep their bits, the moments equal the separate
    one-pass NCC kernel's (fp64 sums in another order: ≤1e-12 relative), and NCCLoss takes them (no second pass)."""
    from liftreg_amd import ops
    from liftreg_amd.layers.losses import NCCLoss
    basis, mean, img, coefs, ids = decode_case
    g = torch.Generator(device=img.device)
    g.manual_seed(5)
    target = torch.rand(img.shape, generator=g, device=img.device) * 2 - 1
    d0, p0, w0 = ops.pca_warp(coefs, basis, mean, ids, img)
    d1, p1, w1, m = ops.pca_warp(coefs, basis, mean, ids, img, target=target)
    assert torch.equal(d0, d1) and torch.equal(p0, p1) and torch.equal(w0, w1)
    want = ops.ncc_moments(w0, target, B)
    rel = ((m - want).abs() / want.abs().clamp_min(1e-300)).max()
    assert float(rel) < 1e-12, float(rel)
    with ops.kernel_timer() as kt:
        fused = NCCLoss(check_nan=False)(w1, target, moments=m)     # handed over explicitly: no identity-keyed cache
        names = set(kt.summary())
    assert "ncc_moments" not in names                      # the epilogue's moments were used
    assert abs(float(fused) - float(NCCLoss(check_nan=False)(w0, target))) < 1e-7
    with pytest.raises(ValueError):
        NCCLoss(check_nan=False)(w1, target, moments=m[:3])


def test_c3_first_block_with_fused_backprojection_every_bit(dev):
    """The headline size: block 0 with the backprojection computed in its staging == backproject + block 0, all
    8·256³·16 outputs, on bench.py's own views and moving volumes."""
    _need_experimental()
    import bench
    from liftreg_amd import ops
    from liftreg_amd.utils.sdct_projection_utils import scan_poses
    inp = bench.synth_inputs(bench.CONFIGS["c3"], dev, seed=2021)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    w = torch.randn((16, P + 1, 3, 3, 3), generator=g, device=dev) / 9
    b = torch.randn((16,), generator=g, device=dev) * 0.1
    poses = scan_poses(30, P, N).astype(np.float32)
    tv = ops.backproject(inp["target_proj"], poses, (N, N, N))
    want = ops.conv3d_first_split(inp["source"], tv, w, b, out_layout=ops.LAYOUT_NDHWC_HPS)
    got = ops.conv3d_first_fused_bp(inp["source"], inp["target_proj"], poses, w, b, out_layout=ops.LAYOUT_NDHWC_HPS)
    assert torch.equal(got, want), float((got - want).abs().max())
