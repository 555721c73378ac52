"""GPU parity tests (run with -m gpu on an MI355X): every HIP kernel, called through the C ABI
(liftreg_amd.ops → libliftreg_hip.so), against
  (1) the golden vectors produced by the reference itself (tests/golden/*.npz),
  (2) the oracles (oracle/liftreg_oracle.c scalar C, oracle/ref_ops.py torch-CPU) on seeded inputs.

Bars (BASELINE.json north_star): sampling coordinates and floor indices bit-exact; fp32 values
within 1e-4 relative (most checks here hold 1e-5 or exact equality with the C oracle, whose
arithmetic order the kernels share)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import ref_ops as ro
from util import pca_basis_32



def _need_experimental():
    """These paths live in the experimental build only (include/liftreg_hip.h, last section): `make -C liftreg_amd/csrc exp`,
    then LIFTREG_HIP_LIB=liftreg_amd/csrc/libliftreg_hip_exp.so python -m pytest -m gpu -k 'fused_backprojection or conv0_split'."""
    from liftreg_amd import _hip
    if not _hip.has_experimental():
        pytest.skip("experimental kernels are not in the product library (make exp + LIFTREG_HIP_LIB)")

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 2e-6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from liftreg_amd import ops as o
    from liftreg_amd import _hip
    _hip.lib()  # fail loudly if the HIP library is missing
    return o


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def unnorm(g, size):
    g = g.astype(np.float32)
    return ((g + np.float32(1)) / np.float32(2)) * np.float32(size - 1)


# ------------------------------------------------------------------------------------- K1 DRR
@pytest.mark.parametrize("tag", ["drr_a", "drr_b", "drr_c"])
def test_drr_coords_bit_exact(golden, ops, dev, tag):
    g = golden(tag)
    shape = g["hu"].shape
    res = tuple(int(v) for v in g["resolution"])
    poses = g["poses"].astype(np.float32)
    pix, dx = ops.drr_sample_coords(poses, g["spacing"], shape, res, dev)
    D, W, H = shape
    want = np.stack([unnorm(g["grid"][..., 0], D), unnorm(g["grid"][..., 1], W), unnorm(g["grid"][..., 2], H)], -1)
    assert np.array_equal(pix.cpu().numpy(), want)                   # every bit of every coordinate
    assert np.array_equal(np.floor(pix.cpu().numpy()), np.floor(want))
    assert np.array_equal(dx.cpu().numpy(), g["dx"])
    grid, _ = ops.drr_sample_coords(poses, g["spacing"], shape, res, dev, normalized=True)
    assert np.array_equal(grid.cpu().numpy(), g["grid"])             # the reference's own grid tensor


@pytest.mark.parametrize("tag", ["drr_a", "drr_b", "drr_c"])
def test_drr_forward_golden(golden, ops, dev, tag):
    g = golden(tag)
    res = tuple(int(v) for v in g["resolution"])
    poses = g["poses"].astype(np.float32)
    want_c = co.drr_forward(g["mu"], poses, g["spacing"], res)
    for nseg in (1, 2, 4):
        got = ops.drr_forward(T(g["mu"], dev), poses, res, g["spacing"], nseg=nseg).cpu().numpy()
        np.testing.assert_allclose(got, g["proj"], rtol=RTOL, atol=ATOL)
        if nseg == 1:
            assert np.array_equal(got, want_c)                         # same op order as the C oracle
    got_hu = ops.drr_forward(T(g["hu"], dev), poses, res, g["spacing"], hu_input=True, nseg=1).cpu().numpy()
    assert np.array_equal(got_hu, want_c)
    flipped = np.flip(g["hu"], 1).copy()
    got_fl = ops.drr_forward(T(flipped, dev), poses, res, g["spacing"], hu_input=True, flip_w=True, nseg=1)
    assert np.array_equal(got_fl.cpu().numpy(), want_c)


def test_drr_reference_signatures(golden, ops, dev):
    from liftreg_amd.utils import sdct_projection_utils as S
    g = golden("drr_default_receptor")
    mu = S.calc_relative_atten_coef(g["hu"])
    assert np.array_equal(mu, ro.calc_relative_atten_coef(g["hu"]))
    proj, poses = S.calculate_projection_wraper(mu, 30, 4, (2.2, 2.2, 2.2))
    assert isinstance(proj, np.ndarray) and proj.dtype == np.float32
    assert np.array_equal(poses, g["poses"])
    np.testing.assert_allclose(proj, g["proj"], rtol=RTOL, atol=ATOL)
    proj2 = S.calculate_projection(mu, poses, [12, 12], [1, 1, 1], (2.2, 2.2, 2.2), torch.device("cuda"))
    np.testing.assert_allclose(proj2, g["proj"], rtol=RTOL, atol=ATOL)
    with pytest.raises(RuntimeError):
        S.calculate_projection(mu, poses, [12, 12], [1, 1, 1], (2.2, 2.2, 2.2), torch.device("cpu"))


def test_drr_medium_vs_oracles_and_slabs(ops, dev):
    rs = np.random.RandomState(5)
    D, W, H, P, R = 40, 36, 44, 3, 48
    mu = rs.uniform(0, 0.4, (D, W, H)).astype(np.float32)
    poses = ro.scan_poses(30, P, W)
    p32 = poses.astype(np.float32)
    sp = np.array((2.2, 2.2, 2.2), np.float32)
    want_t = ro.drr_forward(mu, poses, (R, R), sp)
    want_c = co.drr_forward(mu, p32, sp, (R, R))
    got = ops.drr_forward(T(mu, dev), p32, (R, R), sp, nseg=1).cpu().numpy()
    assert np.array_equal(got, want_c)
    np.testing.assert_allclose(got, want_t, rtol=1e-5, atol=1e-6)
    got_auto = ops.drr_forward(T(mu, dev), p32, (R, R), sp).cpu().numpy()     # library-chosen nseg
    np.testing.assert_allclose(got_auto, want_t, rtol=1e-5, atol=1e-6)
    parts = [ops.drr_forward(T(mu[a:b], dev), p32, (R, R), sp, d0=a, d1=b, full_D=D) for a, b in ((0, 13), (13, 29), (29, D))]
    np.testing.assert_allclose(sum(parts).cpu().numpy(), want_t, rtol=1e-5, atol=1e-6)     # z-slab partial sums


def test_drr_fast_kernel_equals_general_kernel(ops, dev, monkeypatch):
    """Attenuation input runs `drr_forward_fast_kernel`; LIFTREG_DRR_GENERAL=1 forces the general kernel.  Same bits:
    rays leaving through every face (wide detector, close emitter), flipped rows, slabs, explicit segment counts."""
    rs = np.random.RandomState(9)
    sp = np.array((2.2, 1.7, 2.0), np.float32)
    for (D, W, H), R, poses in (((20, 18, 24), (40, 56), np.array([[3.0, 40.0, -2.0], [-9.0, 25.0, 6.0]], np.float32)),
                                ((33, 21, 70), (24, 130), ro.scan_poses(40, 3, 21).astype(np.float32))):
        mu = rs.uniform(0, 0.4, (D, W, H)).astype(np.float32)
        for kw in (dict(nseg=1), dict(nseg=4, flip_w=True), dict(nseg=0),
                   dict(nseg=2, d0=5, d1=D - 4, full_D=D)):
            v = mu[kw.get("d0", 0):kw.get("d1", D)]
            monkeypatch.delenv("LIFTREG_DRR_GENERAL", raising=False)
            fast = ops.drr_forward(T(v, dev), poses, R, sp, **kw).cpu().numpy()
            monkeypatch.setenv("LIFTREG_DRR_GENERAL", "1")
            gen = ops.drr_forward(T(v, dev), poses, R, sp, **kw).cpu().numpy()
            monkeypatch.delenv("LIFTREG_DRR_GENERAL", raising=False)
            assert np.array_equal(fast, gen), ((D, W, H), kw)
            assert fast.max() > 0 and (fast == 0).any()      # some rays cross the volume, some miss it


@pytest.mark.parametrize("n,R,P", [(32, 40, 2), (96, 96, 3), (160, 240, 2), (40, 36, 2)])
def test_drr_batch_and_reciprocal_division_equal_the_c_oracle(ops, dev, n, R, P):
    """lr_drr_forward_batch_f32: B volumes of one geometry in ONE launch = B single launches = the C oracle bit for bit, HU input
    with the flip folded (tools/preprocessingDRR.py:135-154) — at sizes whose normalising divisors (D, W - 1, H) are on the
    projector's reciprocal-division whitelist (32: 31; 96: 96, 95; 160: 160, 159) and at one that is not (40: IEEE division)."""
    rs = np.random.RandomState(n)
    B = 3
    hu = rs.uniform(-1100, 900, (B, n, n, n)).astype(np.float32)
    p32 = ro.scan_poses(30, P, n).astype(np.float32)
    sp = np.array((2.2, 2.2, 2.2), np.float32)
    for nseg in (1, 0):          # one run per ray = the oracle's summation order | the library's choice of runs
        got = ops.drr_forward_batch(T(hu, dev), p32, (R, R), sp, hu_input=True, flip_w=True, nseg=nseg).cpu().numpy()
        for b in range(B):
            one = ops.drr_forward(T(hu[b], dev), p32, (R, R), sp, hu_input=True, flip_w=True, nseg=nseg).cpu().numpy()
            assert np.array_equal(got[b], one), b
        want = co.drr_forward(hu[1][:, ::-1].copy(), p32, sp, (R, R), flags=1)      # the oracle on the flipped volume, HU input
        if nseg == 1:
            assert np.array_equal(got[1], want)
        else:
            np.testing.assert_allclose(got[1], want, rtol=1e-5, atol=1e-6)
    mu = ops.drr_forward_batch(T(hu * 0 + 0.1, dev), p32, (R, R), sp).cpu().numpy()    # attenuation input, no flip
    assert np.array_equal(mu[0], mu[2]) and mu.max() > 0


# ------------------------------------------------------------------------------------- K2 backprojection
@pytest.mark.parametrize("tag", ["bp_a", "bp_b", "bp_c"])
def test_backproject_golden(golden, ops, dev, tag):
    g = golden(tag)
    shape = tuple(int(v) for v in g["shape"])
    pshape = g["proj"].shape[2:]
    poses = g["poses"][0]
    pix = ops.backproject_coords(poses, shape, pshape, dev).cpu().numpy()
    assert np.array_equal(pix[..., 0], unnorm(g["grid"][0, :, 1], pshape[0]))     # bit-exact coordinates
    assert np.array_equal(pix[..., 1], unnorm(g["grid"][0, :, 0], pshape[1]))
    got = ops.backproject(T(g["proj"], dev), poses, shape).cpu().numpy()
    np.testing.assert_allclose(got, g["volume"], rtol=RTOL, atol=ATOL)
    assert np.array_equal(got, co.backproject(g["proj"], poses, shape))
    a, b = 3, shape[0] - 2
    slab = ops.backproject(T(g["proj"], dev), poses, shape, d0=a, d1=b).cpu().numpy()
    assert np.array_equal(slab, got[:, :, a:b])
    from liftreg_amd.utils import sdct_projection_utils as S
    grid = S.backproj_grids_with_poses(g["poses"][0:1], shape, pshape, device=torch.device("cuda"))
    assert np.array_equal(grid.cpu().numpy(), g["grid"])                           # API-parity grid, every bit


def test_backproject_into_concat_buffer_and_ragged(ops, dev):
    rs = np.random.RandomState(9)
    # (last case: a detector much smaller than the volume → whole tiles whose shadows miss it)
    for (D, W, H), (Pw, Ph), P, B in (((20, 18, 24), (22, 26), 2, 3), ((9, 7, 11), (13, 5), 4, 2), ((16, 16, 16), (16, 16), 11, 9),
                                      ((40, 16, 24), (6, 5), 2, 2),
                                      # rows longer than 1024 voxels and a detector too wide for the LDS tile: the
                                      # direct-gather kernel instead of the tiled one
                                      ((2, 3, 1100), (8, 10), 1, 1), ((4, 4, 8), (6, 1000), 2, 1), ((3, 2, 1028), (5, 7), 1, 2)):
        proj = rs.uniform(-1, 1, (B, P, Pw, Ph)).astype(np.float32)
        poses = ro.scan_poses(30, P, W).astype(np.float32)
        want = co.backproject(proj, poses, (D, W, H))
        want_t = ro.backproject(torch.from_numpy(proj), poses[None], (D, W, H)).numpy()
        np.testing.assert_allclose(want, want_t, rtol=RTOL, atol=ATOL)
        buf = torch.full((B, P + 1, D, W, H), -7.0, device=dev)
        ops.backproject(T(proj, dev), poses, (D, W, H), out=buf[:, 1:], out_batch_stride=(P + 1) * D * W * H)
        assert np.array_equal(buf[:, 1:].cpu().numpy(), want)
        assert bool((buf[:, 0] == -7.0).all())                                     # channel 0 untouched


def test_backproject_batch_chunks_and_side_by_side_planes_keep_the_bits(ops, dev, monkeypatch):
    """lr_backproject_f32's tiled kernel deals the batch to blocks in chunks (blockIdx.y) and, for rows shorter than 256
    voxels, works on 2 or 4 planes of its tile side by side: every split (LIFTREG_BP_CHUNK, LIFTREG_BP_JP) writes the bits of
    the C oracle — incl. the reference's shipped row length 160 (JP = 2: 320 threads), ragged tiles and a chunk that does not
    divide the batch."""
    rs = np.random.RandomState(31)
    for (D, W, H), (Pw, Ph), P, B in (((11, 10, 160), (18, 240), 4, 7), ((9, 6, 96), (12, 100), 2, 5), ((10, 9, 40), (14, 44), 3, 4),
                                      ((8, 5, 256), (12, 256), 2, 3)):
        proj = rs.uniform(-1, 1, (B, P, Pw, Ph)).astype(np.float32)
        poses = ro.scan_poses(30, P, W).astype(np.float32)
        want = co.backproject(proj, poses, (D, W, H))
        for chunk in (None, "0", "1", "3"):
            for jp in (None, "1", "2", "4"):
                for name, v in (("LIFTREG_BP_CHUNK", chunk), ("LIFTREG_BP_JP", jp)):
                    if v is None:
                        monkeypatch.delenv(name, raising=False)
                    else:
                        monkeypatch.setenv(name, v)
                got = ops.backproject(T(proj, dev), poses, (D, W, H)).cpu().numpy()
                assert np.array_equal(got, want), ((D, W, H), chunk, jp)
    monkeypatch.delenv("LIFTREG_BP_CHUNK", raising=False)
    monkeypatch.delenv("LIFTREG_BP_JP", raising=False)


# ------------------------------------------------------------------------------------- K6/K7 warp
@pytest.mark.parametrize("tag", ["warp_a", "warp_b"])
def test_warp_golden(golden, ops, dev, tag):
    from liftreg_amd.utils import net_utils as N
    g = golden(tag)
    shape = g["img"].shape[2:]
    assert np.array_equal(N.identity_map(shape, device=dev).cpu().numpy(), g["identity"])
    assert np.array_equal(N.gen_identity_map(list(shape), 1.0, device=dev).cpu().numpy(), g["identity"])
    ids = [T(t, dev) for t in N.identity_axis_tables(shape)]
    img, disp, phi_in = T(g["img"], dev), T(g["disp"], dev), T(g["phi"], dev)
    cases = {"warped_zeros_scale": (True, True, "bilinear", co.USING_SCALE),
             "warped_border_scale": (False, True, "bilinear", co.USING_SCALE | co.BORDER),
             "warped_zeros_noscale": (True, False, "bilinear", 0),
             "warped_nearest": (True, True, "nearest", co.USING_SCALE | co.NEAREST)}
    for key, (zb, sc, mode, flags) in cases.items():
        phi, w = ops.warp(img, disp, ids, None, using_scale=sc, zero_boundary=zb, mode=mode)
        assert np.array_equal(phi.cpu().numpy(), g["phi"]), key
        np.testing.assert_allclose(w.cpu().numpy(), g[key], rtol=RTOL, atol=ATOL, err_msg=key)
        _, cw = co.warp(g["img"], g["disp"], ids=N.identity_axis_tables(shape), flags=flags)
        assert np.array_equal(w.cpu().numpy(), cw), key                            # same op order as C oracle
        out = N.Bilinear(zero_boundary=zb, using_scale=sc, mode=mode)(img, phi_in)  # the reference's module API
        assert np.array_equal(out.cpu().numpy(), w.cpu().numpy()), key
    _, w = ops.warp(img, disp, ids, None)
    a, b = 2, shape[0] - 3
    sphi, sw = ops.warp(img, disp[:, :, a:b].contiguous(), (ids[0][a:b].contiguous(), ids[1], ids[2]), None, d0=a, d1=b)
    assert np.array_equal(sw.cpu().numpy(), w[:, :, a:b].cpu().numpy())            # z-slab of the output


def test_warp_seg_ragged_and_multichannel(ops, dev):
    rs = np.random.RandomState(3)
    from liftreg_amd.utils import net_utils as N
    for shape, B, C in (((9, 11, 13), 2, 1), ((8, 8, 12), 1, 3)):
        img = rs.uniform(-1, 1, (B, C) + shape).astype(np.float32)
        seg = (rs.uniform(0, 1, (B, C) + shape) > 0.3).astype(np.float32)
        disp = rs.normal(0, 0.2, (B, 3) + shape).astype(np.float32)
        tabs = N.identity_axis_tables(shape)
        cphi, cw = co.warp(img, disp, ids=tabs, seg=seg, flags=co.USING_SCALE)
        phi, w = ops.warp(T(img, dev), T(disp, dev), [T(t, dev) for t in tabs], T(seg, dev))
        assert np.array_equal(phi.cpu().numpy(), cphi) and np.array_equal(w.cpu().numpy(), cw)
        ref = ro.warp((torch.from_numpy(img) + 1) * torch.from_numpy(seg) - 1, torch.from_numpy(cphi))
        np.testing.assert_allclose(w.cpu().numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)
        mc = ops.mask_compose(T(img, dev), T(seg, dev)).cpu().numpy()
        assert np.array_equal(mc, co.mask_compose(img, seg))


def test_warp_fast_kernel_equals_general_kernel_on_hostile_input(ops, dev, monkeypatch):
    """The model's case runs `warp_tri_fast_kernel` (buffer-load taps, weights carry the range checks); every
    other flag combination runs the general kernel.  Same bits on samples outside every face, NaN coordinates, and
    NaN/Inf voxels sitting where a dropped tap would have been clamped to."""
    from liftreg_amd.utils import net_utils as N
    rs = np.random.RandomState(11)
    for shape, B, C in (((6, 7, 8), 2, 1), ((5, 9, 16), 1, 2), ((12, 10, 260), 1, 1)):
        img = rs.uniform(-1, 1, (B, C) + shape).astype(np.float32)
        img[:, :, 0, 0, 0] = np.inf                       # face voxels: reached only through clamped (dropped) taps
        img[:, :, -1, -1, -1] = np.nan
        img[:, :, :, :, 1] = rs.choice([np.nan, np.inf, 1.0], size=img[:, :, :, :, 1].shape).astype(np.float32)
        disp = rs.normal(0, 0.9, (B, 3) + shape).astype(np.float32)   # a third of the samples leave the volume
        disp[0, 0, 1, 1, 0] = np.nan
        disp[0, 2, 2, 2, 3] = np.inf
        disp[0, 1, 0, 0, 1] = -np.inf
        tabs = N.identity_axis_tables(shape)
        dt = [T(t, dev) for t in tabs]
        for sc in (True, False):
            monkeypatch.delenv("LIFTREG_WARP_GENERAL", raising=False)
            phi_f, w_f = ops.warp(T(img, dev), T(disp, dev), dt, None, using_scale=sc)
            monkeypatch.setenv("LIFTREG_WARP_GENERAL", "1")
            phi_g, w_g = ops.warp(T(img, dev), T(disp, dev), dt, None, using_scale=sc)
            monkeypatch.delenv("LIFTREG_WARP_GENERAL", raising=False)
            a, b = w_f.cpu().numpy(), w_g.cpu().numpy()
            assert np.array_equal(phi_f.cpu().numpy(), phi_g.cpu().numpy(), equal_nan=True)
            assert np.array_equal(np.isnan(a), np.isnan(b)), (shape, sc)
            assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), (shape, sc)
        clean = np.nan_to_num(img, nan=0.5, posinf=0.25, neginf=-0.25)
        _, cw = co.warp(clean, np.nan_to_num(disp, nan=0.0, posinf=3.0, neginf=-3.0), ids=tabs, flags=co.USING_SCALE)
        _, w = ops.warp(T(clean, dev), T(np.nan_to_num(disp, nan=0.0, posinf=3.0, neginf=-3.0), dev), dt, None)
        assert np.array_equal(w.cpu().numpy(), cw), shape                         # and the C oracle, finite input


def test_pca_warp_one_pass_equals_two_kernels(ops, dev):
    """lr_pca_warp_f32 (PCA reconstruction + identity + warp, displacement never re-read) writes the same
    bits as lr_pca_reconstruct_f32 followed by lr_warp_trilinear_f32 — fp32 and bf16-stored basis, batches above the
    kernel's 8 rows (chunked), displacements that leave the volume, with and without the intensity rescale."""
    from liftreg_amd.utils import net_utils as N
    rs = np.random.RandomState(21)
    for shape, B, C, Lat, scale in (((6, 7, 8), 3, 1, 5, 0.4), ((9, 5, 20), 11, 1, 7, 0.15), ((4, 6, 260), 2, 2, 3, 0.05)):
        V = int(np.prod(shape))
        img = T(rs.uniform(-1, 1, (B, C) + shape).astype(np.float32), dev)
        basis = T(rs.normal(0, scale, (Lat, 3 * V)).astype(np.float32), dev)
        mean = T(rs.normal(0, 0.02, 3 * V).astype(np.float32), dev)
        coefs = T(rs.normal(0, 1, (B, Lat)).astype(np.float32), dev)
        ids = [T(t, dev) for t in N.identity_axis_tables(shape)]
        for bs in (basis, basis.to(torch.bfloat16)):
            for sc in (True, False):
                assert ops.pca_warp_supported(coefs, bs, img)
                d1, p1, w1 = ops.pca_warp(coefs, bs, mean, ids, img, using_scale=sc)
                d2 = ops.pca_reconstruct(coefs, bs, mean).view(B, 3, *shape)
                p2, w2 = ops.warp(img, d2, ids, None, using_scale=sc)
                assert torch.equal(d1, d2) and torch.equal(p1, p2) and torch.equal(w1, w2), (shape, bs.dtype, sc)
        cw = co.warp(img.cpu().numpy(), d2.cpu().numpy(), ids=N.identity_axis_tables(shape), flags=0)[1]
        assert np.array_equal(w1.cpu().numpy(), cw)                                 # and the C oracle (last: no rescale)
    odd = T(rs.uniform(-1, 1, (1, 1, 4, 4, 6)).astype(np.float32), dev)             # H % 4 != 0: the caller falls back
    assert not ops.pca_warp_supported(T(np.zeros((1, 2), np.float32), dev), T(np.zeros((2, 288), np.float32), dev), odd)


# ------------------------------------------------------------------------------------- K8 NCC
def test_ncc_golden_and_sharded(golden, ops, dev):
    from liftreg_amd.layers.losses import NCCLoss
    from liftreg_amd.layers.layers import NCCLoss as NCCSq
    g = golden("ncc")
    x, y = T(g["x"], dev), T(g["y"], dev)
    assert abs(float(NCCLoss()(x, y)) - float(g["loss_configured"])) < 2e-6
    assert abs(float(NCCSq()(x, y)) - float(g["loss_squared"])) < 2e-6
    m = ops.ncc_moments(x, y, 3).cpu().numpy()
    np.testing.assert_allclose(m, co.ncc_moments(g["x"], g["y"], 3), rtol=1e-12)
    # moments of z-slabs add (the 5-moment all-reduce of SURVEY §8e)
    xs, ys = x.reshape(3, -1), y.reshape(3, -1)
    cut = 300
    m2 = ops.ncc_moments(xs[:, :cut].contiguous(), ys[:, :cut].contiguous(), 3) + \
        ops.ncc_moments(xs[:, cut:].contiguous(), ys[:, cut:].contiguous(), 3)
    np.testing.assert_allclose(m2.cpu().numpy(), m, rtol=1e-12)
    loss, rows = ops.ncc_loss_from_moments(m2, xs.shape[1], 3)
    assert abs(float(loss) - float(g["loss_configured"])) < 2e-6
    assert abs(float(NCCLoss()(x, x))) < 1e-6                                       # NCC(x,x) = 1


# ------------------------------------------------------------------------------------- K3/K4 conv, fc
@pytest.mark.parametrize("tag", ["conv_a", "conv_b", "conv_c"])
def test_conv_golden_all_layouts(golden, ops, dev, tag):
    g = golden(tag)
    s = int(g["stride"])
    x, w, b = T(g["x"], dev), T(g["weight"], dev), T(g["bias"], dev)
    y = ops.conv3d_k3_lrelu(x, w, b, s)                                            # NCDHW → NCDHW
    np.testing.assert_allclose(y.cpu().numpy(), g["y"], rtol=1e-4, atol=1e-5)
    y_cl = ops.conv3d_k3_lrelu(x, w, b, s, out_layout=ops.LAYOUT_NDHWC)             # NCDHW → NDHWC
    assert np.array_equal(y_cl.permute(0, 4, 1, 2, 3).cpu().numpy(), y.cpu().numpy())
    if g["x"].shape[1] % 4 == 0:
        x_cl = x.permute(0, 2, 3, 4, 1).contiguous()
        y2 = ops.conv3d_k3_lrelu(x_cl, w, b, s, in_layout=ops.LAYOUT_NDHWC, out_layout=ops.LAYOUT_NDHWC)
        np.testing.assert_allclose(y2.permute(0, 4, 1, 2, 3).cpu().numpy(), g["y"], rtol=1e-4, atol=1e-5)
        y3 = ops.conv3d_k3_lrelu(x_cl, w, b, s, in_layout=ops.LAYOUT_NDHWC, out_layout=ops.LAYOUT_NCDHW)
        assert np.array_equal(y3.cpu().numpy(), y2.permute(0, 4, 1, 2, 3).cpu().numpy())


def test_conv_parity_split_layout(ops, dev, monkeypatch):
    """LAYOUT_NDHWC_HPS (even voxels of a row, then odd) as a block's output and as a stride-2 block's input
    gives bit-identical results to plain NDHWC — with the direct stride-2 walk (LIFTREG_CONV_DIRECT=1: the oracle's
    fmaf chain; the default Winograd rows kernel has its own test, test_winograd_rows_kernel_matches_direct)."""
    monkeypatch.setenv("LIFTREG_CONV_DIRECT", "1")
    rs = np.random.RandomState(21)
    for (D, W, H), B in (((10, 9, 20), 2), ((7, 8, 34), 1), ((16, 16, 64), 1)):
        x = T(rs.uniform(-1, 1, (B, 3, D, W, H)).astype(np.float32), dev)
        w0 = T((rs.normal(0, 1, (16, 3, 3, 3, 3)) / 9).astype(np.float32), dev)
        b0 = T(rs.uniform(-0.1, 0.1, 16).astype(np.float32), dev)
        w1 = T((rs.normal(0, 1, (32, 16, 3, 3, 3)) / 20).astype(np.float32), dev)
        b1 = T(rs.uniform(-0.1, 0.1, 32).astype(np.float32), dev)
        y = ops.conv3d_k3_lrelu(x, w0, b0, 1, out_layout=ops.LAYOUT_NDHWC)
        y_ps = ops.conv3d_k3_lrelu(x, w0, b0, 1, out_layout=ops.LAYOUT_NDHWC_HPS)
        assert torch.equal(ops.hps_to_ndhwc(y_ps), y)
        z = ops.conv3d_k3_lrelu(y, w1, b1, 2, in_layout=ops.LAYOUT_NDHWC, out_layout=ops.LAYOUT_NDHWC)
        z_ps = ops.conv3d_k3_lrelu(y_ps, w1, b1, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=ops.LAYOUT_NDHWC)
        assert torch.equal(z_ps, z)
        if z.shape[3] % 2 == 0:                                                    # 32-channel rows: [2 blocks][parity][H/2][16]
            w2 = T((rs.normal(0, 1, (32, 32, 3, 3, 3)) / 29).astype(np.float32), dev)
            z_hps = ops.conv3d_k3_lrelu(y_ps, w1, b1, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=ops.LAYOUT_NDHWC_HPS)
            assert torch.equal(ops.hps_to_ndhwc(z_hps), z)
            q = ops.conv3d_k3_lrelu(z, w2, b1, 2, in_layout=ops.LAYOUT_NDHWC, out_layout=ops.LAYOUT_NCDHW)
            q_ps = ops.conv3d_k3_lrelu(z_hps, w2, b1, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=ops.LAYOUT_NCDHW)
            assert torch.equal(q_ps, q)
    from liftreg_amd import _hip
    with pytest.raises(_hip.LiftRegHipError):                                     # odd H cannot be parity-split
        ops.conv3d_k3_lrelu(torch.zeros(1, 4, 4, 5, 16, device=dev), w1, b1, 2, in_layout=ops.LAYOUT_NDHWC_HPS)


def _to_hps(x):
    """(B,D,W,H,C) channels-last -> LAYOUT_NDHWC_HPS rows [C/16][parity of h][H/2][16] (H even, C % 16 == 0)."""
    B, D, W, H, C = x.shape
    h = torch.arange(H, device=x.device)
    inv = torch.empty(H, dtype=torch.long, device=x.device)
    inv[(h & 1) * (H // 2) + (h >> 1)] = h
    return x.reshape(B, D, W, H, C // 16, 16)[:, :, :, inv].permute(0, 1, 2, 4, 3, 5).reshape(B, D, W, H, C).contiguous()


def test_winograd_rows_kernel_matches_direct(ops, dev, monkeypatch):
    """The default stride-2 kernel on parity-split input (conv3d_rows.hip: persistent, Winograd F(2,2) along W, fragments in
    LDS) against the direct walk (the oracle's fmaf chain) and against the C oracle itself: same fp32 arithmetic, another
    summation tree -> within 2e-5 of the activation scale (north_star: 1e-4 rel for fp32 convs).  Ragged shapes (odd D/W,
    W and Ho not multiples of the tile), every Cin/Cout combination, every output layout, few and many tiles (both work
    orders), and z_phase = 1 on a slab == the matching planes of the whole volume, bit for bit."""
    import oracle.c_oracle as co
    rs = np.random.RandomState(77)
    cases = [((9, 7, 12), 2, 16, 32), ((12, 18, 64), 1, 16, 32), ((5, 9, 34), 1, 32, 32), ((16, 16, 32), 2, 32, 16),
             ((7, 6, 20), 1, 16, 16), ((40, 72, 96), 1, 16, 32), ((24, 40, 64), 2, 32, 32), ((9, 128, 130), 1, 16, 32)]
    for (D, W, H), B, ci, cout in cases:
        # the dispatcher keeps planes under 64 x 64 outputs on the direct kernels; the small cases force the rows kernel
        if ((W - 1) // 2 + 1) * ((H - 1) // 2 + 1) < 4096:
            monkeypatch.setenv("LIFTREG_CONV_ROWS_ALWAYS", "1")
        else:
            monkeypatch.delenv("LIFTREG_CONV_ROWS_ALWAYS", raising=False)
        x = T(rs.uniform(-1, 1, (B, D, W, H, ci)).astype(np.float32), dev)
        w = T((rs.normal(0, 1, (cout, ci, 3, 3, 3)) / (27 * ci) ** 0.5).astype(np.float32), dev)
        b = T(rs.uniform(-0.1, 0.1, cout).astype(np.float32), dev)
        x_ps = _to_hps(x)
        monkeypatch.setenv("LIFTREG_CONV_DIRECT", "1")
        want = ops.conv3d_k3_lrelu(x_ps, w, b, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=ops.LAYOUT_NDHWC)
        monkeypatch.delenv("LIFTREG_CONV_DIRECT")
        scale = float(want.abs().max())
        got = ops.conv3d_k3_lrelu(x_ps, w, b, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=ops.LAYOUT_NDHWC)
        assert float((got - want).abs().max()) <= 2e-5 * scale, ((D, W, H), ci, cout)
        assert not torch.equal(got, want) or D * W * H < 64          # it IS the other kernel
        g2 = ops.conv3d_k3_lrelu(x_ps, w, b, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=ops.LAYOUT_NCDHW)
        assert torch.equal(g2, got.permute(0, 4, 1, 2, 3).contiguous())
        if got.shape[3] % 2 == 0:
            g3 = ops.conv3d_k3_lrelu(x_ps, w, b, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=ops.LAYOUT_NDHWC_HPS)
            assert torch.equal(ops.hps_to_ndhwc(g3), got)
        if D * W * H * ci <= 200000:                                  # the C oracle (scalar loops) on the small cases
            ref = co.conv3d_k3_lrelu(x.permute(0, 4, 1, 2, 3).contiguous().cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), 2, 0.2)
            assert np.abs(got.permute(0, 4, 1, 2, 3).cpu().numpy() - ref).max() <= 2e-5 * scale
        # a z-slab starting at global input plane 2: local output plane k = global plane k + 1, so local plane 0 is an
        # ODD global plane -> z_phase = 1 reproduces the whole-volume bits (what the sharded model relies on)
        if D >= 7:
            sl = _to_hps(x[:, 2:].contiguous())
            got_sl = ops.conv3d_k3_lrelu(sl, w, b, 2, in_layout=ops.LAYOUT_NDHWC_HPS, out_layout=ops.LAYOUT_NDHWC, z_phase=1)
            # interior planes (local k >= 1: no zero padding below) equal global planes k + 1 bit for bit
            assert torch.equal(got_sl[:, 1:], got[:, 2:]), ((D, W, H), ci, cout)


def test_first_block_split_input_equals_concatenated(ops, dev):
    """lr_conv3d_first_split_f32 reads channel 0 (the moving image) and the backprojected views from their own buffers:
    same bits as the block on torch.cat([...], 1), both output layouts, ragged bricks, 2 and 3 channels."""
    rs = np.random.RandomState(31)
    for (D, W, H), B, P in (((9, 7, 24), 2, 2), ((4, 5, 68), 1, 1), ((16, 16, 64), 1, 2)):
        x0 = T(rs.uniform(-1, 1, (B, 1, D, W, H)).astype(np.float32), dev)
        rest = T(rs.uniform(-1, 1, (B, P, D, W, H)).astype(np.float32), dev)
        w = T((rs.normal(0, 1, (16, P + 1, 3, 3, 3)) / 9).astype(np.float32), dev)
        b = T(rs.uniform(-0.1, 0.1, 16).astype(np.float32), dev)
        assert ops.conv3d_first_split_supported(x0, rest)
        for lay in (ops.LAYOUT_NDHWC, ops.LAYOUT_NDHWC_HPS, ops.LAYOUT_NCDHW):
            want = ops.conv3d_k3_lrelu(torch.cat([x0, rest], 1), w, b, 1, out_layout=lay)
            got = ops.conv3d_first_split(x0, rest, w, b, out_layout=lay)
            assert torch.equal(got, want), ((D, W, H), P, lay)
    assert not ops.conv3d_first_split_supported(x0[..., :62].contiguous(), rest[..., :62].contiguous())   # H % 4 != 0


def test_conv_medium_vs_oracle(ops, dev):
    rs = np.random.RandomState(11)
    torch.manual_seed(11)
    for cin, cout, s, shape, B in ((3, 16, 1, (20, 33, 37), 2), (12, 16, 1, (10, 9, 18), 1), (16, 32, 2, (24, 17, 40), 2),
                                   (32, 32, 2, (16, 16, 16), 3), (32, 32, 2, (4, 4, 4), 2), (8, 32, 1, (6, 6, 20), 1)):
        x = rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32)
        w = (rs.normal(0, 1, (cout, cin, 3, 3, 3)) / np.sqrt(27 * cin)).astype(np.float32)
        b = rs.uniform(-0.1, 0.1, cout).astype(np.float32)
        want = ro.conv_block(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s).numpy()
        got = ops.conv3d_k3_lrelu(T(x, dev), T(w, dev), T(b, dev), s).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5, err_msg=str((cin, cout, s, shape)))
        if cin % 4 == 0:
            xcl = T(x, dev).permute(0, 2, 3, 4, 1).contiguous()
            got2 = ops.conv3d_k3_lrelu(xcl, T(w, dev), T(b, dev), s, in_layout=ops.LAYOUT_NDHWC).cpu().numpy()
            np.testing.assert_allclose(got2, want, rtol=1e-4, atol=1e-5, err_msg="cl " + str((cin, cout, s, shape)))


def test_fc_golden_and_sizes(golden, ops, dev):
    g = golden("fc")
    h1 = ops.linear_lrelu(T(g["x"], dev), T(g["w1"], dev), T(g["b1"], dev), 0.2)
    h2 = ops.linear_lrelu(h1, T(g["w2"], dev), T(g["b2"], dev), 1.0)
    np.testing.assert_allclose(h1.cpu().numpy(), g["h1"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(h2.cpu().numpy(), g["h2"], rtol=1e-5, atol=2e-6)
    rs = np.random.RandomState(2)
    for B, K, O in ((8, 16384, 800), (1, 4000, 800), (13, 801, 57), (32, 256, 56)):
        x = rs.uniform(-1, 1, (B, K)).astype(np.float32)
        w = (rs.normal(0, 1, (O, K)) / np.sqrt(K)).astype(np.float32)
        b = rs.uniform(-0.1, 0.1, O).astype(np.float32)
        want = co.linear_lrelu(x, w, b, 0.2)
        got = ops.linear_lrelu(T(x, dev), T(w, dev), T(b, dev), 0.2).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------- K5 PCA
def test_pca_vs_oracle(ops, dev):
    rs = np.random.RandomState(4)
    for B, L, M in ((8, 56, 3 * 16 ** 3), (3, 6, 3 * 12 * 10 * 14 // 4 * 4), (16, 9, 4096), (1, 56, 1024)):
        coefs = rs.normal(0, 1, (B, L)).astype(np.float32)
        basis = rs.normal(0, 0.02, (L, M)).astype(np.float32)
        mean = rs.normal(0, 0.01, M).astype(np.float32)
        want = co.pca_reconstruct(coefs, basis, mean)
        got = ops.pca_reconstruct(T(coefs, dev), T(basis, dev), T(mean, dev)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7)
        half = M // 2 // 4 * 4                                                      # a column slab of the basis
        got_s = ops.pca_reconstruct(T(coefs, dev), T(basis, dev)[:, half:], T(mean, dev)[half:].contiguous())
        assert np.array_equal(got_s.cpu().numpy(), got[:, half:])


# ------------------------------------------------------------------------------------- a14 whole model
def test_model_forward_golden(golden, dev, tmp_path):
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    from liftreg_amd.utils.general import get_class
    g = golden("model_32")
    vec, mean = pca_basis_32(int(g["latent_dim"]), 32, int(g["pca_seed"]))
    np.save(tmp_path / "pca_vectors.npy", vec)
    np.save(tmp_path / "pca_mean.npy", mean)
    cls = get_class("liftreg_amd.models.LiftRegDeformSubspaceBackproj.model")       # the plugin path
    assert cls is model
    net = cls([32, 32, 32], {"drr_feature_num": 2, "latent_dim": int(g["latent_dim"]), "pca_path": str(tmp_path)})
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())                      # the reference's 19 keys
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    inp = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in::")}
    inp = {k: (v.to(dev) if v.dim() > 3 else v) for k, v in inp.items()}             # set_input: ndim>3 → GPU
    with torch.no_grad():
        out = net(inp)
    assert set(out) == {"warped", "phi", "params", "target", "pca_coefs", "target_proj", "warped_proj"}
    np.testing.assert_allclose(out["pca_coefs"].cpu().numpy(), g["out::pca_coefs"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["params"].cpu().numpy(), g["out::params"], rtol=1e-4, atol=1e-6)    # displacement
    np.testing.assert_allclose(out["phi"].cpu().numpy(), g["out::phi"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out["warped"].cpu().numpy(), g["out::warped"], rtol=1e-4, atol=2e-5)
    assert np.array_equal(out["target"].cpu().numpy(), g["out::target"])
    assert out["warped_proj"] is out["target_proj"]
    out2 = net(inp)                                  # with grad enabled: graph of HIP Functions only.  The training forward runs the
    # two first blocks through the same fused split-operand pair kernel as inference (csrc/conv01_fused.hip, in the form that also
    # writes block 0's activation and sign mask for the backward): the same bits end to end
    np.testing.assert_allclose(out2["params"].detach().cpu().numpy(), g["out::params"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out2["warped"].detach().cpu().numpy(), g["out::warped"], rtol=1e-4, atol=2e-5)
    assert torch.equal(out2["pca_coefs"], out["pca_coefs"]) and torch.equal(out2["warped"], out["warped"])
    assert out2["warped"].grad_fn is not None
    net.fuse_pair01 = False                          # one fp32-MFMA kernel per block in both modes: again the same bits,
    with torch.no_grad():                            # within fp32 rounding of the pair kernel's
        out3 = net(inp)
    out4 = net(inp)
    net.fuse_pair01 = True
    assert torch.equal(out4["warped"], out3["warped"])
    assert float((out3["warped"] - out["warped"]).abs().max()) <= 2e-5
    assert type(out2["warped"].grad_fn).__name__ == "DecodeFnBackward"


# ------------------------------------------------------------------------------------- error behaviour
def test_no_cpu_fallback_and_error_codes(ops, dev):
    from liftreg_amd import _hip
    with pytest.raises(_hip.LiftRegHipError):
        ops.backproject(torch.zeros(1, 2, 4, 4), np.zeros((2, 3), np.float32), (4, 4, 4))   # CPU tensor
    lib = _hip.lib()
    assert lib.lr_backproject_f32(None, None, None, 1, 1, 4, 4, 4, 4, 4, 0, 4, 64, None) == -2   # LR_ENULL
    x = torch.zeros(1, 3, 4, 4, 4, device=dev)
    with pytest.raises(_hip.LiftRegHipError):
        ops.conv3d_k3_lrelu(x, torch.zeros(8, 3, 3, 3, 3, device=dev), None, 1)                  # Cout=8 unsupported
    assert lib.lr_strerror(-3) == b"combination not built into this library"
    # every later entry point keeps the contract: negative code, no exception, nothing launched
    ENULL, EINVAL, EUNSUP = -2, -1, -3
    z = torch.zeros(4096, device=dev)
    p = z.data_ptr()
    assert lib.lr_normalize_clip_f32(None, p, 16, 0.0, 1.0, None) == ENULL
    assert lib.lr_normalize_clip_f32(p, p, 16, 1.0, 1.0, None) == EINVAL                      # empty clip range
    assert lib.lr_label_overlap_f32(p, p, 1.0, 16, None, 4, p, None) == ENULL
    assert lib.lr_jacobi_det_stats_f32(p, 1, 2, 2, 2, 0.0, 1.0, 1.0, p, 1, p, None) == EINVAL  # zero spacing
    assert lib.lr_hu_to_mu_f32(None, p, 4, None) == ENULL
    assert lib.lr_conv3d_k3_lrelu_bf16(p, p, None, p, 1, 16, 32, 4, 4, 4, 1, 3, 3, 0.2, None) == EUNSUP   # stride 1
    assert lib.lr_conv3d_k3_lrelu_bf16(p, p, None, p, 1, 16, 32, 4, 4, 4, 2, 1, 3, 0.2, None) == EINVAL   # fp32 layout id
    assert lib.lr_conv3d_k3_lrelu_bf16(p, p, None, p, 1, 16, 32, 4, 4, 5, 2, 4, 3, 0.2, None) == EUNSUP   # parity split, odd H
    assert lib.lr_conv3d_first_bf16(p, p, None, p, 1, 3, 8, 4, 4, 4, 3, 0.2, None) == EUNSUP              # Cout = 8
    assert lib.lr_conv3d_packed_bf16_bytes(8, 16) == EUNSUP
    assert lib.lr_conv3d_dgrad_f32(p, p, p, 1, 32, 16, 4, 4, 4, 1, 1, None, 1, 0.2, None) == EUNSUP       # stride 1
    assert lib.lr_conv3d_dgrad_f32(p, p, p, 1, 32, 16, 4, 4, 4, 2, 0, None, 1, 0.2, None) == EINVAL       # planar gx
    assert lib.lr_conv3d_dgrad_f32(p, p, p, 1, 32, 16, 4, 4, 4, 2, 1, p, 0, 0.2, None) == EINVAL          # planar mask source
    assert lib.lr_conv3d_wgrad_f32(p, 1, p, None, p, None, 1, 16, 32, 4, 4, 4, 2, 8, None) == ENULL       # no workspace
    assert lib.lr_conv3d_wgrad_f32(p, 1, p, p, p, None, 1, 16, 8, 4, 4, 4, 2, 8, None) == EUNSUP          # Cout = 8
    assert lib.lr_conv3d_wgrad_f32(p, 3, p, p, p, None, 1, 20, 16, 4, 4, 4, 2, 8, None) == EUNSUP         # bf16 x, Cin = 20
    assert lib.lr_disp_reg_bwd_f32(p, None, p, 1, 2, 2, 2, None) == ENULL
    assert lib.lr_pca_bwd_coef_f32(None, p, p, p, 1, 1, 4, 4, 4, 1, None) == ENULL
    torch.cuda.synchronize()                                                               # and the device is still healthy
    assert float(z.sum()) == 0.0


# ------------------------------------------------------------------------------------- f3/f4: file pipeline
def test_preprocessing_drr_cli_matches_reference_files(dev, tmp_path):
    """The reference's tools/preprocessingDRR.py flow on a tiny dataset: same folder layout, same files."""
    from liftreg_amd.tools import preprocessingDRR as tool
    from liftreg_amd.utils.utils import save_deformations
    rs = np.random.RandomState(8)
    root = tmp_path / "data"
    (root / "preprocessed").mkdir(parents=True)
    shape = (12, 10, 14)
    vols = {}
    for phase, ids in (("train", ["a1", "a2"]), ("val", ["v1"])):
        (root / phase).mkdir()
        np.save(root / phase / "data_id.npy", np.array(ids))
        for d in ids:
            for kind in ("source", "target"):
                vols[(d, kind)] = np.clip(rs.normal(-500, 400, shape), -1024, 1000).astype(np.float32)
                np.save(root / "preprocessed" / f"{d}_{kind}.npy", vols[(d, kind)])
    assert tool.main(["-d", str(root), "--drr_folder_name", "t", "--scan_range", "30", "--scan_num", "3",
                      "--receptor_w", "11", "--receptor_h", "9"]) == 0
    out = root / "drr" / "t" / "drr"
    poses = np.load(out / "poses.npy")
    assert np.array_equal(poses, ro.scan_poses(30, 3, shape[1])) and poses.dtype == np.float64
    for (d, kind), hu in vols.items():
        got = np.load(out / f"{d}_{kind}_proj.npy")
        want = ro.drr_forward(ro.calc_relative_atten_coef(np.flip(hu, axis=1)), poses, (9, 11), (2.2, 2.2, 2.2))
        assert got.dtype == np.float32 and got.shape == (3, 9, 11)
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
    phi = torch.from_numpy(rs.uniform(-1, 1, (2, 3, 4, 5, 6)).astype(np.float32)).to(dev)
    save_deformations(phi, ["x", "y"], str(tmp_path))
    assert np.array_equal(np.load(tmp_path / "y_phi.npy"), ((phi[1].cpu().numpy() + 1.) / 2.).astype(np.float32))


def test_register_folder_tool(dev, tmp_path):
    """liftreg_amd.tools.register_folder over a tiny dataset in the reference's folder layout: the saved {id}_phi.npy
    equal the model run on inputs prepared the way the reference's dataset does (flip axis 1, clip-range
    normalisation — restated with numpy), checkpoint loaded through the reference's key names."""
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    from liftreg_amd.tools import preprocessingDRR, register_folder
    rs = np.random.RandomState(12)
    root, out = tmp_path / "data", tmp_path / "out"
    (root / "preprocessed").mkdir(parents=True)
    (root / "test").mkdir()
    shape, ids = (16, 12, 20), ["c1", "c2", "c3"]
    np.save(root / "test" / "data_id.npy", np.array(ids))
    vols = {}
    for c in ids:
        for kind in ("source", "target"):
            vols[(c, kind)] = np.clip(rs.normal(-500, 400, shape), -1024, 1000).astype(np.float32)
            np.save(root / "preprocessed" / f"{c}_{kind}.npy", vols[(c, kind)])
            np.save(root / "preprocessed" / f"{c}_{kind}_seg.npy", (rs.uniform(0, 1, shape) > 0.4).astype(np.float32))
    assert preprocessingDRR.main(["-d", str(root), "--drr_folder_name", "t", "--scan_range", "30", "--scan_num", "2",
                                  "--receptor_w", "20", "--receptor_h", "16", "--phase", "test"]) == 0
    torch.manual_seed(4)
    net = model(list(shape), {"drr_feature_num": 2, "latent_dim": 6, "pca_path": "synthetic:3"}).to(dev).eval()
    torch.save({"state_dict": net.state_dict(), "epoch": 1}, tmp_path / "ck.pth.tar")
    assert register_folder.main(["-d", str(root), "--drr_folder_name", "t", "--phase", "test", "-o", str(out), "--pca_path",
                                 "synthetic:3", "--checkpoint", str(tmp_path / "ck.pth.tar"), "--latent_dim", "6",
                                 "--batch", "2", "--labels"]) == 0
    drr = root / "drr" / "t" / "drr"
    poses = np.load(drr / "poses.npy").astype(np.float32)
    for c in ids:
        prep = lambda a, lo, hi: torch.from_numpy(ro.normalize_clip(a, lo, hi))
        flip = lambda a: np.flip(a, axis=1).copy()
        inp = {"source": prep(flip(vols[(c, "source")]), -1000, 0)[None, None].to(dev),
               "target": prep(flip(vols[(c, "target")]), -1000, 0)[None, None].to(dev),
               "target_proj": prep(np.load(drr / f"{c}_target_proj.npy"), 0, 6)[None].to(dev),
               "source_label": torch.from_numpy(flip(np.load(root / "preprocessed" / f"{c}_source_seg.npy")))[None, None].to(dev),
               "target_label": torch.from_numpy(flip(np.load(root / "preprocessed" / f"{c}_target_seg.npy")))[None, None].to(dev),
               "target_poses": torch.from_numpy(poses[None].copy())}
        with torch.no_grad():
            want = net(inp)["phi"][0].cpu().numpy()
        got = np.load(out / f"{c}_phi.npy")
        assert got.dtype == np.float32 and got.shape == (3,) + shape
        np.testing.assert_allclose(got, (want + 1.0) / 2.0, rtol=0, atol=1e-6)
    import json
    rep = json.load(open(out / "register_folder.json"))
    assert [r["id"] for r in rep] == ids and all(0.0 <= r["dice"] <= 1.0 for r in rep)


def test_pca_warp_slab_and_ncc_epilogue(ops, dev):
    """lr_pca_warp_slab_f32: rows [d0,d1) from a view into the full basis and from a rank's compact column slab give the
    rows of the unsharded decode bit for bit; the epilogue's NCC moments of the slabs add up to the whole volume's
    (≤1e-12 relative, fp64) — ragged planes (W·H/4 not a multiple of the block), batches of 1..9, both basis dtypes."""
    from liftreg_amd.utils import net_utils as N
    rs = np.random.RandomState(33)
    for shape, B, Lat in (((8, 6, 12), 3, 5), ((6, 5, 20), 9, 4), ((5, 36, 32), 1, 3), ((12, 8, 8), 5, 6)):
        D, W, H = shape
        V = int(np.prod(shape))
        img = T(rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32), dev)
        tgt = T(rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32), dev)
        basis32 = T(rs.normal(0, 0.2, (Lat, 3 * V)).astype(np.float32), dev)
        mean = T(rs.normal(0, 0.02, 3 * V).astype(np.float32), dev)
        coefs = T(rs.normal(0, 1, (B, Lat)).astype(np.float32), dev)
        ids = [T(t, dev) for t in N.identity_axis_tables(shape)]
        for basis in (basis32, basis32.to(torch.bfloat16)):
            d_ref, p_ref, w_ref = ops.pca_warp(coefs, basis, mean, ids, img)
            d_f, p_f, w_f, m_full = ops.pca_warp(coefs, basis, mean, ids, img, target=tgt)
            assert torch.equal(d_f, d_ref) and torch.equal(p_f, p_ref) and torch.equal(w_f, w_ref)
            m_want = ops.ncc_moments(w_ref, tgt, B)
            np.testing.assert_allclose(m_full.cpu().numpy(), m_want.cpu().numpy(), rtol=1e-12, atol=1e-300)
            cuts = [0, D // 3, D // 3 + 1, D]
            m_sum = torch.zeros_like(m_full)
            for d0, d1 in zip(cuts[:-1], cuts[1:]):
                if d1 == d0:
                    continue
                plane = W * H
                cols = torch.cat([torch.arange((c * D + d0) * plane, (c * D + d1) * plane, device=dev) for c in range(3)])
                compact, cmean = basis[:, cols].contiguous(), mean[cols].contiguous()
                ts = tgt[:, :, d0:d1].contiguous()
                for bs, ms in ((basis, mean), (compact, cmean)):
                    d_s, p_s, w_s, m_s = ops.pca_warp(coefs, bs, ms, ids, img, d0=d0, d1=d1, target=ts)
                    assert torch.equal(d_s, d_ref[:, :, d0:d1]) and torch.equal(p_s, p_ref[:, :, d0:d1])
                    assert torch.equal(w_s, w_ref[:, :, d0:d1]), (shape, d0, d1, bs.shape)
                m_sum += m_s
            np.testing.assert_allclose(m_sum.cpu().numpy(), m_want.cpu().numpy(), rtol=1e-12, atol=1e-300)


def test_first_block_with_fused_backprojection_equals_two_kernels(ops, dev):
    """lr_conv3d_first_fused_bp_f32 (backprojection computed inside block 0's staging; the (B,P,D,W,H)
    feature volume of …Backproj.py:89-93 never materialised) gives the bits of lr_backproject_f32 followed by
    lr_conv3d_first_split_f32 — detectors smaller / larger than the volume (shadows falling off every edge), oblique
    emitters, one and two views, ragged bricks, both channels-last output layouts."""
    _need_experimental()
    rs = np.random.RandomState(41)
    cases = [((8, 8, 64), (8, 64), 2, 2), ((10, 6, 68), (14, 40), 2, 1), ((5, 9, 132), (6, 200), 1, 3),
             ((12, 12, 16), (30, 30), 2, 2), ((64, 64, 64), (64, 64), 2, 1)]
    for shape, pshape, P, B in cases:
        D, W, H = shape
        moving = T(rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32), dev)
        proj = T(rs.uniform(-1, 1, (B, P) + pshape).astype(np.float32), dev)
        for pose_kind in ("scan", "oblique"):
            if pose_kind == "scan":
                poses = np.stack([np.tan(np.linspace(-15, 15, P) / 180. * np.pi) * 3., np.full(P, 3.5),
                                  np.linspace(-0.2, 0.2, P)], 1).astype(np.float32) * W
            else:
                poses = (np.array([[1.7, 1.3, -0.9], [-2.2, 1.15, 1.4]], np.float32)[:P] * W)
            w = T(rs.normal(0, 0.2, (16, P + 1, 3, 3, 3)).astype(np.float32), dev)
            b = T(rs.normal(0, 0.1, 16).astype(np.float32), dev)
            tv = ops.backproject(proj, poses, shape)
            for lay in (ops.LAYOUT_NDHWC_HPS, ops.LAYOUT_NDHWC):
                assert ops.conv3d_first_fused_bp_supported(moving, proj)
                want = ops.conv3d_first_split(moving, tv, w, b, out_layout=lay)
                got = ops.conv3d_first_fused_bp(moving, proj, poses, w, b, out_layout=lay)
                assert torch.equal(got, want), (shape, pshape, P, pose_kind, lay, float((got - want).abs().max()))
    assert not ops.conv3d_first_fused_bp_supported(T(np.zeros((1, 1, 4, 4, 6), np.float32), dev), proj[:1])   # H % 4

