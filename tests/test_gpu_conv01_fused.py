"""GPU: encoder blocks 0 and 1 as ONE z-marching kernel on the bf16 matrix pipe with exact three-way bf16 operand splits
(csrc/conv01_fused.hip; replaces /root/reference/src/liftreg/layers/layers.py:365-369 for encoders[0] and encoders[1] of
…/models/LiftRegDeformSubspaceBackproj.py:95-100).

Both stages are direct convolutions whose partial products are exact; only the fp32 accumulation rounds.  The bar is the
fp32 bar: against an fp64 evaluation of the two blocks the pair kernel must be at least as close as the fp32-MFMA kernels it
replaces (direct fmaf chain, LIFTREG_CONV_DIRECT / LIFTREG_CONV0_DIRECT — the oracle's bits — and the default Winograd
kernels) and within 2e-6 of the output scale outright.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _ref64(x, w0, b0, w1, b1, s0=0.2, s1=0.2):
    y = F.leaky_relu(F.conv3d(x.double(), w0.double(), b0.double(), stride=1, padding=1), s0)
    return F.leaky_relu(F.conv3d(y, w1.double(), b1.double(), stride=2, padding=1), s1)


def _to_ncdhw(y, layout):
    from liftreg_amd import ops
    if layout == ops.LAYOUT_NDHWC_HPS:
        y = ops.hps_to_ndhwc(y)
    return y.permute(0, 4, 1, 2, 3).contiguous()


def _native(xd, w0, b0, w1, b1, layout, direct):
    """The two blocks through the fp32-MFMA kernels: direct=True = the fmaf chains of the oracle."""
    from liftreg_amd import ops
    keys = ("LIFTREG_CONV_DIRECT", "LIFTREG_CONV0_DIRECT")
    old = {k: os.environ.pop(k, None) for k in keys}
    if direct:
        for k in keys:
            os.environ[k] = "1"
    try:
        H = xd.shape[4]
        mid = ops.LAYOUT_NDHWC_HPS if H % 2 == 0 else ops.LAYOUT_NDHWC
        y = ops.conv3d_k3_lrelu(xd, w0, b0, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=mid)
        y = ops.conv3d_k3_lrelu(y, w1, b1, 2, in_layout=mid, out_layout=layout)
        torch.cuda.synchronize()
    finally:
        for k in keys:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]
    return y


def _make(B, Cin, D, W, H, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, D, W, H, generator=g)
    x[:, 0] = x[:, 0].abs() * 0.3                      # CT-like channel: non-negative, smaller scale
    w0 = torch.randn(16, Cin, 3, 3, 3, generator=g) * (2.0 / (27 * Cin)) ** 0.5
    b0 = torch.randn(16, generator=g) * 0.1
    w1 = torch.randn(32, 16, 3, 3, 3, generator=g) * (2.0 / (27 * 16)) ** 0.5
    b1 = torch.randn(32, generator=g) * 0.1
    return x, w0, b0, w1, b1


CASES = [(2, 3, 12, 32, 32, True), (1, 3, 7, 36, 44, True), (1, 2, 5, 40, 24, False), (1, 3, 9, 34, 28, False),
         (1, 2, 4, 16, 16, True), (1, 4, 6, 48, 32, True), (1, 3, 1, 32, 32, True), (2, 3, 2, 20, 36, False),
         (1, 3, 20, 128, 128, True), (1, 3, 3, 70, 132, True),
         # five channels (four views: the reference's shipped configuration): block 0 with 27 MFMAs per tile, ring 0 six planes deep
         (2, 5, 12, 32, 32, True), (1, 5, 7, 36, 44, True), (1, 5, 9, 34, 28, False), (1, 5, 1, 32, 32, True), (1, 5, 20, 80, 160, True)]


@pytest.mark.parametrize("B,Cin,D,W,H,hps", CASES)
def test_pair_kernel_is_fp32_accurate(B, Cin, D, W, H, hps):
    from liftreg_amd import ops
    x, w0, b0, w1, b1 = _make(B, Cin, D, W, H, 1000 * Cin + D + W)
    layout = ops.LAYOUT_NDHWC_HPS if hps else ops.LAYOUT_NDHWC
    if hps and ((H - 1) // 2 + 1) % 2:
        pytest.skip("parity-split output needs an even Ho")
    ref = _ref64(x, w0, b0, w1, b1)
    scale = float(ref.abs().max())
    xd, w0d, b0d, w1d, b1d = (t.to(DEV) for t in (x, w0, b0, w1, b1))

    x0 = xd[:, 0:1].contiguous()
    rest = xd[:, 1:].contiguous()
    got = _to_ncdhw(ops.conv3d_pair01(x0, rest, w0d, b0d, w1d, b1d, out_layout=layout), layout).cpu().double()
    assert got.shape == ref.shape
    assert torch.isfinite(got).all()
    e_got = float((got - ref).abs().max())
    r_got = float((got - ref).pow(2).mean().sqrt())
    msg = f"Cin {Cin} {D}x{W}x{H}: max |err| vs fp64 — pair {e_got:.3e} (rms {r_got:.3e})"
    nat_ok = H % 4 == 0 and (W * H >= 16 * 16)
    if nat_ok:
        e_nat, r_nat = [], []
        for direct in (True, False):
            nat = _to_ncdhw(_native(xd, w0d, b0d, w1d, b1d, layout, direct), layout).cpu().double()
            e_nat.append(float((nat - ref).abs().max()))
            r_nat.append(float((nat - ref).pow(2).mean().sqrt()))
        msg += f"; fmaf chain {e_nat[0]:.3e} (rms {r_nat[0]:.3e}); Winograd {e_nat[1]:.3e} (rms {r_nat[1]:.3e})"
    print(msg + f"; scale {scale:.3f}")
    assert e_got <= 2e-6 * scale
    if nat_ok:
        assert r_got <= r_nat[0] * 1.05 + 1e-9          # rms error: no worse than the direct fmaf chain
        assert e_got <= 1.5 * e_nat[0] + 1e-9


@pytest.mark.parametrize("B,Cin,D,W,H,hps", [(2, 3, 12, 32, 32, True), (1, 3, 7, 36, 44, True), (1, 2, 6, 40, 24, False),
                                             (1, 3, 4, 72, 136, True), (1, 3, 9, 34, 28, False),
                                             (2, 5, 12, 32, 32, True), (1, 5, 7, 36, 44, True), (1, 4, 6, 40, 24, False),
                                             (1, 5, 9, 34, 28, False), (1, 5, 5, 72, 136, True)])
def test_pair_kernel_training_forward_writes_activation_and_mask(B, Cin, D, W, H, hps):
    """lr_conv3d_pair01_train_f32: the same block-1 output bit for bit, plus block 0's activation (every voxel written exactly
    once, fp32-accurate against an fp64 convolution) and its sign mask in the layout lr_conv3d_k3_lrelu_mask_f32 writes."""
    from liftreg_amd import ops
    x, w0, b0, w1, b1 = _make(B, Cin, D, W, H, 77 * Cin + D + W)
    out_l = ops.LAYOUT_NDHWC_HPS if (hps and ((H - 1) // 2 + 1) % 2 == 0) else ops.LAYOUT_NDHWC
    mid_l = ops.LAYOUT_NDHWC_HPS if hps else ops.LAYOUT_NDHWC
    xd, w0d, b0d, w1d, b1d = (t.to(DEV) for t in (x, w0, b0, w1, b1))
    want1 = ops.conv3d_pair01(xd[:, 0:1].contiguous(), xd[:, 1:].contiguous(), w0d, b0d, w1d, b1d, out_layout=out_l)
    y1, y0, m0 = ops.conv3d_pair01_train(xd, w0d, b0d, w1d, b1d, mid_layout=mid_l, out_layout=out_l)
    assert torch.equal(y1, want1)
    ref0 = F.leaky_relu(F.conv3d(x.double(), w0.double(), b0.double(), stride=1, padding=1), 0.2)
    got0 = _to_ncdhw(y0, mid_l).cpu().double()
    assert torch.isfinite(got0).all()
    assert float((got0 - ref0).abs().max()) <= 2e-6 * float(ref0.abs().max())
    # the mask: bit r of byte q of a voxel = "channel 4q + r > 0" of the STORED activation
    y0p = ops.hps_to_ndhwc(y0) if mid_l == ops.LAYOUT_NDHWC_HPS else y0          # (B,D,W,H,16), plain rows
    bits = (y0p > 0).view(B, D, W, H, 4, 4).to(torch.uint8)
    want_m = bits[..., 0] | (bits[..., 1] << 1) | (bits[..., 2] << 2) | (bits[..., 3] << 3)
    assert torch.equal(m0, want_m)
    # and against the two-kernel training forward's side outputs (same layout, fp32-close values)
    if H % 4 == 0 and W * H >= 256 and Cin <= 3:      # (the generic first-block kernel writes a mask for <= 3 channels only)
        m_nat = torch.empty_like(m0)
        y_nat = ops.conv3d_k3_lrelu(xd, w0d, b0d, 1, in_layout=ops.LAYOUT_NCDHW, out_layout=mid_l, mask_out=m_nat)
        torch.testing.assert_close(y0, y_nat, rtol=0, atol=4e-6 * float(ref0.abs().max()))
        assert (m0 != m_nat).float().mean() < 1e-4        # sign flips only where the activation is ~0


def test_pair_kernel_strided_batches_and_slopes():
    """in0 as a z-slab view of a larger volume (batch stride), the output into a strided batch, other slopes, no bias."""
    from liftreg_amd import ops
    B, Cin, D, W, H = 2, 3, 6, 32, 40
    x, w0, b0, w1, b1 = _make(B, Cin, D + 4, W, H, 77)
    xd, w0d, w1d = x.to(DEV), w0.to(DEV), w1.to(DEV)
    big0 = xd[:, 0:1].contiguous()
    x0 = big0[:, :, 2:2 + D]                                  # view: dense per batch element, larger batch stride
    rest = xd[:, 1:, 2:2 + D].contiguous()
    Do, Wo, Ho = (D - 1) // 2 + 1, (W - 1) // 2 + 1, (H - 1) // 2 + 1
    buf = torch.full((B, Do + 3, Wo, Ho, 32), 7.0, device=DEV)
    out = buf[:, 1:1 + Do]
    y = ops.conv3d_pair01(x0, rest, w0d, None, w1d, None, out_layout=ops.LAYOUT_NDHWC, slope0=0.0, slope1=1.0, out=out)
    torch.cuda.synchronize()
    assert y.data_ptr() == out.data_ptr()
    assert float(buf[:, 0].min()) == 7.0 and float(buf[:, 1 + Do:].min()) == 7.0 and float(buf[:, 1 + Do:].max()) == 7.0
    xs = torch.cat([x0, rest], 1).cpu()
    z = torch.zeros(16), torch.zeros(32)
    ref = _ref64(xs, w0, z[0], w1, z[1], 0.0, 1.0)
    got = y.permute(0, 4, 1, 2, 3).cpu().double()
    assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_pair_kernel_is_deterministic_and_batch_independent():
    from liftreg_amd import ops
    B, Cin, D, W, H = 3, 3, 8, 48, 48
    x, w0, b0, w1, b1 = _make(B, Cin, D, W, H, 5)
    xd, w0d, b0d, w1d, b1d = (t.to(DEV) for t in (x, w0, b0, w1, b1))
    x0, rest = xd[:, 0:1].contiguous(), xd[:, 1:].contiguous()
    a = ops.conv3d_pair01(x0, rest, w0d, b0d, w1d, b1d)
    b = ops.conv3d_pair01(x0, rest, w0d, b0d, w1d, b1d)
    c = ops.conv3d_pair01(x0[1:2].contiguous(), rest[1:2].contiguous(), w0d, b0d, w1d, b1d)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert torch.equal(a[1:2], c)


@pytest.mark.parametrize("world,Cin", [(2, 3), (4, 3), (8, 3), (2, 5), (4, 5)])
def test_pair_kernel_on_z_slabs_gives_the_bits_of_the_whole_volume(world, Cin):
    """The z-slab form (parallel.SlabShardedRegistration): each rank computes output planes [d0/2, d1/2) from input planes
    d0-2 .. d1 — a view of the replicated moving volume and its own planes of the feature volume; no halo exchange."""
    from liftreg_amd import ops
    B, D, W, H = 2, 32, 40, 32
    x, w0, b0, w1, b1 = _make(B, Cin, D, W, H, 11)
    xd, w0d, b0d, w1d, b1d = (t.to(DEV) for t in (x, w0, b0, w1, b1))
    x0, rest = xd[:, 0:1].contiguous(), xd[:, 1:].contiguous()
    full = ops.conv3d_pair01(x0, rest, w0d, b0d, w1d, b1d, out_layout=ops.LAYOUT_NDHWC_HPS)
    for r in range(world):
        d0, d1 = r * D // world, (r + 1) * D // world
        lo, hi = max(d0 - 2, 0), min(d1 + 1, D)
        rows = (d1 - d0) // 2
        buf = torch.full((B, rows + 3, (W - 1) // 2 + 1, (H - 1) // 2 + 1, 32), 5.0, device=DEV)
        y = ops.conv3d_pair01(x0[:, :, lo:hi], rest[:, :, lo:hi].contiguous(), w0d, b0d, w1d, b1d, out_layout=ops.LAYOUT_NDHWC_HPS,
                              out=buf[:, 2:2 + rows], slab=(D, lo, d0 // 2, rows))
        torch.cuda.synchronize()
        assert torch.equal(y, full[:, d0 // 2:d1 // 2]), f"rank {r} of {world}"
        assert float(buf[:, :2].min()) == 5.0 and float(buf[:, 2 + rows:].max()) == 5.0
    with pytest.raises(Exception):      # input planes the slab needs but the buffers do not hold
        ops.conv3d_pair01(x0[:, :, 4:12], rest[:, :, 4:12].contiguous(), w0d, b0d, w1d, b1d, slab=(D, 4, 2, 4))
