#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by IMPORTING the reference
(uncbiag/LiftReg, read-only at /root/reference) in the build container.

Run:  python tests/golden/make_golden.py
Only the build container has /root/reference; the .npz files this writes are
committed and are what travels.  Harness-level shims only (the reference is never
edited): removed numpy aliases, `.cuda()` as identity, a stub `mermaid` module
(imported but unused by the configured NCCLoss), a proxy so that the wrapper's
hard-coded torch.device("cuda") resolves to the CPU.

Every array saved is either a seeded input or an output of a reference function.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    np.float = float  # removed aliases used at sdct_projection_utils.py:141,182,207
    np.int = int
    torch.Tensor.cuda = lambda self, *a, **k: self  # hard .cuda() at …Backproj.py:42-43, net_utils.py:87
    for name in ("mermaid", "mermaid.finite_differences"):
        sys.modules.setdefault(name, types.ModuleType(name))  # layers/losses.py:6 (unused by NCC)
    sys.path.insert(0, REF)
    import liftreg.utils.sdct_projection_utils as S
    import liftreg.utils.net_utils as N
    import liftreg.layers.layers as L
    import liftreg.layers.losses as LL
    import liftreg.models.LiftRegDeformSubspaceBackproj as M
    return S, N, L, LL, M


class _CpuTorch:
    """Stands in for the `torch` name inside sdct_projection_utils so that
    torch.device("cuda") (sdct_projection_utils.py:154) yields the CPU."""

    def __getattr__(self, k):
        if k == "device":
            return lambda *_a, **_k: torch.device("cpu")
        return getattr(torch, k)


def phantom(rs, shape):
    """Small HU-like volume: smooth blobs + noise, clipped to [-1024, 1000]."""
    D, W, H = shape
    z, y, x = np.mgrid[0:D, 0:W, 0:H].astype(np.float32)
    vol = np.full(shape, -1000.0, np.float32)
    for _ in range(4):
        c = rs.uniform(0.25, 0.75, 3) * np.array(shape)
        r = rs.uniform(0.15, 0.35, 3) * np.array(shape)
        hu = rs.choice([-850.0, 40.0, 400.0])
        m = ((z - c[0]) / r[0]) ** 2 + ((y - c[1]) / r[1]) ** 2 + ((x - c[2]) / r[2]) ** 2 < 1
        vol[m] = hu
    vol += rs.normal(0, 20, shape).astype(np.float32)
    return np.clip(vol, -1024, 1000).astype(np.float32)


def main():
    torch.manual_seed(2021)
    torch.set_num_threads(4)
    S, N, L, LL, M = import_reference()
    rs = np.random.RandomState(2021)
    cpu = torch.device("cpu")

    # ---- a1/a2/a3/a4: forward projector ------------------------------------------------
    for tag, shape, res, P in (("drr_a", (12, 10, 14), (9, 11), 3), ("drr_b", (16, 16, 16), (16, 16), 2),
                               ("drr_c", (10, 12, 8), (15, 12), 4)):
        hu = phantom(rs, shape)
        mu = S.calc_relative_atten_coef(hu)
        S_torch = S.torch
        S.torch = _CpuTorch()
        try:
            proj_w, poses = S.calculate_projection_wraper(mu, 30, P, (2.2, 2.2, 2.2), receptor_size=res)
        finally:
            S.torch = S_torch
        spacing = torch.tensor((2.2, 2.2, 2.2))
        grid, dx = S.project_grid_multi(poses, res, [1, 1, 1], torch.Size(shape), spacing, cpu, torch.float32)
        proj = S.calculate_projection(mu, poses, res, [1, 1, 1], (2.2, 2.2, 2.2), cpu)
        assert np.array_equal(proj, proj_w)
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), hu=hu, mu=mu, poses=poses,
                            spacing=np.array((2.2, 2.2, 2.2), np.float32), resolution=np.array(res),
                            grid=grid.numpy(), dx=dx.numpy(), proj=proj)
    # default receptor (1.5x) pose/resolution logic of the wrapper
    hu = phantom(rs, (8, 8, 8))
    S_torch = S.torch
    S.torch = _CpuTorch()
    try:
        proj_def, poses_def = S.calculate_projection_wraper(S.calc_relative_atten_coef(hu), 30, 4, (2.2, 2.2, 2.2))
    finally:
        S.torch = S_torch
    np.savez_compressed(os.path.join(OUT, "drr_default_receptor.npz"), hu=hu, poses=poses_def, proj=proj_def)

    # ---- a6/a7: backprojection ------------------------------------------------------------
    for tag, shape, pshape, P, B in (("bp_a", (12, 10, 14), (9, 11), 3, 2), ("bp_b", (16, 16, 16), (16, 16), 2, 1),
                                     ("bp_c", (8, 12, 10), (20, 6), 4, 2)):
        poses64 = (np.stack([np.tan(np.linspace(-15, 15, P) / 180. * np.pi) * 3., np.full(P, 3.5),
                             np.linspace(-0.2, 0.2, P)], 1) * shape[1])
        poses = np.broadcast_to(poses64.astype(np.float32), (B, P, 3)).copy()
        proj = rs.uniform(-1, 1, (B, P) + pshape).astype(np.float32)
        g = S.backproj_grids_with_poses(poses[0:1], shape, pshape, device=cpu)
        tp = torch.from_numpy(proj)
        D, W, H = shape
        gp = g.permute(0, 1, 3, 4, 5, 2)
        tv = torch.nn.functional.grid_sample(  # the call at …Backproj.py:89-93, verbatim arguments
            tp.reshape(B * P, 1, *pshape), gp.expand(B, -1, -1, -1, -1, -1).reshape(B * P, D * W, H, -1),
            align_corners=True, padding_mode="zeros").reshape(B, P, D, W, H)
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), poses=poses, proj=proj, grid=g.numpy(),
                            volume=tv.numpy(), shape=np.array(shape))

    # ---- a11/a12: identity map + Bilinear ----------------------------------------------------
    for tag, shape, B in (("warp_a", (12, 10, 14), 2), ("warp_b", (16, 16, 16), 1)):
        idm = N.gen_identity_map(list(shape), 1.0)
        img = rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32)
        disp = (rs.normal(0, 0.15, (B, 3) + shape)).astype(np.float32)
        disp[:, :, :2] += 0.8  # push some samples out of the volume (padding paths)
        phi = torch.from_numpy(disp) + idm
        timg = torch.from_numpy(img)
        out = {"identity": idm.numpy(), "img": img, "disp": disp, "phi": phi.numpy()}
        out["warped_zeros_scale"] = N.Bilinear(zero_boundary=True, using_scale=True)(timg, phi).numpy()
        out["warped_border_scale"] = N.Bilinear(zero_boundary=False, using_scale=True)(timg, phi).numpy()
        out["warped_zeros_noscale"] = N.Bilinear(zero_boundary=True, using_scale=False)(timg, phi).numpy()
        out["warped_nearest"] = N.Bilinear(zero_boundary=True, using_scale=True, mode="nearest")(timg, phi).numpy()
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)

    # ---- a13: NCC ----------------------------------------------------------------------------
    x = rs.uniform(-1, 1, (3, 2, 6, 7, 8)).astype(np.float32)
    y = (0.6 * x + 0.4 * rs.uniform(-1, 1, x.shape)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ncc.npz"), x=x, y=y,
                        loss_configured=LL.NCCLoss()(torch.from_numpy(x), torch.from_numpy(y)).numpy(),
                        loss_squared=L.NCCLoss()(torch.from_numpy(x), torch.from_numpy(y)).numpy())

    # ---- a8/a9: convBlock / FullyConnectBlock ----------------------------------------------
    for tag, cin, cout, stride, shape, B in (("conv_a", 3, 16, 1, (6, 7, 9), 2), ("conv_b", 16, 32, 2, (8, 6, 10), 1),
                                             ("conv_c", 32, 32, 2, (5, 8, 7), 2)):
        blk = L.convBlock(cin, cout, stride=stride, bias=True)
        xin = rs.uniform(-1, 1, (B, cin) + shape).astype(np.float32)
        with torch.no_grad():
            yout = blk(torch.from_numpy(xin)).numpy()
        np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), x=xin, weight=blk.conv.weight.detach().numpy(),
                            bias=blk.conv.bias.detach().numpy(), stride=np.array(stride), y=yout)
    fc = L.FullyConnectBlock(96, 40)
    fc_lin = L.FullyConnectBlock(40, 7, nonlinear=None)
    xin = rs.uniform(-1, 1, (3, 96)).astype(np.float32)
    with torch.no_grad():
        h1 = fc(torch.from_numpy(xin))
        h2 = fc_lin(h1)
    np.savez_compressed(os.path.join(OUT, "fc.npz"), x=xin, w1=fc.fc.weight.detach().numpy(),
                        b1=fc.fc.bias.detach().numpy(), w2=fc_lin.fc.weight.detach().numpy(),
                        b2=fc_lin.fc.bias.detach().numpy(), h1=h1.numpy(), h2=h2.numpy())

    # ---- a14: whole model.forward at 32^3 (FC1 width patched: harness-level substitution) ----
    import tempfile
    n, P, Lat, B = 32, 2, 6, 2
    shape = (n, n, n)
    with tempfile.TemporaryDirectory() as td:
        # regenerated from the seed by the tests (keeps the fixture small): see pca_basis_32()
        rs_pca = np.random.RandomState(7)
        pca_vectors = (rs_pca.normal(0, 0.02 / np.sqrt(Lat), (Lat, 3 * n ** 3))).astype(np.float32)
        pca_mean = (rs_pca.normal(0, 0.002, (3 * n ** 3,))).astype(np.float32)
        np.save(os.path.join(td, "pca_vectors.npy"), pca_vectors)
        np.save(os.path.join(td, "pca_mean.npy"), pca_mean)
        opt = {"drr_feature_num": P, "latent_dim": Lat, "pca_path": td}
        net = M.model(list(shape), opt)
    net.encoders[6][1] = L.FullyConnectBlock(32 * (n // 32) ** 3, 800)  # reference hard-codes 32*5^3 (:36)
    net.eval()
    hu_m, hu_t = phantom(rs, shape), phantom(rs, shape)
    norm = lambda v: ((np.clip(v, -1000, 0) + 1000) / 1000 * 2 - 1).astype(np.float32)  # Registration2D3DDataset.py:186-209
    moving, target = norm(hu_m), norm(hu_t)
    poses64 = S_poses = (np.stack([np.tan(np.linspace(-15, 15, P) / 180. * np.pi) * 3., np.full(P, 3.5),
                                   np.linspace(-0.2, 0.2, P)], 1) * n)
    tproj = S.calculate_projection(S.calc_relative_atten_coef(np.flip(hu_t, axis=1).copy()), poses64, (n, n),
                                   [1, 1, 1], (2.2, 2.2, 2.2), cpu)
    tproj = (np.clip(tproj, 0, 6) / 6 * 2 - 1).astype(np.float32)
    seg_m = (rs.uniform(0, 1, shape) > 0.2).astype(np.float32)
    seg_t = (rs.uniform(0, 1, shape) > 0.2).astype(np.float32)
    inp = {"source": torch.from_numpy(np.stack([moving, moving[::-1].copy()])[:, None]),
           "target": torch.from_numpy(np.stack([target, target[:, ::-1].copy()])[:, None]),
           "target_proj": torch.from_numpy(np.stack([tproj, tproj[:, ::-1].copy()])),
           "target_poses": torch.from_numpy(np.broadcast_to(poses64.astype(np.float32), (B, P, 3)).copy()),
           "source_label": torch.from_numpy(np.stack([seg_m, seg_m])[:, None]),
           "target_label": torch.from_numpy(np.stack([seg_t, seg_t])[:, None])}
    with torch.no_grad():
        out = net(inp)
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(
        os.path.join(OUT, "model_32.npz"), pca_seed=np.array(7), latent_dim=np.array(Lat),
        state_keys=np.array(sorted(sd.keys())), **{"sd::" + k: v for k, v in sd.items()},
        **{"in::" + k: v.numpy() for k, v in inp.items()},
        **{"out::" + k: out[k].numpy() for k in ("warped", "phi", "params", "target", "pca_coefs")})
    print("golden vectors written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f"  {f:28s} {os.path.getsize(os.path.join(OUT, f)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
