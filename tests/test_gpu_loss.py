"""GPU: the SubspaceLoss plugin (a16) — NCC similarity + displacement regulariser — against the torch-CPU
restatement.  The regulariser's stencil is mermaid's (absent, parity UNPINNED): both sides implement the
documented assumption, so this test pins the kernel to the restatement, not to the reference."""
import numpy as np
import pytest
import torch

from oracle import ref_ops as ro

pytestmark = pytest.mark.gpu


def test_subspace_loss_plugin_vs_restatement():
    from liftreg_amd import ops
    from liftreg_amd.utils.general import get_class
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(17)
    cls = get_class("liftreg_amd.losses.SubspaceLoss.loss")
    opt = {"sim_class": "liftreg_amd.layers.losses.NCCLoss", "initial_reg_factor": 0.01, "min_reg_factor": 0.01,
           "reg_factor_decay_from": 2}                                   # cur_task_setting.json:47-52
    f = cls(opt)
    for shape, B in (((12, 10, 14), 2), ((9, 16, 33), 1), ((2, 2, 2), 3), ((7, 5, 16), 2), ((2, 3, 8), 1), ((4, 2, 12), 1)):
        warped = rs.uniform(-1, 1, (B, 1) + shape).astype(np.float32)
        target = (0.7 * warped + 0.3 * rs.uniform(-1, 1, warped.shape)).astype(np.float32)
        disp = rs.normal(0, 0.05, (B, 3) + shape).astype(np.float32)
        out = {"warped": torch.from_numpy(warped).to(dev), "target": torch.from_numpy(target).to(dev),
               "params": torch.from_numpy(disp).to(dev), "pca_coefs": None, "epoch": 5}
        with torch.no_grad():
            got = f(out)
        want = ro.subspace_loss({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in out.items()}, 5)
        assert set(got) == {"total_loss", "sim_loss", "reg_loss"}
        assert isinstance(got["sim_loss"], float) and isinstance(got["reg_loss"], float)
        assert abs(got["sim_loss"] - want["sim_loss"]) < 2e-6
        assert abs(got["reg_loss"] - want["reg_loss"]) < 1e-5 * max(1.0, abs(want["reg_loss"]))
        assert abs(float(got["total_loss"]) - float(want["total_loss"])) < 1e-5
    assert f.get_reg_factor(0) == 0.01 and f.get_reg_factor(50) == 0.01   # shipped config: constant factor
    big = cls({"initial_reg_factor": 10, "min_reg_factor": 1e-3, "reg_factor_decay_from": 10})
    assert big.get_reg_factor(3) == 10.0 and abs(big.get_reg_factor(12) - 10 * ro.sigmoid_decay(12, 10, 2)) < 1e-9
    # properties at full size: constant field → 0; linear ramp → its squared slope (central == one-sided)
    n = 256
    const = torch.full((1, 3, n, n, n), 0.3, device=dev)
    assert float(ops.disp_reg(const)) == 0.0
    ramp = torch.zeros((1, 3, n, n, n), device=dev)
    ramp[:, 0] = torch.linspace(-1, 1, n, device=dev)[:, None, None] * 0.05    # 0.05·x along D in normalised units
    assert abs(float(ops.disp_reg(ramp)) - 0.05 ** 2) < 1e-7


def test_disp_reg_marching_kernel_equals_vector_kernel(monkeypatch):
    """The z-marching regulariser (rows with H >= 64: a block walks plane chunks, in-plane neighbours from an LDS tile) against the
    vectorised kernel it replaces (LIFTREG_REG_NOMARCH=1) and the torch oracle: ragged D / W (rows and planes that do not fill
    the last block / chunk), several H, few and many partials per batch element."""
    from liftreg_amd import ops
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(31)
    # H = 160 and 68: R x H/4 = 480 / 272 threads — not whole wavefronts (the block is padded with inactive threads)
    for shape, B, nblk in (((5, 9, 64), 2, None), ((37, 10, 128), 1, 7), ((33, 7, 96), 2, 64), ((20, 33, 256), 1, None), ((3, 2, 64), 1, 3),
                           ((9, 13, 160), 1, None), ((7, 35, 68), 2, None), ((4, 5, 68), 1, 9)):
        disp = rs.normal(0, 0.05, (B, 3) + shape).astype(np.float32)
        d = torch.from_numpy(disp).to(dev)
        got = float(ops.disp_reg(d, nblk=nblk))
        monkeypatch.setenv("LIFTREG_REG_NOMARCH", "1")
        old = float(ops.disp_reg(d, nblk=nblk))
        monkeypatch.delenv("LIFTREG_REG_NOMARCH")
        want = float(ro.disp_reg(torch.from_numpy(disp)))
        assert abs(got - old) <= 2e-6 * abs(old), (shape, got, old)
        assert abs(got - want) <= 1e-5 * abs(want), (shape, got, want)


def test_disp_reg_gradient_marching_kernel_equals_vector_kernel(monkeypatch):
    """The z-marching gradient of the regulariser (H >= 64) against the vectorised kernel (LIFTREG_REG_NOMARCH=1) and torch
    autograd of the oracle: ragged D / W / plane chunks, faces in every axis (the +-1 / +-2 stencil changes there)."""
    from liftreg_amd import ops_bwd
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(37)
    for shape, B in (((5, 9, 64), 2), ((37, 10, 128), 1), ((33, 7, 96), 1), ((70, 6, 64), 1), ((3, 4, 64), 1), ((9, 13, 160), 1), ((7, 35, 68), 1)):
        disp = rs.normal(0, 0.05, (B, 3) + shape).astype(np.float32)
        d = torch.from_numpy(disp).to(dev)
        gout = torch.tensor(0.7, device=dev)
        got = ops_bwd.disp_reg_bwd(d, gout)
        monkeypatch.setenv("LIFTREG_REG_NOMARCH", "1")
        old = ops_bwd.disp_reg_bwd(d, gout)
        monkeypatch.delenv("LIFTREG_REG_NOMARCH")
        scale = float(old.abs().max())
        assert float((got - old).abs().max()) <= 2e-6 * scale, shape
        dt = torch.from_numpy(disp).requires_grad_(True)
        (ro.disp_reg(dt) * 0.7).backward()
        assert float((got.cpu() - dt.grad).abs().max()) <= 2e-5 * float(dt.grad.abs().max()), shape
