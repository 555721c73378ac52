"""CPU: the C-ABI library loads, exports every symbol include/liftreg_hip.h declares, and its
argument-validation paths answer without touching a GPU; host-side logic (plugin loader, identity
tables, poses, model construction/state-dict) matches the reference's golden vectors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    from liftreg_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        _hip.build_library()
    return _hip


def test_library_exports_every_declared_symbol(hip):
    header = open(os.path.join(ROOT, "include", "liftreg_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(lr_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 16
    raw = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
    assert sorted(hip.SIGNATURES) == declared                      # binding covers exactly the header
    lib = hip.lib()
    assert lib.lr_abi_version() == 1 and lib.lr_target_arch() == b"gfx950"


def test_argument_validation_without_gpu(hip):
    lib = hip.lib()
    assert lib.lr_backproject_f32(None, None, None, 1, 1, 4, 4, 4, 4, 4, 0, 4, 64, None) == -2
    assert lib.lr_conv3d_packed_floats(3, 16, 0) == 3 * 7 * 64
    assert lib.lr_conv3d_packed_floats(16, 32, 1) == 27 * 1 * 2 * 64 * 4
    assert lib.lr_conv3d_packed_floats(3, 8, 0) == -3
    assert lib.lr_strerror(0) == b"ok" and lib.lr_strerror(-5).startswith(b"pointer")
    one = ctypes.c_float(0)
    p = ctypes.addressof(one)
    assert lib.lr_pca_reconstruct_f32(p, p, p, p, 64, 4, 8, 8, 8, None) == -1          # B > 32
    assert lib.lr_drr_forward_f32(p, p, p, p, 4, 4, 4, 2, 1, 1, 4, 4, 0, 0, None) == -1  # d1 <= d0
    assert lib.lr_warp_trilinear_f32(p, None, p, p, None, None, None, p, 1, 1, 4, 4, 4, 0, 4, 0, None) == -2


def test_ops_refuse_cpu_tensors(hip):
    from liftreg_amd import ops
    with pytest.raises(hip.LiftRegHipError):
        ops.warp(torch.zeros(1, 1, 4, 4, 4), torch.zeros(1, 3, 4, 4, 4))
    with pytest.raises(hip.LiftRegHipError):
        ops.ncc_loss(torch.zeros(2, 8), torch.zeros(2, 8))


def test_get_class_and_plugin_paths():
    from liftreg_amd.utils.general import get_class
    for path in ("liftreg_amd.models.LiftRegDeformSubspaceBackproj.model", "liftreg_amd.layers.losses.NCCLoss",
                 "liftreg_amd.layers.layers.NCCLoss", "liftreg_amd.utils.net_utils.Bilinear"):
        assert callable(get_class(path))
    with pytest.raises(ValueError):
        get_class("nodots")


def test_identity_tables_and_poses_match_reference(golden):
    from liftreg_amd.utils.net_utils import identity_axis_tables
    from liftreg_amd.utils.sdct_projection_utils import scan_poses, calc_relative_atten_coef
    g = golden("warp_a")
    shape = g["img"].shape[2:]
    t0, t1, t2 = identity_axis_tables(shape)
    idm = np.stack(np.broadcast_arrays(t0[:, None, None], t1[None, :, None], t2[None, None, :]))
    assert np.array_equal(idm, g["identity"])
    d = golden("drr_default_receptor")
    assert np.array_equal(scan_poses(30, 4, d["hu"].shape[1]), d["poses"])
    a = golden("drr_a")
    assert np.array_equal(calc_relative_atten_coef(a["hu"]), a["mu"])


def test_model_state_dict_matches_reference_keys(golden, tmp_path):
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    g = golden("model_32")
    L = int(g["latent_dim"])
    np.save(tmp_path / "pca_vectors.npy", np.zeros((L, 3 * 32 ** 3), np.float32))
    np.save(tmp_path / "pca_mean.npy", np.zeros((3 * 32 ** 3,), np.float32))
    net = model([32, 32, 32], {"drr_feature_num": 2, "latent_dim": L, "pca_path": str(tmp_path)})
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd::")}
    assert sorted(net.state_dict()) == sorted(sd) and len(sd) == 19
    net.load_state_dict(sd, strict=True)
    assert np.array_equal(net.gaussian_smooth.weight.numpy(), g["sd::gaussian_smooth.weight"])
    assert net.pca_vectors.shape == (3 * 32 ** 3, L)                  # the reference's .T view
    assert tuple(net.id_transform.shape) == (3, 32, 32, 32)
    assert net.get_extra_to_plot() == (None, None) and net.get_disp() == (None, "")
    # reference's native size: flatten width 32*5^3 = 4000 (…Backproj.py:36)
    big = model([160, 160, 160], {"drr_feature_num": 4, "latent_dim": 56, "pca_path": "synthetic"})
    assert big.encoders[6][1].fc.in_features == 4000 and big.encoders[0].conv.in_channels == 5
    with pytest.raises(Exception):
        with torch.no_grad():                                         # CPU tensors: no fallback
            net({"source": torch.zeros(1, 1, 32, 32, 32), "target": torch.zeros(1, 1, 32, 32, 32),
                 "target_proj": torch.zeros(1, 2, 32, 32), "target_poses": torch.zeros(1, 2, 3)})
