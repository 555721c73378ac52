"""GPU: the small rows of SURVEY §8 closed in round 2, each against golden vectors produced by importing the reference
(tests/golden/make_golden_r2.py): forward_grids / forward_grids_with_poses (a5), the pose-less float64 backproj_grids
(a6'), the CSV-geometry projector wrapper (a2') and the DirLab landmark sampler / TRE (f4)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def test_forward_grids_bit_exact(golden, dev):
    """sdct_projection_utils.py:204-225 (emitter at 3.0·W, default 1.5x receptor) and :252-265 (explicit poses)."""
    from liftreg_amd.utils import sdct_projection_utils as S
    g = golden("fwd_grids")
    shape, spacing = tuple(int(v) for v in g["shape"]), tuple(float(v) for v in g["spacing"])
    g1, dx1 = S.forward_grids(30, 3, spacing, shape, device=dev, receptor_size=(7, 9))
    assert np.array_equal(g1.cpu().numpy(), g["g1"]) and np.array_equal(dx1.cpu().numpy(), g["dx1"])
    g2, dx2 = S.forward_grids(24, 2, spacing, shape, device=dev)
    assert tuple(g2.shape) == g["g2"].shape
    assert np.array_equal(g2.cpu().numpy(), g["g2"]) and np.array_equal(dx2.cpu().numpy(), g["dx2"])
    g3, dx3 = S.forward_grids_with_poses(g["poses"], spacing, shape, device=dev, receptor_size=(6, 11))
    assert np.array_equal(g3.cpu().numpy(), g["g3"]) and np.array_equal(dx3.cpu().numpy(), g["dx3"])
    with pytest.raises(RuntimeError):
        S.forward_grids(30, 3, spacing, shape, device=torch.device("cpu"))       # no CPU path


def test_backproj_grids_poseless_float64_bit_exact(golden, dev):
    """sdct_projection_utils.py:179-202: float64 (the reference's dtype), scale·g + trans op order."""
    from liftreg_amd.utils import sdct_projection_utils as S
    g = golden("bp_poseless")
    for tag in ("a", "b"):
        shp, pshape = tuple(int(v) for v in g[f"{tag}_shape"]), tuple(int(v) for v in g[f"{tag}_pshape"])
        got = S.backproj_grids(int(g[f"{tag}_range"]), int(g[f"{tag}_P"]), shp, pshape, device=dev)
        assert got.dtype == torch.float64 and tuple(got.shape) == g[f"{tag}_grid"].shape
        assert np.array_equal(got.cpu().numpy(), g[f"{tag}_grid"]), tag


def test_csv_geometry_wrapper(golden, dev, tmp_path):
    """calculate_projection_wraper_with_geo_csv_file (sdct_projection_utils.py:161-177): poses = csv[1:]/spacing,
    explicit and default (1.5x) receptor."""
    from liftreg_amd.utils import sdct_projection_utils as S
    g = golden("csv_geo")
    path = os.path.join(tmp_path, "geo.csv")
    with open(path, "w") as fh:
        fh.write("x,y,z\n" + "\n".join(",".join(repr(float(v)) for v in row) for row in g["geo_mm"]) + "\n")
    sp = tuple(float(v) for v in g["img_spacing"])
    proj, poses = S.calculate_projection_wraper_with_geo_csv_file(g["mu"], sp, path, receptor_size=(9, 13))
    assert poses.dtype == np.float64 and np.array_equal(poses, g["poses"])
    assert proj.dtype == np.float32 and proj.shape == g["proj"].shape
    np.testing.assert_allclose(proj, g["proj"], rtol=1e-5, atol=1e-6)
    proj_def, _ = S.calculate_projection_wraper_with_geo_csv_file(g["mu"], sp, path)
    assert proj_def.shape == g["proj_default"].shape
    np.testing.assert_allclose(proj_def, g["proj_default"], rtol=1e-5, atol=1e-6)
    # HU → μ folded into the projector gives the same DRR as the host conversion
    proj_hu, _ = S.calculate_projection_wraper_with_geo_csv_file(g["hu"], sp, path, receptor_size=(9, 13), hu_input=True)
    assert np.array_equal(proj_hu, proj)


def test_landmark_sampler_and_tre(golden, dev, tmp_path):
    """tools/evaluate_dir_lab.py:46-59 (calc_warped_points, float64 grid_sample) and :81-138 (eval_with_data)."""
    from liftreg_amd import ops
    from liftreg_amd.tools import evaluate_dir_lab as E
    g = golden("tre")
    warped = E.calc_warped_points(torch.from_numpy(g["source_norm"]), torch.from_numpy(g["phi"][None]).double(), g["dim"],
                                  g["spacing"], g["phi_spacing"])
    assert warped.dtype == torch.float64 and tuple(warped.shape) == g["warped"].shape
    np.testing.assert_allclose(warped.numpy(), g["warped"], rtol=0, atol=1e-12)
    tre, xyz = E.eval_with_data(g["source"], g["target"], g["phi"][None], g["dim"], g["spacing"], g["origin"], g["phi_spacing"])
    assert abs(tre - float(g["tre"])) < 1e-12 and np.allclose(xyz, g["tre_xyz"], rtol=0, atol=1e-12)
    # file variant: landmark text files + the .npy map save_deformations writes
    for name, pts in (("s.txt", g["source"]), ("t.txt", g["target"])):
        with open(os.path.join(tmp_path, name), "w") as fh:
            fh.write("".join(f"{float(a)!r}\t{float(b)!r}\t{float(c)!r}\n" for a, b, c in pts))
    np.save(os.path.join(tmp_path, "phi.npy"), g["phi"])
    assert np.array_equal(E.readPoint(os.path.join(tmp_path, "s.txt"))[:5], g["pts_read"])
    tre_f, xyz_f = E.eval_with_file(os.path.join(tmp_path, "s.txt"), os.path.join(tmp_path, "t.txt"),
                                    os.path.join(tmp_path, "phi.npy"), g["dim"], g["spacing"], g["origin"], g["phi_spacing"])
    assert abs(tre_f - float(g["tre"])) < 1e-12
    # sampler properties: an exact grid node returns the node's value; a point outside by more than a cell returns 0
    vol = torch.arange(2 * 3 * 4 * 5, dtype=torch.float64, device=dev).reshape(2, 3, 4, 5)
    pts = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], [3.5, 0.0, 0.0], [float("nan"), 0.0, 0.0]], dtype=torch.float64,
                       device=dev)
    out = ops.sample_points_f64(vol, pts).cpu().numpy()
    assert np.array_equal(out[0], [0.0, 60.0]) and np.array_equal(out[1], [59.0, 119.0])
    assert np.array_equal(out[2], [0.0, 0.0]) and np.array_equal(out[3], [0.0, 0.0])
    with pytest.raises(Exception):
        ops.sample_points_f64(vol.cpu(), pts)                                           # no CPU path
