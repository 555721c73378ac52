"""Cone-beam projection utilities with the reference's signatures
(src/liftreg/utils/sdct_projection_utils.py), backed by the HIP projector.

The reference builds a (P,Rd,Rh,W,3) sampling grid and calls F.grid_sample; here the
projector derives the grid in registers.  `device` arguments are honoured when they
name a GPU; there is no CPU path.
"""
import numpy as np
import torch
from numpy import genfromtxt

from .. import ops


def calc_relative_atten_coef(img):
    """HU → linear attenuation (water = 0.2/cm), sdct_projection_utils.py:6-9 (host numpy).

    The projector can also do this on load: calculate_projection(..., hu_input=True).
    """
    new_img = np.asarray(img).astype(np.float32).copy()
    new_img[new_img < -1000] = -1000
    return (new_img + 1000.) / 1000. * 0.2


def _gpu(device):
    device = torch.device("cuda") if device is None else torch.device(device)
    if device.type != "cuda":
        raise RuntimeError(f"liftreg_amd projects on the GPU only, got device={device}")
    return device


def calculate_projection(img, poses, resolution, sample_rate, spacing, device=None, *, hu_input=False,
                         flip_w=False, as_tensor=False):
    """DRR of `img` (D,W,H) for emitter `poses` (P,3) → (P,Rd,Rh) float32 numpy array
    (sdct_projection_utils.py:59-100).  `img` may be a numpy array (copied to the GPU, as the
    reference does at :71) or a GPU tensor (no copy).  Extra keyword-only options fold
    calc_relative_atten_coef / the axis-1 flip into the volume load and skip the D2H copy."""
    if list(sample_rate) != [1, 1, 1]:
        raise NotImplementedError("the reference only ever uses sample_rate [1,1,1] "
                                  "(sdct_projection_utils.py:152,174,221,254)")
    device = _gpu(device)
    vol = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))
    vol = vol.to(device=device, dtype=torch.float32)
    poses32 = torch.from_numpy(np.asarray(poses)).type(torch.float32).numpy()  # .type(dtype) at :28
    proj = ops.drr_forward(vol, poses32, resolution, spacing, hu_input=hu_input, flip_w=flip_w)
    return proj if as_tensor else proj.cpu().numpy()


def scan_poses(scan_range, proj_num, W, y_scale=3.5):
    """Emitter poses of the scan-range geometry (sdct_projection_utils.py:139-145,155), float64 (P,3)."""
    angle_half = scan_range / 2.
    poses_scale = np.ndarray((proj_num, 3), dtype=float)
    poses_scale[:, 1] = y_scale
    poses_scale[:, 0] = np.tan(np.linspace(-angle_half, angle_half, num=proj_num) / 180. * np.pi) * 3.
    poses_scale[:, 2] = np.linspace(-0.2, 0.2, num=proj_num)
    return poses_scale * W


def _resolution(img_shape, receptor_size):
    if receptor_size is not None:
        return list(receptor_size)
    resolution_scale = 1.5
    return [int(img_shape[0] * resolution_scale), int(img_shape[2] * resolution_scale)]


def calculate_projection_wraper(img_3d, scan_range, proj_num, spacing, receptor_size=None, **kw):
    """sdct_projection_utils.py:138-159 → (proj (P,Rd,Rh), poses (P,3) float64)."""
    poses = scan_poses(scan_range, proj_num, img_3d.shape[1])
    resolution = _resolution(img_3d.shape, receptor_size)
    return calculate_projection(img_3d, poses, resolution, [1, 1, 1], spacing, torch.device("cuda"), **kw), poses


def calculate_projection_wraper_with_geo_csv_file(img_3d, img_spacing, geo_path, receptor_size=None, **kw):
    """sdct_projection_utils.py:161-177: emitter positions from a CSV (first row = header), in mm."""
    geo_txt = genfromtxt(geo_path, delimiter=',')[1:]
    poses = geo_txt / img_spacing
    resolution = _resolution(img_3d.shape, receptor_size)
    return calculate_projection(img_3d, poses, resolution, [1, 1, 1], img_spacing, torch.device("cuda"), **kw), poses


def backproj_grids_with_poses(poses, img_shape, proj_shape, device=None):
    """The (B=1,P,2,D,W,H) normalised backprojection grid of sdct_projection_utils.py:227-250
    (channel 0 ↔ detector Ph axis, channel 1 ↔ Pw, i.e. after the reference's flip(2)).
    The model does not need it (the kernel derives it per voxel); kept for API parity."""
    device = _gpu(device)
    p = np.asarray(poses, dtype=np.float32)
    if p.ndim != 3:
        raise ValueError("poses must be (B,P,3)")
    g = ops.backproject_coords(p[0], img_shape, proj_shape, device, normalized=True)  # (P,D,W,H,2) (Pw,Ph)
    return g.permute(0, 4, 1, 2, 3).flip(1).unsqueeze(0).contiguous()


def backproj_grids(scan_range, proj_num, img_shape, proj_shape, device=None):
    """Pose-less variant (sdct_projection_utils.py:179-202): emitter at y = 3.0·W.  The reference's float64 pose
    array promotes the whole computation, so this returns a FLOAT64 (P,2,D,W,H) grid built as scale·g + trans
    (:194-197) — its own kernel, not the fp32 with-poses one (different type and op order)."""
    device = _gpu(device)
    poses = scan_poses(scan_range, proj_num, img_shape[1], y_scale=3.)   # float64, as poses_scale*w at :193
    return ops.backproject_coords_poseless_f64(poses, img_shape, proj_shape, device)


def forward_grids_with_poses(poses, spacing, img_shape, device=None, receptor_size=None):
    """(grids (P,Rd,Rh,W,3) flipped to (z,y,x) order, dx (P,Rd,Rh)) of sdct_projection_utils.py:252-265."""
    device = _gpu(device)
    resolution = _resolution(img_shape, receptor_size)
    poses32 = torch.from_numpy(np.asarray(poses)).type(torch.float32).numpy()
    g, dx = ops.drr_sample_coords(poses32, spacing, img_shape, resolution, device, normalized=True)
    return torch.flip(g, [4]), dx


def forward_grids(scan_range, proj_num, spacing, img_shape, device=None, receptor_size=None):
    """sdct_projection_utils.py:204-225 (emitter at y = 3.0·W)."""
    poses = scan_poses(scan_range, proj_num, img_shape[1], y_scale=3.)
    return forward_grids_with_poses(poses, spacing, img_shape, device, receptor_size)
