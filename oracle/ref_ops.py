"""oracle/ref_ops.py — TEST INFRASTRUCTURE ONLY.

CPU restatement of the LiftReg hot path (uncbiag/LiftReg) with the SAME ATen op
sequence the reference executes (F.grid_sample 2-D/3-D align_corners=True,
conv3d + LeakyReLU(0.2), linear, mean/sqrt reductions), written against this
repo's own data flow.  Every function cites the reference file:line it follows
(paths relative to the reference checkout).

Used as (i) the checker in tests/ and __graft_entry__.smoke(), and (ii) the
`cpu_baseline` ("kind": "port") leg of bench.py — it is what the reference's
Python would run on the host cores.  liftreg_amd never imports it.

Parity status: PINNED — tests/test_oracle_golden.py checks every function here
against golden vectors produced by importing the reference itself
(tests/golden/make_golden.py), bit-for-bit for sampling grids and indices.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- a1
def calc_relative_atten_coef(img):
    """src/liftreg/utils/sdct_projection_utils.py:6-9."""
    new_img = np.asarray(img).astype(np.float32).copy()
    new_img[new_img < -1000] = -1000
    return (new_img + 1000.) / 1000. * 0.2


# --------------------------------------------------------------------------- a2
def scan_poses(scan_range, proj_num, W, y_scale=3.5):
    """Emitter poses of calculate_projection_wraper (sdct_projection_utils.py:138-145,155).

    Returns float64 (P,3) in voxel units: poses_scale * W.
    """
    angle_half = scan_range / 2.
    poses_scale = np.ndarray((proj_num, 3), dtype=float)
    poses_scale[:, 1] = y_scale
    poses_scale[:, 0] = np.tan(np.linspace(-angle_half, angle_half, num=proj_num) / 180. * np.pi) * 3.
    poses_scale[:, 2] = np.linspace(-0.2, 0.2, num=proj_num)
    return poses_scale * W


# --------------------------------------------------------------------------- a3
def project_grid(poses, resolution, obj_shape, spacing, dtype=torch.float32):
    """Ray/plane sample grid and dx of project_grid_multi (sdct_projection_utils.py:15-57).

    Same per-element fp32 ops as the reference (its two K=1 matmuls at :50-51 are
    plain products), expressed with broadcasting.  Returns grid (P,Rd,Rh,W,3) in
    the reference's (x,y,z)=(D,W,H) order, normalised; dx (P,Rd,Rh).
    """
    d, w, h = obj_shape
    res_d, res_h = resolution
    e = torch.from_numpy(np.asarray(poses)).type(dtype)                    # I0 (:28)
    lin_x = torch.linspace(-res_d / 2, res_d / 2 - 1, steps=res_d)          # :32
    lin_y = torch.linspace(-res_h / 2, res_h / 2 - 1, steps=res_h)          # :33
    I = torch.zeros((res_d, res_h, 3), dtype=dtype)
    I[:, :, 0] = lin_x[:, None]
    I[:, :, 2] = lin_y[None, :]
    I = torch.add(I, -e[:, None, None, :])                                   # (P,Rd,Rh,3) :38
    dx = torch.mul(I, 1. / I[..., 1:2])                                      # :39
    I = I / torch.norm(I, dim=3, keepdim=True)                               # :40
    dx = torch.norm(dx * torch.as_tensor(spacing, dtype=dtype)[None, None], dim=3)  # :41
    yj = torch.linspace(0, w - 1, w, dtype=dtype)                            # P0 rows (:24-26)
    t = (1. / I[..., 1])[..., None] * (yj[None, :] - e[:, 1:2])[:, None, None, :]   # T (:50)
    grid = I[..., None, :] * t[..., None] + e[:, None, None, None, :]        # :51
    grid[..., 0] = grid[..., 0] / d * 2.0                                    # :54
    grid[..., 1] = (grid[..., 1] - 0.) / (w - 1.) * 2.0 + -1.                # :55
    grid[..., 2] = grid[..., 2] / h * 2.0                                    # :56
    return grid, dx


# --------------------------------------------------------------------------- a4
def drr_forward(img, poses, resolution, spacing):
    """calculate_projection (sdct_projection_utils.py:59-100) on the CPU.

    img: (D,W,H) fp32 attenuation volume (numpy); returns (P,Rd,Rh) fp32 numpy.
    """
    I0 = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))[None, None]
    grids, dx = project_grid(poses, resolution, I0.shape[2:], spacing, I0.dtype)
    grids = torch.flip(grids, [4])                                           # :76
    p, rd, rh, w = grids.shape[:4]
    samp = F.grid_sample(I0, grids.reshape(1, 1, 1, -1, 3), align_corners=True)     # :81
    out = torch.mul(torch.sum(samp.reshape(1, p, rd, rh, w), dim=4), dx).float()
    out *= 0.1                                                               # :85
    return out[0].numpy()


# --------------------------------------------------------------------------- a6
def backproj_grid(poses, img_shape, proj_shape):
    """backproj_grids_with_poses (sdct_projection_utils.py:227-250); poses (1,P,3) fp32 numpy.

    Returns (1,P,2,D,W,H): channel 0 pairs with the detector's Ph axis (grid x),
    channel 1 with Pw (grid y) — i.e. after the reference's flip(2).
    """
    d, w, h = img_shape
    proj_w, proj_h = proj_shape
    x = torch.linspace(-d / 2, d / 2 - 1, d)
    y = torch.linspace(w - 1, 0, w)
    z = torch.linspace(-h / 2, h / 2 - 1, h)
    gx, gy, gz = torch.meshgrid(x, y, z, indexing="ij")
    ps = torch.from_numpy(np.asarray(poses))[..., None, None, None]          # (1,P,3,1,1,1)
    scale = ps[:, :, 1:2] / (ps[:, :, 1:2] - gy)                              # :239
    grids = torch.stack((gx, gz), dim=0)[None]                                # (1,2,D,W,H)
    grids = grids - ps[:, :, ::2]                                             # :241
    grids = torch.mul(grids, scale) + ps[:, :, ::2]                           # :242
    grids[:, :, 0] = grids[:, :, 0] / proj_w * 2.0                            # :247
    grids[:, :, 1] = grids[:, :, 1] / proj_h * 2.0                            # :248
    return grids.flip(2)                                                      # :250


def backproj_grid_poseless(scan_range, proj_num, img_shape, proj_shape):
    """backproj_grids (sdct_projection_utils.py:179-202), the pose-less variant: emitter at y = 3.0·W, and — because
    `poses` is a float64 array meeting float32 linspaces — a FLOAT64 grid built as scale·g + trans (:194-197).
    Returns (P,2,D,W,H) float64 after the reference's flip(1)."""
    d, w, h = img_shape
    proj_w, proj_h = proj_shape
    x = torch.linspace(-d / 2, d / 2 - 1, d)
    y = torch.linspace(w - 1, 0, w)
    z = torch.linspace(-h / 2, h / 2 - 1, h)
    gx, _gy, gz = torch.meshgrid(x, y, z, indexing="ij")
    poses = torch.from_numpy(scan_poses(scan_range, proj_num, w, y_scale=3.))            # float64 (:193)
    scale = poses[:, 1:2] / (poses[:, 1:2] - y)                                          # (P,W) :194
    trans = poses[:, 0::2, None] * (-y / (poses[:, 1:2] - y)).reshape(proj_num, 1, w)    # (P,2,W) :195
    grids = torch.cat((gx[None, :], gz[None, :]), dim=0).unsqueeze(0)                    # (1,2,D,W,H) :196
    grids = torch.mul(scale.reshape(proj_num, 1, 1, w, 1), grids) + trans.reshape(proj_num, 2, 1, w, 1)   # :197
    grids[:, 0] = grids[:, 0] / proj_w * 2.0
    grids[:, 1] = grids[:, 1] / proj_h * 2.0
    return grids.flip(1)


# --------------------------------------------------------------------------- a5
def _default_resolution(img_shape, receptor_size):
    if receptor_size is not None:
        return list(receptor_size)
    return [int(img_shape[0] * 1.5), int(img_shape[2] * 1.5)]


def forward_grids_with_poses(poses, spacing, img_shape, receptor_size=None):
    """sdct_projection_utils.py:252-265 → (grids (P,Rd,Rh,W,3) flipped to (z,y,x), dx)."""
    grid, dx = project_grid(poses, _default_resolution(img_shape, receptor_size), img_shape,
                            torch.tensor(spacing).float(), torch.float32)
    return torch.flip(grid, [4]), dx


def forward_grids(scan_range, proj_num, spacing, img_shape, receptor_size=None):
    """sdct_projection_utils.py:204-225: as above with the emitter at y = 3.0·W (:208), not 3.5."""
    return forward_grids_with_poses(scan_poses(scan_range, proj_num, img_shape[1], y_scale=3.), spacing, img_shape,
                                    receptor_size)


def csv_geometry_poses(geo_rows_mm, img_spacing):
    """calculate_projection_wraper_with_geo_csv_file (sdct_projection_utils.py:161-163): emitter positions of the CSV
    (header row already dropped) in mm → voxel units, float64."""
    return np.asarray(geo_rows_mm, dtype=np.float64) / np.asarray(img_spacing)


# --------------------------------------------------------------------------- f4 (tools/evaluate_dir_lab.py)
def calc_warped_points(source_list_t, phi_t, dim, phi_spacing):
    """tools/evaluate_dir_lab.py:46-59: grid_sample of the float64 map at the normalised landmarks, channel flip,
    × (dim-1) × phi_spacing."""
    warped = F.grid_sample(phi_t, source_list_t, align_corners=True)
    warped = torch.flip(warped.permute(0, 2, 3, 4, 1), [4])[0, 0, 0]
    return torch.mul(torch.mul(warped, torch.from_numpy(np.asarray(dim) - 1.)), torch.from_numpy(np.asarray(phi_spacing)))


def landmark_tre(source_list, target_list, phi, dim, spacing, origin, phi_spacing):
    """eval_with_data (tools/evaluate_dir_lab.py:81-138) → (mean TRE, [|dx|,|dy|,|dz|] means, warped (N,3))."""
    dim, spacing, phi_spacing = (np.asarray(v, dtype=np.float64) for v in (dim, spacing, phi_spacing))
    origin_list = np.repeat([origin, ], target_list.shape[0], axis=0)
    t = torch.from_numpy((target_list - 1.) * spacing) - torch.from_numpy(origin_list * phi_spacing)
    s = torch.from_numpy((source_list - 1.) * spacing) - torch.from_numpy(origin_list * phi_spacing)
    t[:, 1] = (dim[1] - 1) * phi_spacing[1] - t[:, 1]
    s[:, 1] = (dim[1] - 1) * phi_spacing[1] - s[:, 1]
    s_norm = (s / torch.from_numpy(phi_spacing) / torch.from_numpy(dim - 1.) * 2.0 - 1.0)[None, None, None]
    w = calc_warped_points(s_norm, torch.from_numpy(np.asarray(phi)).double(), dim, phi_spacing)
    dist = torch.nn.PairwiseDistance(p=2)(t, w)
    return (torch.mean(dist).item(), [torch.mean(torch.abs(t[:, i] - w[:, i])).item() for i in range(3)], w, s_norm)


# --------------------------------------------------------------------------- a7
def backproject(target_proj, poses, img_shape):
    """Backprojection of model._estimate_flow (LiftRegDeformSubspaceBackproj.py:85-93).

    target_proj (B,P,Pw,Ph) tensor; poses (B,P,3) tensor/array — geometry of batch
    element 0 is used for every sample, as the reference caches it (:85-87).
    """
    B, P, pw, ph = target_proj.shape
    D, W, H = img_shape
    p0 = np.asarray(poses, dtype=np.float32)[0:1]
    g = backproj_grid(p0, (D, W, H), (pw, ph)).permute(0, 1, 3, 4, 5, 2)     # (1,P,D,W,H,2)
    return F.grid_sample(target_proj.reshape(B * P, 1, pw, ph),
                         g.expand(B, -1, -1, -1, -1, -1).reshape(B * P, D * W, H, -1),
                         align_corners=True, padding_mode="zeros").reshape(B, P, D, W, H).detach()


# --------------------------------------------------------------------------- a8/a9
def conv_block(x, weight, bias, stride, slope=0.2):
    """convBlock (layers/layers.py:335-372): Conv3d(k3,p1) + LeakyReLU(0.2)."""
    return F.leaky_relu(F.conv3d(x, weight, bias, stride=stride, padding=1), slope)


def fc_block(x, weight, bias, slope=0.2):
    """FullyConnectBlock (layers/layers.py:413-439); slope=None → no nonlinearity."""
    y = F.linear(x, weight, bias)
    return y if slope is None else F.leaky_relu(y, slope)


# --------------------------------------------------------------------------- a10
def pca_reconstruct(coefs, pca_vectors_LxM, pca_mean):
    """F.linear(x, pca_vectors, pca_mean) (…Backproj.py:42-43,102); basis given as the (L,3V) file."""
    return F.linear(coefs, pca_vectors_LxM.T, pca_mean)


# --------------------------------------------------------------------------- a11
def identity_axis_tables(sz):
    """The three 1-D tables whose outer broadcast is identity_map(sz) (net_utils.py:59-87).

    Follows the reference's numpy arithmetic: float32 index * float64 spacing
    (rounded back to float32 under numpy>=2 promotion rules), then *2-1 in float32.
    """
    spacing = 1. / (np.array(sz) - 1)
    tabs = []
    for d in range(3):
        t = np.arange(sz[d]).astype(np.float32)
        t *= spacing[d]
        t = t * 2 - 1
        tabs.append(t.astype(np.float32))
    return tabs


def identity_map(sz):
    """identity_map (net_utils.py:59-87) → (3,D,W,H) float32 tensor."""
    idm = np.mgrid[0:sz[0], 0:sz[1], 0:sz[2]]
    idm = np.array(idm.astype(np.float32))
    spacing = 1. / (np.array(sz) - 1)
    for d in range(3):
        idm[d] *= spacing[d]
        idm[d] = idm[d] * 2 - 1
    return torch.from_numpy(idm.astype(np.float32))


# --------------------------------------------------------------------------- a12
def warp(img, phi, zero_boundary=True, using_scale=True, mode="bilinear"):
    """Bilinear.forward (net_utils.py:26-56)."""
    ordered = torch.zeros_like(phi)
    ordered[:, 0] = phi[:, 2]
    ordered[:, 1] = phi[:, 1]
    ordered[:, 2] = phi[:, 0]
    pad = "zeros" if zero_boundary else "border"
    src = (img + 1) / 2 if using_scale else img
    out = F.grid_sample(src, ordered.permute(0, 2, 3, 4, 1), padding_mode=pad, mode=mode,
                        align_corners=True)
    return out * 2 - 1 if using_scale else out


# --------------------------------------------------------------------------- a13
def ncc_loss(inp, target):
    """Configured NCCLoss (layers/losses.py:14-29)."""
    inp = inp.reshape(inp.shape[0], -1)
    target = target.reshape(target.shape[0], -1)
    a = inp - torch.mean(inp, 1).view(inp.shape[0], 1) + 1e-10
    b = target - torch.mean(target, 1).view(inp.shape[0], 1) + 1e-10
    ncc = ((a * b).mean(1)) / torch.sqrt(((a ** 2).mean(1)) * ((b ** 2).mean(1)))
    return 1 - ncc.mean()


def ncc_loss_squared(x, y):
    """NCCLoss variant (layers/layers.py:238-255)."""
    n_batch = x.shape[0]
    shp = [x.shape[0], x.shape[1], -1]
    x = x.reshape(*shp)
    y = y.reshape(*shp)
    xm = x - x.mean(dim=2, keepdim=True)
    ym = y - y.mean(dim=2, keepdim=True)
    ncc_sqr = ((xm * ym).mean(dim=2) ** 2) / ((xm ** 2).mean(dim=2) * (ym ** 2).mean(dim=2) + 1e-12)
    return 1. - ncc_sqr.mean(dim=1).sum() / n_batch


# --------------------------------------------------------------------------- a14
def model_forward(params, inp, pca_vectors_LxM, pca_mean, strides=(1, 2, 2, 2, 2, 2), conv_dtype="fp32",
                  grad_dtype="fp32"):
    """model.forward (LiftRegDeformSubspaceBackproj.py:49-104) on CPU tensors.

    params: state-dict-like {encoders.i.conv.weight/bias, encoders.6.{1,2,3}.fc.weight/bias}.
    The FC1 width is whatever params hold (the reference hard-codes 32*5^3 for 160^3, :36).
    """
    moving, target, target_proj = inp["source"], inp["target"], inp["target_proj"]
    if "source_label" in inp:
        moving_cp = (moving + 1) * inp["source_label"] - 1
        target_cp = (target + 1) * inp["target_label"] - 1
    else:
        moving_cp, target_cp = moving, target
    B, _, D, W, H = moving.shape
    tv = backproject(target_proj, inp["target_poses"], (D, W, H))
    x = torch.cat([moving, tv], dim=1)
    if conv_dtype == "bf16":      # the build's bf16 storage variant (not in the reference): see encoder_bf16
        x = encoder_bf16(params, x, strides, grad_bf16=(grad_dtype == "bf16"))
    else:
        for i, s in enumerate(strides):
            x = conv_block(x, params[f"encoders.{i}.conv.weight"], params[f"encoders.{i}.conv.bias"], s)
    x = x.flatten(1)
    x = fc_block(x, params["encoders.6.1.fc.weight"], params["encoders.6.1.fc.bias"])
    x = fc_block(x, params["encoders.6.2.fc.weight"], params["encoders.6.2.fc.bias"])
    coefs = fc_block(x, params["encoders.6.3.fc.weight"], params["encoders.6.3.fc.bias"], slope=None)
    disp = pca_reconstruct(coefs, pca_vectors_LxM, pca_mean).reshape(B, 3, D, W, H)
    phi = disp + identity_map((D, W, H))
    warped = warp(moving_cp, phi, zero_boundary=True, using_scale=True)
    return {"warped": warped, "phi": phi, "params": disp, "target": target_cp, "pca_coefs": coefs,
            "target_proj": target_proj, "warped_proj": target_proj, "target_volume": tv}


# --------------------------------------------------------------------------- a16
def sigmoid_decay(ep, static=5, k=5):
    """utils/utils.py:93-107."""
    if ep < static:
        return float(1.)
    return float(k / (k + np.exp((ep - static) / k)))


def disp_reg(disp):
    """compute_reg_loss (losses/SubspaceLoss.py:51-67).  PARITY UNPINNED — mermaid 0.3.2 is absent; ASSUMED
    stencil: dXc = (I[x+1]-I[x-1])*0.5/spacing, linearly extrapolated faces (one-sided differences there),
    spacing = 2/(shape-1) (the reference passes FD_torch(spacing*2) with spacing = 1/(shape-1))."""
    total = torch.zeros(disp.shape[0], *disp.shape[2:], dtype=disp.dtype)
    for c in range(3):
        f = disp[:, c]
        for ax in (1, 2, 3):
            n = f.shape[ax]
            inv_h = 0.5 * (n - 1)
            g = torch.zeros_like(f)
            sl = lambda a, b: tuple(slice(a, b) if i == ax else slice(None) for i in range(4))
            g[sl(1, n - 1)] = (f[sl(2, n)] - f[sl(0, n - 2)]) * (0.5 * inv_h)
            g[sl(0, 1)] = (f[sl(1, 2)] - f[sl(0, 1)]) * inv_h
            g[sl(n - 1, n)] = (f[sl(n - 1, n)] - f[sl(n - 2, n - 1)]) * inv_h
            total = total + g ** 2
    return total.mean()


def subspace_loss(output, epoch, initial_reg_factor=0.01, min_reg_factor=0.01, reg_factor_decay_from=2):
    """loss.forward (losses/SubspaceLoss.py:20-49) with the configured NCC similarity."""
    sim = ncc_loss(output["warped"], output["target"])
    reg = disp_reg(output["params"])
    factor = float(max(sigmoid_decay(epoch, static=reg_factor_decay_from, k=2) * initial_reg_factor, min_reg_factor))
    return {"total_loss": sim + factor * reg, "sim_loss": float(sim.detach()), "reg_loss": float(reg.detach())}


# --------------------------------------------------------------------------- f3 / f4 (prologue, evaluation)
def normalize_clip(img, lo, hi):
    """_normalize_intensity(linear_clip=True, clip_range=[lo,hi]) (dataset/Registration2D3DDataset.py:196-199,207),
    float32 arithmetic like numpy's with python-int scalars."""
    x = np.array(img, dtype=np.float32, copy=True)
    x[x < lo] = lo
    x[x > hi] = hi
    x = (x - np.float32(lo)) / np.float32(hi - lo)
    return x * np.float32(2) - np.float32(1)


def cal_metric(label_pred, label_gt, label=1):
    """utils/metrics.py:83-121 from the three set sizes."""
    eps = 1e-11
    p, g = np.asarray(label_pred).ravel() == label, np.asarray(label_gt).ravel() == label
    n_p, n_g, n_b = int(p.sum()), int(g.sum()), int((p & g).sum())
    tp, fn, fp, union = float(n_b), float(n_g - n_b), float(n_p - n_b), n_p + n_g - n_b
    if n_g != 0:
        return {"iou": tp / (float(union) + eps), "dice": 2 * tp / (2 * tp + fn + fp + eps),
                "recall": tp / (tp + fn + eps), "precision": tp / (tp + fp + eps)}
    v = 0. if n_p > 0 else 1.
    return {"iou": v, "dice": v, "recall": v, "precision": v}


def _fd_c(f, axis, h):
    """ASSUMED mermaid stencil (see disp_reg): central differences, one-sided at the two faces."""
    f = np.asarray(f, dtype=np.float32)
    d = np.empty_like(f)
    n = f.shape[axis]
    sl = lambda a, b: tuple(slice(a, b) if ax == axis else slice(None) for ax in range(f.ndim))
    ih = np.float32(1.0 / h)
    d[sl(1, n - 1)] = (f[sl(2, n)] - f[sl(0, n - 2)]) * (np.float32(0.5) * ih)
    d[sl(0, 1)] = (f[sl(1, 2)] - f[sl(0, 1)]) * ih
    d[sl(n - 1, n)] = (f[sl(n - 1, n)] - f[sl(n - 2, n - 1)]) * ih
    return d


def compute_jacobi_map(phi, spacing, use_01=False):
    """utils/utils.py:20-55 (values over the whole volume, as the reference returns).  PARITY UNPINNED (mermaid FD_np)."""
    phi = np.asarray(phi, dtype=np.float32)
    sp = np.asarray(spacing, dtype=np.float64) * (1.0 if use_01 else 2.0)
    m = [[_fd_c(phi[:, c], ax + 1, np.float32(sp[ax])) for ax in range(3)] for c in range(3)]
    (a, b, c), (d, e, f), (g, h, i) = m
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    neg = det < 0
    return float(-(det[neg].astype(np.float64)).sum()) / phi.shape[0], float(neg.sum()) / phi.shape[0]


# --------------------------------------------------------------------------- bf16 storage variant (configs C4/C5)
def _bf16(t):
    """Round to nearest-even bfloat16 and back (the values a bf16 tensor holds).  Under autograd the cast is
    straight-through (the gradient passes unrounded): where the build rounds GRADIENTS it says so explicitly
    (_RoundGradBf16), so the two training variants stay distinguishable."""
    r = t.detach().to(torch.bfloat16).to(torch.float32)
    return t + (r - t.detach()) if t.requires_grad else r


class _RoundGradBf16(torch.autograd.Function):
    """Identity whose backward rounds the gradient to bf16 (the storage of the bf16-gradient training variant)."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


def conv_block_bf16(x, weight, bias, stride, slope=0.2, round_out=True, grad_bf16=False):
    """The numerics contract of lr_conv3d_k3_lrelu_bf16: bf16-representable inputs, weights rounded to bf16,
    exact products accumulated in fp32, fp32 bias + LeakyReLU, output rounded to bf16 (not for the last block).
    grad_bf16: the gradient w.r.t. the pre-activation is rounded to bf16 before it goes on (grad_dtype="bf16")."""
    pre = F.conv3d(_bf16(x), _bf16(weight), bias, stride=stride, padding=1)
    if grad_bf16:
        pre = _RoundGradBf16.apply(pre)
    y = F.leaky_relu(pre, slope)
    return _bf16(y) if round_out else y


def encoder_bf16(params, x, strides=(1, 2, 2, 2, 2, 2), grad_bf16=False):
    """conv_dtype="bf16" of the model: every block as conv_block_bf16 (block 0 rounds its fp32 input on the way in);
    the last block's output stays fp32."""
    for i, s in enumerate(strides):
        w, b = params[f"encoders.{i}.conv.weight"], params[f"encoders.{i}.conv.bias"]
        x = conv_block_bf16(x, w, b, s, round_out=(i != len(strides) - 1), grad_bf16=grad_bf16)
    return x
