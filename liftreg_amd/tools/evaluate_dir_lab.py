#!/usr/bin/env python3
"""Landmark target-registration error (TRE) of a deformation map — the sampler half of the reference's
tools/evaluate_dir_lab.py (`readPoint` :18-43, `calc_warped_points` :46-59, `eval_with_file` :61-79,
`eval_with_data` :81-138), same function names, argument meaning and return values.

The one piece of device arithmetic is `calc_warped_points`: `F.grid_sample(phi, points, align_corners=True)` on
DOUBLES (trilinear, zeros padding).  It runs as `lr_sample_points_f64` (csrc/metrics.hip) — ≈300 points, so this is
about results identical to the reference's, not about speed; everything around it (landmark → phi coordinates, the
SAR→SPR axis-1 reversal, distances) is the reference's float64 host arithmetic on (N,3) arrays and stays there.
There is no CPU fallback for the sampler: without a GPU or the HIP library this raises.

  python -m liftreg_amd.tools.evaluate_dir_lab --source S.txt --target T.txt --phi case_phi.npy \\
         --dim 256 256 256 --spacing 0.97 0.97 2.5 --phi_spacing 1.4 1.4 1.4 [--origin 0 0 0]
(`case_phi.npy` = the `(phi+1)/2` file `save_deformations` writes, which is what the reference's script reads too.)
"""
import argparse
import json

import numpy as np
import torch

from .. import ops


def readPoint(f_path):
    """DirLab landmark file: one `x<TAB>y<TAB>z` row per landmark, a trailing newline (evaluate_dir_lab.py:18-43;
    like the reference, the last element of the split — the empty string after that newline — is not a point)."""
    with open(f_path) as fp:
        content = fp.read().split('\n')
    count = len(content) - 1
    points = np.ndarray([count, 3], dtype=np.float32)
    n = 0
    for i in range(count):
        if content[i] == "":
            break
        temp = content[i].split('\t')
        points[i, 0], points[i, 1], points[i, 2] = float(temp[0]), float(temp[1]), float(temp[2])
        n += 1
    return points[:n] if n < count else points


def calc_warped_points(source_list_t, phi_t, dim, spacing, phi_spacing, device=None):
    """evaluate_dir_lab.py:46-59.  source_list_t: (1,1,1,N,3) float64 normalised landmark positions (x,y,z);
    phi_t: (1,3,D,W,H) float64 map; returns the (N,3) float64 CPU tensor of warped positions in mm."""
    device = torch.device("cuda") if device is None else torch.device(device)
    pts = torch.as_tensor(source_list_t, dtype=torch.float64).reshape(-1, 3).to(device)
    phi = torch.as_tensor(phi_t, dtype=torch.float64)
    if phi.dim() != 5 or phi.shape[0] != 1:
        raise ValueError("phi_t must be (1,C,D,W,H)")
    sampled = ops.sample_points_f64(phi[0].to(device).contiguous(), pts.contiguous())     # (N,C) = grid_sample(...)[0,:,0,0,:].T
    warped = torch.flip(sampled, [1]).cpu()                                               # torch.flip(..., [4]) (:55)
    return torch.mul(torch.mul(warped, torch.from_numpy(np.asarray(dim) - 1.)), torch.from_numpy(np.asarray(phi_spacing)))


def eval_with_data(source_list, target_list, phi, dim, spacing, origin, phi_spacing, plot_result=False, device=None):
    """evaluate_dir_lab.py:81-138 → (mean TRE in mm, [mean |dx|, mean |dy|, mean |dz|])."""
    dim, spacing, phi_spacing = (np.asarray(v, dtype=np.float64) for v in (dim, spacing, phi_spacing))
    origin_list = np.repeat([origin, ], target_list.shape[0], axis=0)
    target_list_t = torch.from_numpy((target_list - 1.) * spacing) - origin_list * phi_spacing
    source_list_t = torch.from_numpy((source_list - 1.) * spacing) - origin_list * phi_spacing
    # landmarks are SAR, the model's volumes SPR: reverse axis 1 (:98-103)
    target_list_t[:, 1] = (dim[1] - 1) * phi_spacing[1] - target_list_t[:, 1]
    source_list_t[:, 1] = (dim[1] - 1) * phi_spacing[1] - source_list_t[:, 1]
    source_list_norm = source_list_t / phi_spacing / (dim - 1.) * 2.0 - 1.0
    source_list_norm = source_list_norm.unsqueeze(0).unsqueeze(0).unsqueeze(0)
    phi_t = torch.from_numpy(np.asarray(phi)).double()
    warped_list_t = calc_warped_points(source_list_norm, phi_t, dim, spacing, phi_spacing, device=device)
    dist = torch.nn.PairwiseDistance(p=2)(target_list_t, warped_list_t)
    dist_x = torch.mean(torch.abs(target_list_t[:, 0] - warped_list_t[:, 0])).item()
    dist_y = torch.mean(torch.abs(target_list_t[:, 1] - warped_list_t[:, 1])).item()
    dist_z = torch.mean(torch.abs(target_list_t[:, 2] - warped_list_t[:, 2])).item()
    return torch.mean(dist).item(), [dist_x, dist_y, dist_z]


def eval_with_file(source_file, target_file, phi_file, dim, spacing, origin, phi_spacing, plot_result=False, device=None):
    """evaluate_dir_lab.py:61-79."""
    source_list, target_list = readPoint(source_file), readPoint(target_file)
    phi = np.expand_dims(np.load(phi_file), axis=0)
    return eval_with_data(source_list, target_list, phi, dim, spacing, origin, phi_spacing, plot_result, device=device)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--source", required=True)
    ap.add_argument("--target", required=True)
    ap.add_argument("--phi", required=True, help="(3,D,W,H) .npy map in [0,1] units, as save_deformations writes it")
    ap.add_argument("--dim", type=float, nargs=3, required=True)
    ap.add_argument("--spacing", type=float, nargs=3, required=True)
    ap.add_argument("--phi_spacing", type=float, nargs=3, required=True)
    ap.add_argument("--origin", type=float, nargs=3, default=[0., 0., 0.])
    a = ap.parse_args(argv)
    phi = np.load(a.phi)
    res, sep = eval_with_data(readPoint(a.source), readPoint(a.target), phi[None], np.array(a.dim), np.array(a.spacing),
                              np.array(a.origin), np.array(a.phi_spacing))
    print(json.dumps({"tre_mm": res, "abs_xyz_mm": sep}))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
