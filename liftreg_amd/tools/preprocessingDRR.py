#!/usr/bin/env python3
"""DRR generation for a preprocessed dataset — same flags, folder layout and output files as the
reference's tools/preprocessingDRR.py (:12-32 flags, :94-108 folders, :123-154 loop), on the HIP
projector.

  <data_path>/preprocessed/{id}_{source,target}.npy      (D,W,H) HU volumes            (input)
  <data_path>/{train,debug,val,test}/data_id.npy          case ids per phase            (input)
  <data_path>/drr/<drr_folder_name>/drr/{id}_{source,target}_proj.npy  (P,Rd,Rh) float32 (output)
  <data_path>/drr/<drr_folder_name>/drr/poses.npy         (P,3) float64                 (output)

Differences that do not change the files: the HU→attenuation conversion (`calc_relative_atten_coef`)
and the SAR→SPR `np.flip(axis=1)` (:135-136) are folded into the projector's volume load; cases are
independent, so under `torch.distributed.run` they are dealt round-robin to the ranks (one GPU each,
no collective).  `--preview` (matplotlib figures) is not built.
"""
import argparse
import os

import numpy as np
import torch

from liftreg_amd import ops, parallel
from liftreg_amd.utils.sdct_projection_utils import scan_poses, _resolution
from numpy import genfromtxt


def build_parser():
    p = argparse.ArgumentParser(description="Generate DRR for given dataset (HIP projector)")
    p.add_argument("-d", "--data_path", required=True, type=str, help="root of the preprocessed dataset")
    p.add_argument("--drr_folder_name", required=True, type=str)
    p.add_argument("--scan_range", type=float, default=30., help="scan range in degrees")
    p.add_argument("--scan_num", type=int, default=4, help="number of emitter positions")
    p.add_argument("--geo_path", type=str, default="", help="CSV with emitter positions in mm (header row)")
    p.add_argument("--receptor_w", type=int, default=0)
    p.add_argument("--receptor_h", type=int, default=0)
    p.add_argument("-g", "--gpu_id", type=int, default=0)
    p.add_argument("--phase", type=str, default="all", help="train | debug | val | test | all")
    p.add_argument("--spacing", type=float, nargs=3, default=(2.2, 2.2, 2.2))
    return p


def project_case(vol_hu, poses, resolution, spacing, dev):
    """One volume as the reference treats it: flip axis 1, HU→μ, project — all inside the kernel's load."""
    v = torch.from_numpy(np.ascontiguousarray(vol_hu, dtype=np.float32)).to(dev)
    poses32 = torch.from_numpy(np.asarray(poses)).type(torch.float32).numpy()
    return ops.drr_forward(v, poses32, resolution, spacing, hu_input=True, flip_w=True).cpu().numpy()


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, world = (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))
    local = int(os.environ.get("LOCAL_RANK", args.gpu_id if world == 1 else 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    receptor = [args.receptor_h, args.receptor_w] if args.receptor_h and args.receptor_w else None

    root = os.path.abspath(args.data_path)
    pre = os.path.join(root, "preprocessed")
    assert os.path.exists(pre), "No preprocessed folder found."
    drr_folder = os.path.join(root, "drr", args.drr_folder_name, "drr")
    os.makedirs(drr_folder, exist_ok=True)
    phases = ["train", "debug", "val", "test"]
    if args.phase in phases:
        phases = [args.phase]
    else:
        assert args.phase == "all", "Wrong phase value."

    cases = []
    for p in phases:
        ids_file = os.path.join(root, p, "data_id.npy")
        if os.path.exists(ids_file):
            cases += [str(d) for d in np.load(ids_file)]
    cases = sorted(set(cases))
    poses = None
    for i in parallel.shard_items(len(cases), world, rank):
        d = cases[i]
        vols = {kind: np.load(os.path.join(pre, f"{d}_{kind}.npy")) for kind in ("target", "source")}
        shape = vols["target"].shape
        if args.geo_path != "":
            poses = genfromtxt(args.geo_path, delimiter=',')[1:] / np.asarray(args.spacing)
        else:
            poses = scan_poses(args.scan_range, args.scan_num, shape[1])
        if vols["source"].shape == shape:
            # the case's two volumes share the geometry: ONE launch (lr_drr_forward_batch_f32; the bits of two separate ones)
            both = torch.from_numpy(np.stack([np.ascontiguousarray(vols[k], dtype=np.float32) for k in ("target", "source")])).to(dev)
            poses32 = torch.from_numpy(np.asarray(poses)).type(torch.float32).numpy()
            projs = ops.drr_forward_batch(both, poses32, _resolution(shape, receptor), args.spacing, hu_input=True,
                                          flip_w=True).cpu().numpy()
            for k, kind in enumerate(("target", "source")):
                np.save(os.path.join(drr_folder, f"{d}_{kind}_proj.npy"), projs[k])
        else:
            for kind in ("target", "source"):
                if args.geo_path == "":
                    poses = scan_poses(args.scan_range, args.scan_num, vols[kind].shape[1])
                proj = project_case(vols[kind], poses, _resolution(vols[kind].shape, receptor), args.spacing, dev)
                np.save(os.path.join(drr_folder, f"{d}_{kind}_proj.npy"), proj)
    if rank == 0 and cases:
        if poses is None:  # this rank owned no case (more ranks than cases): same geometry for every case
            vol = np.load(os.path.join(pre, f"{cases[0]}_target.npy"), mmap_mode="r")
            poses = (genfromtxt(args.geo_path, delimiter=',')[1:] / np.asarray(args.spacing) if args.geo_path != ""
                     else scan_poses(args.scan_range, args.scan_num, vol.shape[1]))
        np.save(os.path.join(drr_folder, "poses.npy"), poses)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
