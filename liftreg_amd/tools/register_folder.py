#!/usr/bin/env python3
"""Register every case of a LiftReg data folder on the GPU and write the deformation maps the reference's
evaluation scripts read.

Data contract (the reference's own, nothing new):
  <root>/<phase>/data_id.npy                       case ids                      (dataset/Registration2D3DDataset.py:60-70)
  <root>/preprocessed/{id}_{source,target}.npy     CT volumes in HU (D,W,H)      (:82,100; flipped along axis 1 on load)
  <root>/preprocessed/{id}_{source,target}_seg.npy optional label maps           (:88,106)
  <root>/drr/<name>/drr/{id}_target_proj.npy       DRR views (P,Rd,Rh)           (:110; tools/preprocessingDRR.py output)
  <root>/drr/<name>/drr/poses.npy                  emitter poses (P,3)           (:120)
  <out>/{id}_phi.npy                               (phi+1)/2, float32 (3,D,W,H)  (utils/utils.py:57-68 save_deformations)

What the reference does per case on the host (flip, clip-range normalisation of volumes [-1000,0] and views (0,6),
`_read_case` :82-112) runs here as GPU prologues (`ops.normalize_clip`); the model is the plugin class
`liftreg_amd.models.LiftRegDeformSubspaceBackproj.model` with an optional reference checkpoint (`state_dict` keys are
identical).  This is an I/O wrapper around the hot path, not a replacement for the reference's training harness.

  python -m liftreg_amd.tools.register_folder -d DATA --drr_folder_name NAME --phase test -o OUT --pca_path PCA_DIR \\
         [--checkpoint model_best.pth.tar] [--latent_dim 56] [--batch 4] [--conv_dtype fp32|bf16] [--labels]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("-d", "--data_path", required=True, help="dataset root (contains preprocessed/, <phase>/, drr/)")
    p.add_argument("--drr_folder_name", required=True)
    p.add_argument("--phase", default="test")
    p.add_argument("-o", "--output_path", required=True)
    p.add_argument("--pca_path", required=True, help='folder with pca_vectors.npy / pca_mean.npy, or "synthetic[:seed]"')
    p.add_argument("--checkpoint", default="", help="reference-format checkpoint (a dict with 'state_dict', or a bare state dict)")
    p.add_argument("--latent_dim", type=int, default=56)
    p.add_argument("--batch", type=int, default=4)
    p.add_argument("--conv_dtype", default="fp32", choices=("fp32", "bf16"))
    p.add_argument("--labels", action="store_true", help="also warp {id}_source_seg.npy (nearest) and report Dice vs the target label")
    p.add_argument("--gpu", type=int, default=0)
    return p


def load_case(root, drr_dir, cid, dev, labels):
    """One dataset sample on the GPU (the reference's _read_case + ToTensor), volumes as (1,D,W,H)."""
    from .. import ops

    def volume(kind):
        hu = torch.from_numpy(np.load(os.path.join(root, "preprocessed", f"{cid}_{kind}.npy")).astype(np.float32)).to(dev)
        return ops.normalize_clip(torch.flip(hu, dims=(1,)).contiguous(), -1000.0, 0.0)[None]

    sample = {"source": volume("source"), "target": volume("target")}
    proj = torch.from_numpy(np.load(os.path.join(drr_dir, f"{cid}_target_proj.npy")).astype(np.float32)).to(dev)
    sample["target_proj"] = ops.normalize_clip(proj, 0.0, 6.0)
    if labels:
        for kind in ("source", "target"):
            seg = np.flip(np.load(os.path.join(root, "preprocessed", f"{cid}_{kind}_seg.npy")).astype(np.float32), axis=1)
            sample[f"{kind}_label"] = torch.from_numpy(seg.copy()).to(dev)[None]
    return sample


def main(argv=None):
    args = build_parser().parse_args(argv)
    from ..layers.losses import NCCLoss
    from ..models.LiftRegDeformSubspaceBackproj import model
    from ..utils.metrics import cal_metric
    from ..utils.net_utils import Bilinear
    from ..utils.utils import save_deformations

    dev = torch.device("cuda", args.gpu)
    torch.cuda.set_device(dev)
    ids = [str(s) for s in np.load(os.path.join(args.data_path, args.phase, "data_id.npy"))]
    drr_dir = os.path.join(args.data_path, "drr", args.drr_folder_name, "drr")
    poses = np.load(os.path.join(drr_dir, "poses.npy")).astype(np.float32)
    os.makedirs(args.output_path, exist_ok=True)
    if not ids:
        print(json.dumps({"cases": 0}))
        return 0

    first = load_case(args.data_path, drr_dir, ids[0], dev, args.labels)
    img_sz = list(first["source"].shape[1:])
    net = model(img_sz, {"drr_feature_num": int(poses.shape[0]), "latent_dim": args.latent_dim, "pca_path": args.pca_path,
                         "conv_dtype": args.conv_dtype}).to(dev).eval()
    if args.checkpoint:
        ck = torch.load(args.checkpoint, map_location="cpu")
        net.load_state_dict(ck.get("state_dict", ck), strict=True)
    sim = NCCLoss(check_nan=False)
    nearest = Bilinear(zero_boundary=True, using_scale=False, mode="nearest")
    report, t_gpu = [], 0.0
    with torch.no_grad():
        for i in range(0, len(ids), args.batch):
            chunk = ids[i:i + args.batch]
            samples = [first if (i == 0 and k == 0) else load_case(args.data_path, drr_dir, c, dev, args.labels)
                       for k, c in enumerate(chunk)]
            batch = {k: torch.stack([s[k] for s in samples]).contiguous() for k in samples[0]}
            batch["target_poses"] = torch.from_numpy(np.broadcast_to(poses, (len(chunk),) + poses.shape).copy())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = net(batch)
            torch.cuda.synchronize()
            t_gpu += time.perf_counter() - t0
            save_deformations(out["phi"], chunk, args.output_path)
            for k, c in enumerate(chunk):
                row = {"id": c, "ncc": 1.0 - float(sim(out["warped"][k:k + 1], out["target"][k:k + 1]))}
                if args.labels:   # evaluate_dir_lab.py:216-221: warp the source label map, nearest neighbour, then Dice
                    wseg = nearest(batch["source_label"][k:k + 1], out["phi"][k:k + 1])
                    row["dice"] = cal_metric(wseg, batch["target_label"][k:k + 1])["dice"]
                report.append(row)
    with open(os.path.join(args.output_path, "register_folder.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps({"cases": len(ids), "registrations_per_s_gpu": len(ids) / t_gpu, "image_size": img_sz,
                      "views": int(poses.shape[0]), "mean_ncc": float(np.mean([r["ncc"] for r in report]))}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
