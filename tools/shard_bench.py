#!/usr/bin/env python3
"""Cost of the z-slab decomposition itself: the slab-sharded forward with N virtual ranks on ONE GPU (LocalComm: the
halo exchange is a device copy) against the unsharded forward of the same batch.  Development aid.

  python tools/shard_bench.py [--n 256] [--world 4] [--batch 8] [--views 2] [--conv-dtype fp32|bf16]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from liftreg_amd import parallel as par  # noqa: E402
from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model  # noqa: E402
from liftreg_amd.utils.sdct_projection_utils import scan_poses  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--conv-dtype", default="fp32")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--exchange", action="store_true", help="blocks 0/1 of the general / bf16 path with the round-3 halo exchange (halo_free01 = False)")
    ap.add_argument("--max-overhead", type=float, default=None, help="exit non-zero when sum_of_slabs / unsharded - 1 exceeds this")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, P, B = a.n, a.views, a.batch
    torch.manual_seed(1)
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": 56, "pca_path": "synthetic:1", "conv_dtype": a.conv_dtype}).to(dev).eval()
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    poses = scan_poses(30, P, n).astype(np.float32)
    inp = {"source": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
           "target": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
           "target_proj": torch.rand((B, P, n, n), generator=g, device=dev) * 2 - 1,
           "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}
    sh = par.SlabShardedRegistration(net, par.LocalComm(a.world))
    if a.exchange:
        sh.halo_free01 = False

    def timeit(fn):
        with torch.no_grad():
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iters):
                fn()
            e.record()
            torch.cuda.synchronize()
        return s.elapsed_time(e) / a.iters

    from liftreg_amd import ops

    def table(fn):
        with torch.no_grad(), ops.kernel_timer() as kt:
            fn()
            torch.cuda.synchronize()
        out = {}
        for name, rec in kt.summary().items():
            key = name.split("_s")[0] if name.startswith("conv3d") else name
            out[key] = round(out.get(key, 0.0) + float(np.sum(rec["ms"])), 3)
        return out

    from liftreg_amd.layers.losses import NCCLoss
    sim = NCCLoss(check_nan=False)

    def full():          # the unsharded step INCLUDING the similarity (the sharded forward computes it from all-reduced moments)
        out = net(inp)
        return sim(out["warped"], out["target"])

    # the decomposition must not change a bit: every rank's slab of the outputs equals the rows of the unsharded forward
    with torch.no_grad():
        ref = net(inp)
        ref_loss = float(sim(ref["warped"], ref["target"]))
        for r, o in enumerate(sh.forward([inp] * a.world)):
            d0, d1 = par.slab_bounds(n, a.world, r)
            assert torch.equal(o["pca_coefs"], ref["pca_coefs"]), f"rank {r}: coefficients differ"
            for k in ("params", "phi", "warped"):
                assert torch.equal(o[k], ref[k][:, :, d0:d1]), f"rank {r}: {k} differs"
            assert abs(float(o["sim_loss"]) - ref_loss) < 2e-7, f"rank {r}: loss differs"
        del ref
    if os.environ.get("SHARD_BENCH_TABLE"):
        print("unsharded", json.dumps(table(full)))
        print("sharded  ", json.dumps(table(lambda: sh.forward([inp] * a.world))))
    t_full = timeit(full)
    t_shard = timeit(lambda: sh.forward([inp] * a.world))
    # the library kernels' own time (event-bracketed one by one): what the decomposition costs the GPUs, without the host — one
    # Python process issues every virtual rank's launches here, and at 8 ranks that, not the GPU, sets the wall time above
    k_full = sum(table(full).values())
    k_shard = sum(table(lambda: sh.forward([inp] * a.world)).values())
    print(json.dumps({"n": n, "views": P, "batch": B, "world": a.world, "conv_dtype": a.conv_dtype,
                      "unsharded_ms": round(t_full, 3), "sum_of_slabs_ms": round(t_shard, 3),
                      "decomposition_overhead": round(t_shard / t_full - 1, 3),
                      "kernels_unsharded_ms": round(k_full, 3), "kernels_slabs_ms": round(k_shard, 3),
                      "kernel_overhead": round(k_shard / k_full - 1, 3), "slabs_equal_unsharded": True}))
    if a.max_overhead is not None and t_shard / t_full - 1 > a.max_overhead:
        sys.exit(f"decomposition overhead {t_shard / t_full - 1:.3f} above --max-overhead {a.max_overhead}")


if __name__ == "__main__":
    main()
