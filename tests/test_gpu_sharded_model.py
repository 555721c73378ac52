"""GPU: the z-slab sharded forward of the whole model (liftreg_amd.parallel.SlabShardedRegistration) run as
2 and 4 virtual ranks on one GPU (LocalComm) reproduces the unsharded model: every rank's slab of the
displacement field, phi and the warped image equals the corresponding rows, the PCA coefficients are
replicated, and the NCC from all-reduced moments equals the unsharded NCC."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,world,conv_dtype", [(64, 2, "fp32"), (128, 4, "fp32"), (128, 2, "fp32"), (128, 4, "bf16"),
                                                # the metric's largest world (BASELINE configs[2..4] at 8 ranks): C3's slabs are 32 planes
                                                # = the gather depth, the pair kernel's front step runs on 7 of the 8 ranks
                                                (256, 8, "fp32"), (256, 8, "bf16"), (384, 8, "bf16-c5"),
                                                # four views (the reference's shipped drr_feature_num): the five-channel pair kernel on slabs
                                                (160, 2, "fp32-p4"), (128, 4, "fp32-p4"),
                                                # six views in fp32: the general first block (no pair kernel), its halo plane recomputed;
                                                # and the exchange form of blocks 0 / 1 (`halo_free01 = False`)
                                                (64, 2, "fp32-p6"), (128, 4, "bf16-exchange"), (64, 2, "fp32-p6-exchange"),
                                                # round 6: the tail behind the gather depth is sharded by SAMPLE (all-to-all); the
                                                # replicated tail of round 5 (all-gather + FC1 by neurons) stays selectable; B = 3 on 4
                                                # ranks leaves one rank without a sample
                                                (256, 8, "fp32-reptail"), (128, 4, "bf16-reptail"), (128, 4, "fp32-b3")])
def test_slab_sharded_forward_equals_unsharded(n, world, conv_dtype):
    from liftreg_amd import parallel as par
    from liftreg_amd.layers.losses import NCCLoss
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
    from liftreg_amd.utils.sdct_projection_utils import scan_poses
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    P, L, B, R = 2, 12, 2, n
    exchange = conv_dtype.endswith("-exchange")
    conv_dtype = conv_dtype[:-len("-exchange")] if exchange else conv_dtype
    reptail = conv_dtype.endswith("-reptail")
    conv_dtype = conv_dtype[:-len("-reptail")] if reptail else conv_dtype
    if conv_dtype == "fp32-b3":
        conv_dtype, B = "fp32", 3
    if conv_dtype == "fp32-p6":
        conv_dtype, P = "fp32", 6
    if conv_dtype == "bf16":
        P = 11      # C4: 11-view limited-angle DRR, bf16 convs, 4-way z-slab sharding (bf16 halo planes on the wire)
    if conv_dtype == "fp32-p4":
        conv_dtype, P, R = "fp32", 4, int(1.5 * n)
    if conv_dtype == "bf16-c5":
        conv_dtype, B, R = "bf16", 4, 512      # C5's shapes: 384^3, 2 x 512^2 views, 4 registrations per GPU, bf16 convs
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:9",
                            "conv_dtype": conv_dtype}).to(dev).eval()
    poses = scan_poses(30, P, n).astype(np.float32)
    inp = {"source": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
           "target": torch.rand((B, 1, n, n, n), generator=g, device=dev) * 2 - 1,
           "target_proj": torch.rand((B, P, R, R), generator=g, device=dev) * 2 - 1,
           "target_poses": torch.from_numpy(np.broadcast_to(poses, (B, P, 3)).copy())}
    with torch.no_grad():
        ref = net(inp)
        ref_loss = NCCLoss()(ref["warped"], ref["target"])
        sharded = par.SlabShardedRegistration(net, par.LocalComm(world))
        if exchange:
            sharded.halo_free01 = False
        if reptail:
            sharded.sample_sharded_tail = False
        outs = sharded.forward([inp] * world)
    assert len(outs) == world
    for r, out in enumerate(outs):
        d0, d1 = par.slab_bounds(n, world, r)
        assert torch.equal(out["pca_coefs"], ref["pca_coefs"]), f"rank {r}: coefficients differ"
        assert torch.equal(out["params"], ref["params"][:, :, d0:d1])
        assert torch.equal(out["phi"], ref["phi"][:, :, d0:d1])
        assert torch.equal(out["warped"], ref["warped"][:, :, d0:d1])
        assert abs(float(out["sim_loss"]) - float(ref_loss)) < 2e-7        # (fp64 moments summed in another order)
    with pytest.raises(ValueError):
        par.SlabShardedRegistration(net, par.LocalComm(5 if n == 384 else 3))     # planes per rank: a multiple of slab_multiple(net)


@pytest.mark.parametrize("procs,n,conv_dtype", [(2, 128, "fp32"), (4, 128, "fp32"), (2, 128, "bf16")])
def test_slab_sharding_across_real_processes_on_one_gpu(procs, n, conv_dtype):
    """The sharded forward in SEPARATE processes joined by torch.distributed (parallel.DistComm, not the one-process LocalComm
    of the tests above): tools/shard_bench.py --procs N starts N fresh children that all sit on cuda:0 (RCCL refuses several
    ranks on one device — 'Duplicate GPU detected' — so the group is gloo and DistComm stages through the host; over RCCL the
    same calls carry device pointers).  Every rank asserts its slab == the rows of its own unsharded forward, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    views = "11" if conv_dtype == "bf16" else "2"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "shard_bench.py"), "--procs", str(procs), "--n", str(n), "--batch", "2",
                        "--views", views, "--conv-dtype", conv_dtype, "--backend", "gloo", "--iters", "2"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    summary = rows[-1]
    assert summary["procs"] == procs and summary["backend"] == "gloo" and summary["slabs_equal_unsharded"] is True
    ranks = [x for x in rows if "rank" in x]
    assert sorted(x["rank"] for x in ranks) == list(range(procs)) and all(x["slab_equals_unsharded"] for x in ranks)
    assert [x["rows"] for x in sorted(ranks, key=lambda x: x["rank"])] == [[i * n // procs, (i + 1) * n // procs] for i in range(procs)]
