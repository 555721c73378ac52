"""CPU, world_size 2, gloo: the N>1 host logic of liftreg_amd.parallel — slab bounds, the partial-DRR
all-reduce, the NCC-moment all-reduce, and the collective-free slabs — reproduces the unsharded
result.  There is no GPU here, so the test (and only the test) injects an oracle-backed stand-in for
`liftreg_amd.parallel.ops`; the product default is the HIP ops and has no CPU path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleOps:
    """Same call signatures as liftreg_amd.ops for the functions parallel.py uses, on CPU tensors."""

    def __init__(self):
        from oracle import c_oracle as co
        self.co = co

    def drr_forward(self, vol, poses, resolution, spacing, *, d0=0, d1=None, full_D=None, **kw):
        out = self.co.drr_forward(vol.numpy(), np.asarray(poses, np.float32), np.asarray(spacing, np.float32),
                                  resolution, d0=d0, d1=d1, full_D=full_D)
        return torch.from_numpy(out)

    def backproject(self, proj, poses, img_shape, *, d0=0, d1=None, **kw):
        return torch.from_numpy(self.co.backproject(proj.numpy(), np.asarray(poses, np.float32), img_shape, d0=d0, d1=d1))

    def pca_reconstruct(self, coefs, basis, mean):
        return torch.from_numpy(self.co.pca_reconstruct(coefs.numpy(), basis.numpy(), mean.numpy()))

    def warp(self, img, disp, ids, seg, *, d0=0, d1=None, **kw):
        phi, w = self.co.warp(img.numpy(), disp.numpy(), ids=[t.numpy() for t in ids],
                              seg=None if seg is None else seg.numpy(), d0=d0, d1=d1)
        return torch.from_numpy(phi), torch.from_numpy(w)

    def ncc_moments(self, x, y, rows):
        return torch.from_numpy(self.co.ncc_moments(x.numpy(), y.numpy(), rows))

    def ncc_loss_from_moments(self, m, n_total, n_batch, variant=0):
        m = m.numpy()
        n = float(n_total)
        mx, my = m[:, 0] / n, m[:, 1] / n
        cov, vx, vy = m[:, 2] / n - mx * my, m[:, 3] / n - mx * mx, m[:, 4] / n - my * my
        rows = (cov + 1e-20) / np.sqrt((vx + 1e-20) * (vy + 1e-20)) if variant == 0 else cov ** 2 / (vx * vy + 1e-12)
        return torch.tensor(1.0 - rows.mean(), dtype=torch.float32), torch.from_numpy(rows.astype(np.float32))


def _worker(rank, world_size, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from liftreg_amd import parallel as par
        from oracle import c_oracle as co
        from oracle import ref_ops as ro
        par.ops = OracleOps()  # test-only injection
        rs = np.random.RandomState(11)
        D, W, H, P, R, B, L = 13, 10, 12, 3, 14, 2, 5
        d0, d1 = par.slab_bounds(D, world_size, rank)
        assert par.world() == (rank, world_size)
        mu = rs.uniform(0, 0.4, (D, W, H)).astype(np.float32)
        poses = ro.scan_poses(30, P, W).astype(np.float32)
        sp = np.array((2.2, 2.2, 2.2), np.float32)
        # DRR: partial sums over slabs all-reduce to the unsharded projection
        full = co.drr_forward(mu, poses, sp, (R, R))
        got = par.drr_forward_sharded(torch.from_numpy(mu[d0:d1].copy()), poses, (R, R), sp, D, d0, d1)
        np.testing.assert_allclose(got.numpy(), full, rtol=1e-5, atol=1e-6)
        # backprojection, PCA, warp: slabs need no collective and equal rows of the unsharded result
        proj = rs.uniform(-1, 1, (B, P, R, R)).astype(np.float32)
        tv = co.backproject(proj, poses, (D, W, H))
        assert np.array_equal(par.backproject_slab(torch.from_numpy(proj), poses, (D, W, H), d0, d1).numpy(), tv[:, :, d0:d1])
        coefs = rs.normal(0, 1, (B, L)).astype(np.float32)
        basis = rs.normal(0, 0.05, (L, 3 * D * W * H)).astype(np.float32)
        mean = rs.normal(0, 0.01, 3 * D * W * H).astype(np.float32)
        disp = co.pca_reconstruct(coefs, basis, mean).reshape(B, 3, D, W, H)
        dslab = par.pca_reconstruct_slab(torch.from_numpy(coefs), torch.from_numpy(basis), torch.from_numpy(mean), (D, W, H), d0, d1)
        assert np.array_equal(dslab.numpy(), disp[:, :, d0:d1])
        img = rs.uniform(-1, 1, (B, 1, D, W, H)).astype(np.float32)
        tabs = [torch.from_numpy(t) for t in ro.identity_axis_tables((D, W, H))]
        phi, warped = co.warp(img, disp, ids=[t.numpy() for t in tabs])
        sphi, sw = par.warp_slab(torch.from_numpy(img), dslab.contiguous(), tabs, d0, d1)
        assert np.array_equal(sw.numpy(), warped[:, :, d0:d1]) and np.array_equal(sphi.numpy(), phi[:, :, d0:d1])
        # NCC: five moments per sample all-reduce; loss equals the unsharded loss
        tgt = rs.uniform(-1, 1, (B, 1, D, W, H)).astype(np.float32)
        want, _ = co.ncc_loss(warped, tgt, 0)
        loss = par.ncc_loss_sharded(sw.contiguous(), torch.from_numpy(tgt[:, :, d0:d1].copy()), D * W * H)
        assert abs(float(loss) - want) < 2e-6
        # replicas: every item owned exactly once
        owned = torch.zeros(11)
        owned[par.shard_items(11, world_size, rank)] = 1
        dist.all_reduce(owned)
        assert bool((owned == 1).all())
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_slab_bounds_partition():
    from liftreg_amd.parallel import slab_bounds, shard_items
    for D in (1, 7, 13, 256):
        for ws in (1, 2, 3, 4, 8):
            b = [slab_bounds(D, ws, r) for r in range(ws)]
            assert b[0][0] == 0 and b[-1][1] == D
            assert all(b[i][1] == b[i + 1][0] for i in range(ws - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1
    assert sorted(sum((shard_items(10, 4, r) for r in range(4)), [])) == list(range(10))


@pytest.mark.timeout(300)
def test_two_rank_sharding_matches_unsharded():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
