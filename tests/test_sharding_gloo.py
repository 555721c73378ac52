"""CPU, world_size 2, gloo: the N>1 host logic of liftreg_amd.parallel — slab bounds, the partial-DRR
all-reduce, the NCC-moment all-reduce, the collective-free slabs, and the slab-sharded forward of the
whole model (halo send/recv, feature all-gather) — reproduces the unsharded result.  There is no GPU here,
so the tests (and only the tests) inject an oracle-backed stand-in for the `ops` module the product
code calls; the product default is the HIP ops and has no CPU path."""
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
    """Same call signatures as liftreg_amd.ops for what parallel.py / the model use, on CPU tensors."""
    LAYOUT_NCDHW, LAYOUT_NDHWC, LAYOUT_NDHWC_HPS = 0, 1, 2
    LAYOUT_BF16_NDHWC, LAYOUT_BF16_NDHWC_HPS = 3, 4

    def __init__(self):
        from oracle import c_oracle as co
        from oracle import ref_ops as ro
        self.co, self.ro = co, ro

    @staticmethod
    def pca_warp_supported(*_):
        return False      # the shim keeps the two-kernel decode (pca_reconstruct + warp)

    @staticmethod
    def conv3d_first_split_supported(*_):
        return False      # … and the concatenated encoder input

    @staticmethod
    def conv3d_pair01_supported(*_, **__):
        return False      # … and one kernel per encoder block

    conv3d_pair01_shapes_supported = conv3d_pair01_supported

    @staticmethod
    def encoder_input_bf16_supported(*_):
        return False      # … and the fp32 feature volume

    # ---- layouts (mirror of LR_LAYOUT_*) -------------------------------------------------------
    def _to_ncdhw(self, x, layout):
        if layout == self.LAYOUT_NCDHW:
            return x
        if layout == self.LAYOUT_NDHWC_HPS:
            x = self.hps_to_ndhwc(x)
        return x.permute(0, 4, 1, 2, 3).contiguous()

    def _from_ncdhw(self, y, layout):
        if layout == self.LAYOUT_NCDHW:
            return y.contiguous()
        y = y.permute(0, 2, 3, 4, 1).contiguous()
        if layout == self.LAYOUT_NDHWC_HPS:
            B, D, W, H, C = y.shape
            h = torch.arange(H)
            inv = torch.empty(H, dtype=torch.long)
            inv[(h & 1) * (H // 2) + (h >> 1)] = h
            y = y.reshape(B, D, W, H, C // 16, 16)[:, :, :, inv].permute(0, 1, 2, 4, 3, 5).reshape(B, D, W, H, C)
        return y

    def hps_to_ndhwc(self, y):
        B, D, W, H, C = y.shape
        h = torch.arange(H)
        rows = y.reshape(B, D, W, C // 16, H, 16)[:, :, :, :, (h & 1) * (H // 2) + (h >> 1)]
        return rows.permute(0, 1, 2, 4, 3, 5).reshape(B, D, W, H, C)

    # ---- ops -------------------------------------------------------------------------------------
    def conv3d_pack_weights(self, weight, in_layout):
        return None

    def conv3d_mask_supported(self, *a, **k):
        return False           # the shim writes no LeakyReLU sign masks

    def conv3d_k3_lrelu(self, x, weight, bias, stride, *, in_layout=0, out_layout=0, negative_slope=0.2, packed=None,
                        out=None, z_phase=0):
        y = self.ro.conv_block(self._to_ncdhw(x, in_layout), weight.detach(), bias.detach(), stride, negative_slope)
        y = self._from_ncdhw(y, out_layout)
        if out is None:
            return y
        out.copy_(y)                       # the sharded forward writes into plane ranges of its halo-padded buffers
        return out

    def linear_lrelu(self, x, weight, bias, negative_slope=1.0):
        y = torch.nn.functional.linear(x, weight.detach(), bias.detach())
        return y if negative_slope == 1.0 else torch.nn.functional.leaky_relu(y, negative_slope)

    def drr_forward(self, vol, poses, resolution, spacing, *, d0=0, d1=None, full_D=None, **kw):
        out = self.co.drr_forward(vol.numpy(), np.asarray(poses, np.float32), np.asarray(spacing, np.float32),
                                  resolution, d0=d0, d1=d1, full_D=full_D)
        return torch.from_numpy(out)

    def backproject(self, proj, poses, img_shape, *, d0=0, d1=None, out=None, **kw):
        res = torch.from_numpy(self.co.backproject(proj.numpy(), np.asarray(poses, np.float32), img_shape, d0=d0, d1=d1))
        if out is not None:      # the model writes straight into channels 1..P of the encoder input
            out.copy_(res)
            return out
        return res

    def pca_warp_supported(self, *a, **k):
        return False           # the shim has no one-pass decode: the sharded forward takes its two-op path

    def pca_reconstruct(self, coefs, basis, mean):
        return torch.from_numpy(self.co.pca_reconstruct(coefs.numpy(), basis.numpy(), mean.numpy()))

    def warp(self, img, disp, ids, seg, *, d0=0, d1=None, **kw):
        phi, w = self.co.warp(img.numpy(), disp.numpy(), ids=[t.numpy() for t in ids],
                              seg=None if seg is None else seg.numpy(), d0=d0, d1=d1)
        return torch.from_numpy(phi), torch.from_numpy(w)

    def mask_compose(self, img, seg):
        return torch.from_numpy(self.co.mask_compose(img.numpy(), seg.numpy()))

    def ncc_moments(self, x, y, rows):
        return torch.from_numpy(self.co.ncc_moments(x.numpy(), y.numpy(), rows))

    def ncc_loss_from_moments(self, m, n_total, n_batch, variant=0):
        m = m.numpy()
        n = float(n_total)
        mx, my = m[:, 0] / n, m[:, 1] / n
        cov, vx, vy = m[:, 2] / n - mx * my, m[:, 3] / n - mx * mx, m[:, 4] / n - my * my
        rows = (cov + 1e-20) / np.sqrt((vx + 1e-20) * (vy + 1e-20)) if variant == 0 else cov ** 2 / (vx * vy + 1e-12)
        return torch.tensor(1.0 - rows.mean(), dtype=torch.float32), torch.from_numpy(rows.astype(np.float32))


def _inject(shim):
    """Test-only: route every `ops.` call of the host-side modules to the oracle shim."""
    import liftreg_amd.parallel as par
    import liftreg_amd.models.LiftRegDeformSubspaceBackproj as mod
    import liftreg_amd.layers.layers as lay
    import liftreg_amd.autograd as ag
    par.ops = mod.ops = lay.ops = ag.ops = shim
    return par, mod


def _worker(rank, world_size, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from oracle import c_oracle as co
        from oracle import ref_ops as ro
        shim = OracleOps()
        par, mod = _inject(shim)
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

        # ---- the whole model, slab-sharded over the two ranks (halo isend/irecv, all_gather, all_reduce)
        n = 64
        torch.manual_seed(5)
        net = mod.model([n, n, n], {"drr_feature_num": 2, "latent_dim": 4, "pca_path": "synthetic:3"}).eval()
        g = torch.Generator().manual_seed(5)
        net.set_pca(torch.randn((4, 3 * n ** 3), generator=g) * 0.01, torch.zeros(3 * n ** 3))
        inp = {"source": torch.rand((1, 1, n, n, n), generator=g) * 2 - 1,
               "target": torch.rand((1, 1, n, n, n), generator=g) * 2 - 1,
               "target_proj": torch.rand((1, 2, n, n), generator=g) * 2 - 1,
               "target_poses": torch.from_numpy(ro.scan_poses(30, 2, n).astype(np.float32))[None]}
        with torch.no_grad():
            ref = net(inp)                                                     # unsharded, through the same shim
            out = par.SlabShardedRegistration(net, par.DistComm()).forward([inp])[0]
        s0, s1 = par.slab_bounds(n, world_size, rank)
        np.testing.assert_allclose(out["pca_coefs"].numpy(), ref["pca_coefs"].numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out["params"].numpy(), ref["params"][:, :, s0:s1].numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out["warped"].numpy(), ref["warped"][:, :, s0:s1].numpy(), rtol=1e-4, atol=1e-5)
        want_loss = ro.ncc_loss(ref["warped"], ref["target"])
        assert abs(float(out["sim_loss"]) - float(want_loss)) < 1e-5
        # the exchange form of blocks 0 / 1 (the rank below's top plane over isend / irecv instead of the recomputed halo plane)
        with torch.no_grad():
            sh = par.SlabShardedRegistration(net, par.DistComm())
            sh.halo_free01 = False
            out2 = sh.forward([inp])[0]
        for k in ("pca_coefs", "params", "warped"):
            assert torch.equal(out2[k], out[k]), k
        # the tail behind the gather depth: sharded by SAMPLE (all_to_all_single + one small all-gather of the coefficients; with
        # B = 1 rank 1 owns no sample) == the replicated tail of round 5 (all-gather of the activation, FC1 sharded by neurons)
        with torch.no_grad():
            sh = par.SlabShardedRegistration(net, par.DistComm())
            sh.sample_sharded_tail = False
            out3 = sh.forward([inp])[0]
            inp3 = {k: (torch.cat([v, v.flip(-1), v * 0.5], 0) if k != "target_poses" else v.repeat(3, 1, 1)) for k, v in inp.items()}   # B = 3: 2 + 1 samples
            ref3 = net(inp3)
            out4 = par.SlabShardedRegistration(net, par.DistComm()).forward([inp3])[0]
        for k in ("pca_coefs", "params", "warped"):
            assert torch.equal(out3[k], out[k]), k
        np.testing.assert_allclose(out4["pca_coefs"].numpy(), ref3["pca_coefs"].numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out4["warped"].numpy(), ref3["warped"][:, :, s0:s1].numpy(), rtol=1e-4, atol=1e-5)
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()[-600:]))
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


@pytest.mark.timeout(600)
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
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


# ------------------------------------------------------------------------------- data-parallel training collective
def _ddp_worker(rank, world_size, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from liftreg_amd.parallel import GradientAllReduce

        class Net(torch.nn.Module):                      # same naming split as the model: encoders.6.* = FC head
            def __init__(self):
                super().__init__()
                self.encoders = torch.nn.ModuleList([torch.nn.Linear(6, 6) for _ in range(6)] +
                                                    [torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))])

            def forward(self, x):
                for m in self.encoders:
                    x = m(x)
                return x

        torch.manual_seed(3)                              # identical replicas
        net = Net()
        ddp = GradientAllReduce(net)
        assert len(ddp.buckets) == 2 and ddp.nbytes() == 4 * sum(p.numel() for p in net.parameters())
        xs = [torch.randn(4, 6, generator=torch.Generator().manual_seed(100 + r)) for r in range(world_size)]
        for step in range(2):                             # second step checks zero_grad / hook re-arming
            ddp.zero_grad()
            net(xs[rank]).square().sum().backward()
            ddp.finish()
            ref = Net()
            ref.load_state_dict(net.state_dict())
            tot = sum(ref(x).square().sum() for x in xs) / world_size
            tot.backward()
            for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
                assert torch.allclose(p.grad, pr.grad, rtol=1e-5, atol=1e-6), n
                assert any(p.grad.data_ptr() >= b["flat"].data_ptr() and
                           p.grad.data_ptr() < b["flat"].data_ptr() + 4 * b["flat"].numel() for b in ddp.buckets)
            torch.optim.SGD(net.parameters(), lr=0.01).step()
        ddp.remove()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, repr(e) + traceback.format_exc()[-1500:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=250) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
