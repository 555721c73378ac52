"""Development aid: the ShadowRegistrar step against the plain step, with allocator statistics and variants."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from liftreg_amd.layers.losses import NCCLoss  # noqa: E402
from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model  # noqa: E402
from liftreg_amd.pipeline import ShadowRegistrar  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = bench.CONFIGS["c3"]
    n, P, L = cfg["n"], cfg["P"], cfg["L"]
    torch.manual_seed(2021)
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:2021"}).to(dev).eval()
    inp = bench.synth_inputs(cfg, dev)
    sim = NCCLoss(check_nan=False)
    reg = ShadowRegistrar(net, sim, head_start=int(os.environ.get('HEAD_START', '200000')))

    def plain():
        out = net(inp)
        return sim(out["warped"], out["target"])

    def shadow():
        return reg.submit(inp)[1]

    def run(fn, steps=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s0 = torch.cuda.memory_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        s1 = torch.cuda.memory_stats()
        return (t / steps * 1e3, t_host / steps * 1e3, s1["segment.all.allocated"] - s0["segment.all.allocated"],
                s1["num_alloc_retries"] - s0["num_alloc_retries"], torch.cuda.memory_reserved() / 2 ** 30, float(loss))

    with torch.no_grad():
        t_end = time.perf_counter() + 2.0
        while time.perf_counter() < t_end:
            plain()
            torch.cuda.synchronize()
        from liftreg_amd import ops
        for name, fn in (("plain", plain), ("shadow", shadow)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            with ops.kernel_timer() as kt:
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
            print(name, {k: round(sum(v["ms"]) / len(v["ms"]), 3) for k, v in kt.summary().items() if sum(v["ms"]) / len(v["ms"]) > 0.1}, flush=True)
        for rep in range(2):
            for name, fn in (("plain", plain), ("shadow", shadow)):
                ms, host, seg, retry, res, loss = run(fn)
                print(f"{name}: {ms:.3f} ms/step (host enqueue {host:.3f} ms/step), new segments {seg}, retries {retry}, reserved {res:.1f} GiB, loss {loss:.6f}", flush=True)


if __name__ == "__main__":
    main()
