"""Probe (development aid): the encode and the decode halves of a C3 batch on CU-masked HIP streams
(hipExtStreamCreateWithCUMask) — each leg alone on n CUs, then both concurrently on complementary masks."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from liftreg_amd import _hip  # noqa: E402
from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model  # noqa: E402
from liftreg_amd.pipeline import _masked_stream  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = bench.CONFIGS["c3"]
    n, P, L = cfg["n"], cfg["P"], cfg["L"]
    torch.manual_seed(2021)
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:2021"}).to(dev).eval()
    inp = bench.synth_inputs(cfg, dev)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    with torch.no_grad():
        coefs = net.encode(inp["source"], inp["target_proj"], inp["target_poses"])
        net.decode(inp["source"], coefs, None)
        torch.cuda.synchronize()

        def enc():
            return net.encode(inp["source"], inp["target_proj"], inp["target_poses"])

        def dec():
            return net.decode(inp["source"], coefs, None)

        def timed(stream, fn, reps=5):
            with torch.cuda.stream(stream):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    fn()
                e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        plain = torch.cuda.Stream()
        print(f"plain stream: encode {timed(plain, enc):.3f} ms, decode {timed(plain, dec):.3f} ms", flush=True)
        for r in (256, 128, 64, 48, 32):
            s = _masked_stream(dev, range(0, r), ncu)
            print(f"decode on CUs [0,{r}): {timed(s, dec):.3f} ms", flush=True)
        for r in (0, 32, 48, 64):
            s = _masked_stream(dev, range(r, ncu), ncu)
            os.environ["LIFTREG_PAIR01_BLOCKS"] = str(ncu - r)
            _hip.reload_switches()
            print(f"encode on CUs [{r},{ncu}) with {ncu - r} pair blocks: {timed(s, enc):.3f} ms", flush=True)
        for r in (32, 48, 64):
            sd = _masked_stream(dev, range(0, r), ncu)
            se = _masked_stream(dev, range(r, ncu), ncu)
            os.environ["LIFTREG_PAIR01_BLOCKS"] = str(ncu - r)
            _hip.reload_switches()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream()
            e0.record(cur)
            sd.wait_stream(cur); se.wait_stream(cur)
            for _ in range(5):
                with torch.cuda.stream(se):
                    enc()
                with torch.cuda.stream(sd):
                    dec()
            cur.wait_stream(sd); cur.wait_stream(se)
            e1.record(cur)
            torch.cuda.synchronize()
            print(f"both, {r} decode CUs: {e0.elapsed_time(e1) / 5:.3f} ms per (encode || decode)", flush=True)


if __name__ == "__main__":
    main()
