"""Serving-loop helpers over registration batches: the whole step captured once into a hipGraph and replayed.

(Rounds 2-4 also carried two-stream / shadow-stream pipelines that overlapped the HBM-bound kernels with the MFMA-bound pair
kernel: on this power-limited part they measured -3 ... +2 %, inside the box-to-box spread — profiles/NOTES_r04.md — and were
removed in round 5 together with their register-light kernels and the CU-masked streams.)
"""
import torch


class GraphedRegistrar:
    """The whole forward (+ similarity) of one fixed-shape batch captured once in a HIP graph and replayed.

    At the small configurations (C1 64³/B=1, C2 128³/B=4) a forward is ≈25 launches of tens of microseconds each
    and the step is launch-bound; replaying one graph removes the per-launch host cost.  Every kernel on the path
    launches on torch's current stream and allocates only through torch's (graph-aware) caching allocator, and
    the geometry travels by value in the kernel arguments, so the capture needs no special casing.

        reg = GraphedRegistrar(net, example_batch, sim=NCCLoss(check_nan=False))
        out, loss = reg(batch)        # copies the batch into the static inputs, replays, returns the static outputs

    Outputs are the graph's static tensors (overwritten by the next call); the emitter geometry is the one of the
    example batch (the model caches it the same way, …Backproj.py:85-87).  Inference only (no autograd graph).
    """

    _KEYS = ("source", "target", "target_proj", "source_label", "target_label")

    def __init__(self, net, example, sim=None, warmup=2):
        self.net, self.sim = net, sim
        self.static_in = {k: example[k].clone() for k in self._KEYS if k in example}
        self.static_in["target_poses"] = example["target_poses"]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():      # warm-up outside the capture: PCA basis, packed weights
            for _ in range(warmup):
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = self._run()

    def _run(self):
        out = self.net(self.static_in)
        if self.sim is None:
            return out, None
        mom = out.get("ncc_moments")
        return out, (self.sim(out["warped"], out["target"], moments=mom) if mom is not None else self.sim(out["warped"], out["target"]))

    def __call__(self, batch):
        for k, t in self.static_in.items():
            if k != "target_poses":
                t.copy_(batch[k], non_blocking=True)
        self.graph.replay()
        return self.static_out
