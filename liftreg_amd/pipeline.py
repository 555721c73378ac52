"""Two-stream software pipeline over registration batches (throughput serving / evaluation loops).

The path has two halves with opposite bottlenecks:
  encode  backprojection → 6 conv blocks → FC head            fp32-MFMA-bound   (≈9.6 ms at C3)
  decode  PCA reconstruction → identity add + warp → NCC      HBM-bound         (≈3.7 ms at C3)
Registrations are independent, so batch i's decode can run on one HIP stream while batch i+1's
encode runs on another: the HBM-bound kernels fill the memory system while the matrix pipe is busy.
Results are identical to `model.forward` + `NCCLoss` (same kernels, same order within a batch).
"""
import torch


class TwoStreamRegistrar:
    """`submit(batch)` enqueues one batch; outputs are valid after `synchronize()` (or an event wait)."""

    def __init__(self, net, sim=None):
        self.net = net
        self.sim = sim
        dev = next(net.parameters()).device
        self.enc = torch.cuda.Stream(device=dev)
        self.dec = torch.cuda.Stream(device=dev)
        self._hold = []  # keeps the previous batch's cross-stream tensors alive while the GPU still uses them

    def submit(self, batch):
        net = self.net
        moving, target = batch["source"], batch["target"]
        seg = batch.get("source_label") if isinstance(batch, dict) else None
        cur = torch.cuda.current_stream()
        self.enc.wait_stream(cur)  # inputs produced on the caller's stream
        self.dec.wait_stream(cur)
        with torch.cuda.stream(self.enc):
            coefs = net.encode(moving, batch["target_proj"], batch["target_poses"])
            done = torch.cuda.Event()
            done.record(self.enc)
        coefs.record_stream(self.dec)
        with torch.cuda.stream(self.dec):
            self.dec.wait_event(done)
            disp, phi, warped = net.decode(moving, coefs, seg)
            from . import ops
            target_cp = ops.mask_compose(target, batch["target_label"]) if seg is not None else target
            loss = self.sim(warped, target_cp) if self.sim is not None else None
        out = {"warped": warped, "phi": phi, "params": disp, "target": target_cp, "pca_coefs": coefs,
               "target_proj": batch["target_proj"], "warped_proj": batch["target_proj"]}
        self._hold = [self._hold[-1] if self._hold else None, (out, loss)][-2:]
        return out, loss

    def synchronize(self):
        self.enc.synchronize()
        self.dec.synchronize()
        self._hold = []
