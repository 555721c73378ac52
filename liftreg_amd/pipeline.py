"""Two-stream software pipeline over registration batches (throughput serving / evaluation loops).

The path has two halves with opposite bottlenecks:
  encode  backprojection → 6 conv blocks → FC head            fp32-MFMA-bound   (≈9.6 ms at C3)
  decode  PCA reconstruction → identity add + warp → NCC      HBM-bound         (≈3.7 ms at C3)
Registrations are independent, so batch i's decode can run on one HIP stream while batch i+1's
encode runs on another: the HBM-bound kernels fill the memory system while the matrix pipe is busy.
Results are identical to `model.forward` + `NCCLoss` (same kernels, same order within a batch).
"""
import torch


def _masked_stream(dev, cus, ncu):
    """A HIP stream whose kernels run only on the compute units `cus` (lr_stream_create_cu_mask: hipExtStreamCreateWithCUMask
    through the runtime the library is bound to; bit i of the mask = CU i, and the driver deals consecutive bits round-robin over
    the XCDs, so a contiguous range is spread evenly over the 8 dies).  The stream lives as long as the process."""
    import ctypes
    from . import _hip
    nw = (ncu + 31) // 32
    words = [0] * nw
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    arr = (ctypes.c_uint32 * nw)(*words)
    st = ctypes.c_void_p()
    with torch.cuda.device(dev):
        _hip.check(_hip.lib().lr_stream_create_cu_mask(ctypes.cast(arr, ctypes.c_void_p), nw, ctypes.cast(ctypes.byref(st), ctypes.c_void_p)),
                   "lr_stream_create_cu_mask")
    return torch.cuda.ExternalStream(st.value, device=dev)


class TwoStreamRegistrar:
    """`submit(batch)` enqueues one batch; outputs are valid after `synchronize()` (or an event wait)."""

    def __init__(self, net, sim=None, decode_cus=0):
        self.net = net
        self.sim = sim
        dev = next(net.parameters()).device
        if decode_cus > 0:
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            self.dec = _masked_stream(dev, range(0, decode_cus), ncu)
            self.enc = _masked_stream(dev, range(decode_cus, ncu), ncu)
        else:
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
        # the batch's tensors were allocated on the caller's stream but are read by kernels on enc/dec: tell the
        # caching allocator, or a serving loop that drops the batch after submit() gets its blocks re-used (and
        # overwritten on the caller's stream) while the previous batch's warp / first conv block still reads them
        for k in ("source", "target", "target_proj", "source_label", "target_label"):
            t = batch.get(k) if isinstance(batch, dict) else None
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(self.enc)
                t.record_stream(self.dec)
        with torch.cuda.stream(self.enc):
            coefs = net.encode(moving, batch["target_proj"], batch["target_poses"])
            done = torch.cuda.Event()
            done.record(self.enc)
        coefs.record_stream(self.dec)
        with torch.cuda.stream(self.dec):
            self.dec.wait_event(done)
            from . import ops
            target_cp = ops.mask_compose(target, batch["target_label"]) if seg is not None else target
            disp, phi, warped, *mom = net.decode(moving, coefs, seg, target=target_cp if self.sim is not None else None)
            if self.sim is None:
                loss = None
            else:                       # moments from the decode's epilogue (opt key fuse_ncc) go to the similarity explicitly
                loss = self.sim(warped, target_cp, moments=mom[0]) if mom else self.sim(warped, target_cp)
        out = {"warped": warped, "phi": phi, "params": disp, "target": target_cp, "pca_coefs": coefs,
               "target_proj": batch["target_proj"], "warped_proj": batch["target_proj"]}
        # outputs were allocated on enc/dec and are handed to code running on the caller's stream
        for t in (warped, phi, disp, coefs, target_cp, loss):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        self._hold = [self._hold[-1] if self._hold else None, (batch, out, loss)][-2:]
        return out, loss

    def synchronize(self):
        self.enc.synchronize()
        self.dec.synchronize()
        self._hold = []


class ShadowRegistrar:
    """Pipeline over registration batches that hides the two small HBM-bound kernels of the fp32 path behind the fused
    pair kernel (csrc/conv01_fused.hip), which is bound by the matrix pipe / vector issue and moves < 1 TB/s for 5.5 ms:

        main stream    pair(i) → blocks 2..5 → FC → decode(i)          pair(i+1) → …
        side stream A                bp(i+1) ─────────┘ (waited for by pair(i+1))
        side stream B                                   ncc(i)  (after decode(i), beside pair(i+1))

    bp = backprojection of the NEXT batch's views (…Backproj.py:89-93), ncc = the similarity's moments pass of the PREVIOUS
    batch (layers/losses.py:14-29).  Measured beside the pair kernel both are free (tools/overlap_probe.py: 5.75 ms for
    pair ‖ backproject against 5.75 + 0.27 serial).  The decode itself cannot hide there (213 registers per lane do not fit
    beside the pair kernel's waves: profiles/NOTES_r04.md).  Every batch's kernels and their order are those of
    `model.forward` + `NCCLoss`: identical results.  bp(i+1) is released by an event at the start of batch i's encode, so the
    pipeline runs exactly one batch ahead (two backprojection buffers).

        reg = ShadowRegistrar(net, NCCLoss(check_nan=False))
        out, loss = reg.submit(batch)      # asynchronous; reg.synchronize() (or a stream/event wait) before reading
    """

    def __init__(self, net, sim=None, head_start=200000):
        self.net, self.sim = net, sim
        self.head_start = int(head_start)    # spin cycles in front of the side kernels (see submit)
        dev = next(net.parameters()).device
        self.main = torch.cuda.Stream(device=dev)
        self.side_bp = torch.cuda.Stream(device=dev)
        self.side_ncc = torch.cuda.Stream(device=dev)
        self._tv = [None, None]          # the two backprojection buffers
        self._n = 0
        self._enc_start = None           # event: the previous batch's encode has started on the main stream
        self._hold = []

    def submit(self, batch):
        net = self.net
        moving, target, proj = batch["source"], batch["target"], batch["target_proj"]
        seg = batch.get("source_label") if isinstance(batch, dict) else None
        cur = torch.cuda.current_stream()
        for st in (self.main, self.side_bp, self.side_ncc):
            st.wait_stream(cur)          # inputs produced on the caller's stream
        for k in ("source", "target", "target_proj", "source_label", "target_label"):
            t = batch.get(k) if isinstance(batch, dict) else None
            if isinstance(t, torch.Tensor) and t.is_cuda:
                for st in (self.main, self.side_bp, self.side_ncc):
                    t.record_stream(st)
        B, _, D, W, H = moving.shape
        P = proj.shape[1]
        slot = self._n & 1
        self._n += 1
        with torch.cuda.stream(self.side_bp):
            if self._enc_start is not None:
                self.side_bp.wait_event(self._enc_start)     # one batch ahead, not more (and buffer `slot` is free again:
                if self.head_start and hasattr(torch.cuda, "_sleep"):
                    torch.cuda._sleep(self.head_start)       # let the pair kernel's 256 blocks take their CUs first
            if self._tv[slot] is None or tuple(self._tv[slot].shape) != (B, P, D, W, H):   # its reader finished before that event)
                self._tv[slot] = torch.empty((B, P, D, W, H), dtype=torch.float32, device=moving.device)
            tv = net.backproject_views(proj, batch["target_poses"], (D, W, H), out=self._tv[slot], light=True)
            bp_done = torch.cuda.Event()
            bp_done.record(self.side_bp)
        tv.record_stream(self.main)
        with torch.cuda.stream(self.main):
            self._enc_start = torch.cuda.Event()
            self._enc_start.record(self.main)
            self.main.wait_event(bp_done)
            coefs = net.encode(moving, proj, batch["target_poses"], target_volume=tv)
            from . import ops
            target_cp = ops.mask_compose(target, batch["target_label"]) if seg is not None else target
            disp, phi, warped, *mom = net.decode(moving, coefs, seg, target=target_cp if self.sim is not None else None)
            dec_done = torch.cuda.Event()
            dec_done.record(self.main)
        loss = None
        if self.sim is not None:
            for t in (warped, target_cp):
                t.record_stream(self.side_ncc)
            with torch.cuda.stream(self.side_ncc):
                self.side_ncc.wait_event(dec_done)
                if self.head_start and hasattr(torch.cuda, "_sleep"):
                    torch.cuda._sleep(self.head_start)
                loss = self.sim(warped, target_cp, moments=mom[0]) if mom else self.sim(warped, target_cp)
        out = {"warped": warped, "phi": phi, "params": disp, "target": target_cp, "pca_coefs": coefs,
               "target_proj": proj, "warped_proj": proj}
        for t in (warped, phi, disp, coefs, target_cp, loss):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        self._hold = [self._hold[-1] if self._hold else None, (batch, out, loss)][-2:]
        return out, loss

    def synchronize(self):
        for st in (self.main, self.side_bp, self.side_ncc):
            st.synchronize()
        self._hold = []


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
