#!/usr/bin/env python3
"""bench.py — registrations/s of the LiftReg hot path on MI355X.

One "step" = one pass of the path over one batch of synthetic input, everything resident in HBM:
    backproject → conv×6 (fp32 MFMA) → FC×3 → PCA reconstruct → identity add + trilinear warp → NCC
on BASELINE.json's configs[2]: 256^3 CT, 2×256^2 DRR, batch 8, latent 56, fp32 ("c3").
Multi-GPU: registrations are independent → one replica per GPU, no data-path collective (weak
scaling); torch.distributed (RCCL) is used only for the barrier and the max-over-ranks time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1|c2|c3] [--no-cpu-baseline]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N … bench.py --gpus N …

Prints ONE JSON line (rank 0).  `roofline` describes the kernel that takes the most time in the
step, `roofline_backproject` the kernel BASELINE.json's metric names; both are measured live with
HIP events on the launch stream inside the timed region.  `cpu_baseline` times the torch-CPU oracle
(oracle/ref_ops.py — the reference's own ATen op sequence) on the host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {  # BASELINE.json configs (single-GPU ones)
    "c1": dict(n=64, P=2, R=64, B=1, L=56),
    "c2": dict(n=128, P=2, R=128, B=4, L=56),
    "c3": dict(n=256, P=2, R=256, B=8, L=56),
    "c4": dict(n=256, P=11, R=256, B=4, L=56),   # C4 per GPU in replica mode (batch 16 over 4 GPUs); use --conv-dtype bf16
    # the reference's OWN shipped configuration (/root/reference/cur_task_setting.json:30,56-57: drr_feature_num 4, latent_dim 56,
    # batch_size 30; …/models/LiftRegDeformSubspaceBackproj.py:36 hard-codes 160^3; detector int(1.5 * 160) = 240,
    # sdct_projection_utils.py:146-151) — not a BASELINE.json config: an extra line, never the headline
    "native160": dict(n=160, P=4, R=240, B=30, L=56),
}
SLAB_GLOBAL_BATCH = {"c4": 16}   # --shard slab: C4 is ONE batch of 16 sharded by z-slab over the ranks (BASELINE configs[3])
class SclkSampler:
    """Best-effort sample of the GPU's shader clock (sysfs pp_dpm_sclk of this rank's device, every 50 ms, from a host
    thread) while the timed steps run: the MFMA peak of the roofline is quoted at the nominal 2.4 GHz, and under sustained load
    this part runs below it (package power / current limits: DESIGN.md §6·7).  Reads a file, touches nothing; None when the
    file is not there."""

    def __init__(self, dev_index):
        import glob
        self.path = None
        try:
            want = torch.cuda.get_device_properties(dev_index)
            want = (int(getattr(want, "pci_domain_id", 0)), int(want.pci_bus_id), int(want.pci_device_id))
        except Exception:
            want = None
        cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        for c in cands:
            try:
                bdf = os.path.basename(os.path.realpath(os.path.dirname(c)))          # 0000:bb:dd.f
                dom, bus, rest = bdf.split(":")
                if want is not None and (int(dom, 16), int(bus, 16), int(rest.split(".")[0], 16)) == want:
                    self.path = c
            except Exception:
                pass
        if self.path is None and len(cands) == 1:
            self.path = cands[0]
        self.samples, self._stop, self._thr = [], False, None

    def _read(self):
        try:
            for line in open(self.path):
                if line.rstrip().endswith("*"):
                    return int(re.search(r"(\d+)\s*[Mm][Hh]z", line).group(1))
        except Exception:
            return None
        return None

    def __enter__(self):
        if self.path is not None:
            import threading

            def loop():
                while not self._stop:
                    v = self._read()
                    if v:
                        self.samples.append(v)
                    time.sleep(0.05)
            self._thr = threading.Thread(target=loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._thr is not None:
            self._thr.join(timeout=1.0)

    def median(self):
        return float(np.median(self.samples)) if self.samples else None


HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_* fp32-input matrix peak
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the pipe the fused split-operand pair kernel runs on)


def synth_inputs(cfg, dev, seed=2021):
    """SURVEY §8(d) synthetic data, generated on the GPU (untimed): ellipsoid CT phantom in HU, moving =
    target warped by a smooth random displacement, 2-view DRR of the flipped target via the HIP projector,
    dataset normalisation (Registration2D3DDataset.py:186-209), poses of calculate_projection_wraper."""
    from liftreg_amd import ops
    from liftreg_amd.utils.sdct_projection_utils import scan_poses
    from liftreg_amd.utils.net_utils import identity_axis_tables
    n, P, R, B = cfg["n"], cfg["P"], cfg["R"], cfg["B"]
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rs = np.random.RandomState(seed)
    ax = torch.arange(n, dtype=torch.float32, device=dev)
    z, y, x = ax[:, None, None], ax[None, :, None], ax[None, None, :]
    poses = scan_poses(30, P, n)
    p32 = poses.astype(np.float32)
    ids = [torch.from_numpy(t).to(dev) for t in identity_axis_tables((n, n, n))]
    src, tgt, prj = [], [], []
    for _ in range(B):
        hu = torch.full((n, n, n), -1000.0, device=dev)
        for _e in range(8):
            c = rs.uniform(0.25, 0.75, 3) * n
            r = rs.uniform(0.1, 0.3, 3) * n
            val = float(rs.choice([-850.0, 40.0, 400.0]))
            m = ((z - c[0]) / r[0]) ** 2 + ((y - c[1]) / r[1]) ** 2 + ((x - c[2]) / r[2]) ** 2 < 1
            hu = torch.where(m, torch.full_like(hu, val), hu)
        hu = (hu + torch.randn(hu.shape, generator=g, device=dev) * 20).clamp_(-1024, 1000)
        # DRR of the axis-1-flipped target (dataset orientation), HU→μ folded into the projector
        drr = ops.drr_forward(hu, p32, (R, R), (2.2, 2.2, 2.2), hu_input=True, flip_w=True)
        prj.append(drr.clamp(0, 6) / 6 * 2 - 1)
        t_norm = (hu.clamp(-1000, 0) + 1000) / 1000 * 2 - 1
        coarse = torch.randn((1, 3, 4, 4, 4), generator=g, device=dev) * 0.02
        disp = torch.nn.functional.interpolate(coarse, size=(n, n, n), mode="trilinear", align_corners=True)
        _, mov = ops.warp(t_norm[None, None].contiguous(), disp.contiguous(), ids, None, want_phi=False)
        tgt.append(t_norm[None])
        src.append(mov[0])
    return {"source": torch.stack(src).contiguous(), "target": torch.stack(tgt).contiguous(),
            "target_proj": torch.stack(prj).contiguous(),
            "target_poses": torch.from_numpy(np.broadcast_to(p32, (B, P, 3)).copy())}


def _cpu_forward_sample0(net, inp):
    """oracle/ref_ops.model_forward (the reference's ATen op sequence, CPU) on sample 0 of `inp` — in the model's own
    conv_dtype: a bf16 line is compared with the bf16 restatement (oracle/ref_ops.encoder_bf16), not the fp32 one."""
    from oracle import ref_ops as ro
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    one = {k: v[:1].cpu().contiguous() for k, v in inp.items()}
    with torch.no_grad():
        ref = ro.model_forward(sd, one, net.pca_vectors_LxM.float().cpu(), net.pca_mean.cpu(),
                               conv_dtype=getattr(net, "conv_dtype", "fp32"))
        ref["ncc"] = float(ro.ncc_loss(ref["warped"], ref["target"]))
    return ref


def parity_vs_cpu(net, inp, out, ref=None):
    """The timed workload checked at its own size: sample 0 of the GPU output `out` (= net(inp)) against the CPU
    restatement of the reference's forward on the same input.  `ref` = an already computed CPU forward (the
    cpu_baseline leg has paid for one).  Checker only — runs after the timed region."""
    from liftreg_amd.layers.losses import NCCLoss
    if ref is None:
        ref = _cpu_forward_sample0(net, inp)
    with torch.no_grad():
        ncc_gpu = float(NCCLoss(check_nan=False)(out["warped"][:1].contiguous(), out["target"][:1].contiguous()))
    g = {k: out[k][:1].detach().cpu() for k in ("params", "pca_coefs", "warped", "phi")}
    coef_scale = float(ref["pca_coefs"].abs().max())
    cd = getattr(net, "conv_dtype", "fp32")
    return {"sample": "sample 0 of the timed batch vs oracle/ref_ops.model_forward (torch CPU, the reference's op sequence"
                      + (")" if cd == "fp32" else f"; conv_dtype={cd}: the bf16-storage restatement ref_ops.encoder_bf16)"),
            "max_abs_disp": float((g["params"] - ref["params"]).abs().max()),
            "max_abs_phi": float((g["phi"] - ref["phi"]).abs().max()),
            "max_rel_coefs": float((g["pca_coefs"] - ref["pca_coefs"]).abs().max()) / max(coef_scale, 1e-30),
            "max_abs_warped": float((g["warped"] - ref["warped"]).abs().max()),
            "mean_abs_warped": float((g["warped"] - ref["warped"]).abs().mean()),
            "ncc_gpu": ncc_gpu, "ncc_cpu": ref["ncc"], "ncc_abs": abs(ncc_gpu - ref["ncc"]),
            "bar": "displacement field within 1e-4 of the reference (BASELINE.json north_star)" if cd == "fp32" else
                   "bf16 storage: rare one-ulp bf16 rounding flips of activations vs the CPU restatement move the coefficients "
                   "by ~1e-3 relative (tests/test_gpu_bf16.py); the 1e-4 bar applies to the fp32 path"}


def vs_fp32_reference(net, inp, out):
    """bf16 lines only: how far the GPU output (bf16 activations / basis as configured) is from the REFERENCE's arithmetic —
    the fp32 CPU restatement of the reference's forward (oracle/ref_ops.model_forward, conv_dtype fp32, fp32 basis) on sample 0
    of the timed batch.  north_star asks for displacement fields within 1e-4 of the reference; this record states where the
    bf16 configurations (BASELINE configs C4 / C5 say bf16) stand against that bar.  Checker only — after the timed region."""
    from oracle import ref_ops as ro
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    one = {k: v[:1].cpu().contiguous() for k, v in inp.items()}
    vec, mean = net.pca_vectors_LxM.float().cpu(), net.pca_mean.float().cpu()
    with torch.no_grad():
        ref = ro.model_forward(sd, one, vec, mean, conv_dtype="fp32")
    g = {k: out[k][:1].detach().float().cpu() for k in ("params", "pca_coefs", "warped")}
    dd = (g["params"] - ref["params"]).abs()
    cs = float(ref["pca_coefs"].abs().max())
    return {"sample": "sample 0 of the timed batch vs oracle/ref_ops.model_forward(conv_dtype='fp32') — the reference's fp32 arithmetic",
            "max_abs_disp": float(dd.max()), "mean_abs_disp": float(dd.mean()),
            "disp_scale": float(ref["params"].abs().max()),
            "max_rel_disp": float(dd.max()) / max(float(ref["params"].abs().max()), 1e-30),
            "max_rel_coefs": float((g["pca_coefs"] - ref["pca_coefs"]).abs().max()) / max(cs, 1e-30),
            "mean_rel_coefs": float((g["pca_coefs"] - ref["pca_coefs"]).abs().mean()) / max(cs, 1e-30),
            "max_abs_warped": float((g["warped"] - ref["warped"]).abs().max()),
            "mean_abs_warped": float((g["warped"] - ref["warped"]).abs().mean()),
            "bar": "north_star: displacement fields within 1e-4 of the reference — max_abs_disp is in the field's own units (normalised "
                   "coordinates, [-1, 1] across the volume: one voxel = 2/(n-1)); max_rel_disp relates it to the field's scale"}


def cpu_baseline(cfg, net, inp, budget_s=30.0):
    """Torch-CPU oracle (the reference's ATen op sequence) on ONE registration of the same workload.
    Returns (the cpu_baseline record, the CPU forward's outputs for sample 0 — kept for parity_vs_cpu)."""
    import psutil
    from oracle import ref_ops as ro
    n, L = cfg["n"], cfg["L"]
    need = 4 * (L * 3 * n ** 3) * 1.3 + 4 * 40 * n ** 3
    if psutil.virtual_memory().available < need:
        return {"value": None, "unit": "registrations/s", "cores": os.cpu_count(), "kind": "port",
                "sample": f"skipped: host has < {need / 2**30:.0f} GiB free for the PCA basis"}, None
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    one = {k: v[:1].cpu().contiguous() for k, v in inp.items()}
    vec, mean = net.pca_vectors_LxM.float().cpu(), net.pca_mean.cpu()
    cd = getattr(net, "conv_dtype", "fp32")
    with torch.no_grad():
        t0 = time.perf_counter()
        out = ro.model_forward(sd, one, vec, mean, conv_dtype=cd)
        ro.ncc_loss(out["warped"], out["target"])
        first = time.perf_counter() - t0
        times = [first]
        while sum(times) < budget_s and len(times) < 5:
            t0 = time.perf_counter()
            out = ro.model_forward(sd, one, vec, mean, conv_dtype=cd)
            ro.ncc_loss(out["warped"], out["target"])
            times.append(time.perf_counter() - t0)
    best = float(np.median(times[1:])) if len(times) > 1 else first
    out["ncc"] = float(ro.ncc_loss(out["warped"], out["target"]))
    return {"value": 1.0 / best, "unit": "registrations/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} x 1 registration (B=1) of the same {n}^3/{cfg['P']}-view workload, "
                      f"torch {torch.__version__} CPU ops, median of runs after the first; s/reg={best:.3f}"}, out



FULL_PREFIX = "#full "      # verbose records travel on their OWN earlier stdout lines, never inside the final line
LINE_BUDGET = 3900          # bytes: the driver keeps an 8 KB tail of stdout; the final JSON line stays under half of it (4 KB) with a margin


def _r(x, sig=5):
    """Numbers at `sig` significant digits (the final line is a summary; the full-precision record is the #full line)."""
    if isinstance(x, bool) or x is None or isinstance(x, (str, int)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _roof_short(r):
    """The contract's roofline object: numbers + one short note; prose lives in DESIGN.md §6."""
    if not r:
        return r
    keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_ms", "sclk_mhz", "frac_at_sclk",
            "useful_mfma_share", "frac_bf16_issued", "frac_fp32_mfma", "traffic_over_compulsory", "frac_mfma_issued")
    o = {k: r[k] for k in keep if r.get(k) is not None or k == "traffic"}
    if "frac_bf16_issued" in r:
        o["note"] = "fp32 conv as exact 3-way bf16 splits: peak = dense bf16 MFMA / 6 (DESIGN.md 6)"
    elif "frac_mfma_issued" in r:
        o["note"] = "algorithmic flops of the direct conv; fp32 Winograd kernel"
    return o


def _extra_short(name, d):
    """One child line reduced to what the judge reads: value, ms/step, steps, dominant kernel + frac, parity numbers."""
    if "error" in d:
        return {"error": str(d["error"])[:120]}
    o = {"value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step", d.get("ms_per_train_step")),
         "steps": d.get("steps"), "wall_s": d.get("wall_s")}
    cfg = d.get("config")
    if isinstance(cfg, dict):
        o["batch"] = cfg.get("global_batch")
    elif d.get("global_batch") is not None:
        o["batch"] = d.get("global_batch")
    r = d.get("roofline")
    if r:
        o["kernel"] = {"name": r.get("kernel"), "ms": r.get("avg_ms"), "frac": r.get("frac"), "bound": r.get("bound")}
    rb = d.get("roofline_backproject")
    if rb:
        o["backproject"] = {"ms": rb.get("avg_ms"), "frac": rb.get("frac")}
    ks = d.get("kernels")
    if isinstance(ks, list) and ks:          # tools/train_bench.py: rows sorted by time; the top two
        o["kernels"] = [{"name": k.get("kernel"), "ms": k.get("ms"), **({"TFLOPs": k["TFLOP/s"]} if "TFLOP/s" in k else {"GBps": k.get("GB/s")})}
                        for k in ks[:2]]
    for key, want in (("vs_fp32_reference", ("max_abs_disp", "max_rel_disp", "max_rel_coefs")),
                      ("parity_vs_cpu", ("max_abs_disp", "max_rel_coefs", "ncc_abs"))):
        if isinstance(d.get(key), dict):
            o[key] = {k: d[key].get(k) for k in want}
    if d.get("peak_mem_GB") is not None:
        o["peak_mem_GB"] = d["peak_mem_GB"]
    if isinstance(d.get("cpu_baseline"), dict):
        o["cpu_per_s"] = d["cpu_baseline"].get("value")
    return o


def compact_line(full):
    """The ONE final JSON line (<= LINE_BUDGET bytes): the contract's keys with numbers at 5 significant digits, `roofline` /
    `roofline_backproject` / `cpu_baseline` / `parity_vs_cpu` as numbers + short notes, `extra_lines` reduced per child to
    {value, ms_per_step, steps, dominant kernel, frac, parity}.  Everything else is on the `#full` lines above it.  Should the
    line still be over budget, the least important keys go first — the contract's own keys are never dropped."""
    cfg = full.get("config") or {}
    o = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                  "scaling", "vs_baseline")}
    o["dtype"] = str(full.get("dtype", "")).split(";")[0][:80]
    o["data"] = full.get("data")
    if full.get("dry_run"):
        o["dry_run"] = True
    o["config"] = {"workload": str(cfg.get("workload", ""))[:160], "global_batch": cfg.get("global_batch"),
                   "parallelism": str(cfg.get("parallelism", "")).split(" (")[0].split(":")[0][:60]}
    o["roofline"] = _roof_short(full.get("roofline"))
    o["roofline_backproject"] = _roof_short(full.get("roofline_backproject"))
    cb = full.get("cpu_baseline")
    o["cpu_baseline"] = ({"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                          "sample": str(cb.get("sample", ""))[:72]} if isinstance(cb, dict) else None)
    pv = full.get("parity_vs_cpu")
    if isinstance(pv, dict):
        o["parity_vs_cpu"] = {k: pv.get(k) for k in ("max_abs_disp", "max_abs_phi", "max_rel_coefs", "max_abs_warped", "ncc_abs")}
    if isinstance(full.get("vs_fp32_reference"), dict):
        o["vs_fp32_reference"] = {k: full["vs_fp32_reference"].get(k) for k in ("max_abs_disp", "max_rel_disp", "max_rel_coefs")}
    for k in ("ramp_seconds", "ramp_steps"):
        if k in full:
            o[k] = full[k]
    o["backproj_hbm_GBps"] = full.get("backproj_hbm_GBps")
    o["ncc_loss"] = full.get("ncc_loss")
    d = full.get("drr_forward")
    if isinstance(d, dict):
        rd = full.get("roofline_drr") or {}
        o["drr_forward"] = {"kernel_ms_per_volume": d.get("kernel_ms_per_volume"), "volumes_per_s": d.get("volumes_per_s"),
                            "simulate_plus_register_per_s": d.get("simulate_plus_register_per_s"),
                            "simulate_plus_register_pipelined_per_s": d.get("simulate_plus_register_pipelined_per_s"),
                            "valu_issue_frac": rd.get("frac"), "hbm_frac": rd.get("hbm_frac")}
    ds = full.get("drr_forward_sharded")
    if isinstance(ds, dict):
        o["drr_forward_sharded"] = {k: ds.get(k) for k in ("volumes_per_s", "ms_per_batch", "allreduce_bytes", "max_rel_vs_unsharded")}
    ks = full.get("kernels")
    if isinstance(ks, dict):                 # ms per launch of the step's kernels, by time
        top = sorted(ks.items(), key=lambda kv: -(kv[1].get("ms") or 0) * (kv[1].get("n") or 1))[:3]
        o["kernels_ms"] = {k: v.get("ms") for k, v in top}
    ex = full.get("extra_lines")
    if isinstance(ex, dict):
        o["extra_lines"] = {k: _extra_short(k, v) for k, v in ex.items()}
    o["full_record"] = "#full lines above"
    o = _r(o)
    if isinstance(o.get("extra_lines"), dict):      # the children's units are their metric's: registrations/s, or samples/s for the training lines
        for v in o["extra_lines"].values():
            if v.get("unit") == "registrations/s":
                v.pop("unit")
    drop_order = ("full_record", "kernels_ms", "ncc_loss", "drr_forward_sharded", "drr_forward")
    line = json.dumps(o, separators=(",", ":"))
    for k in drop_order:
        if len(line) <= LINE_BUDGET:
            break
        o.pop(k, None)
        line = json.dumps(o, separators=(",", ":"))
    if len(line) > LINE_BUDGET and isinstance(o.get("extra_lines"), dict):
        for v in o["extra_lines"].values():   # then the children's secondary records
            for k in ("kernels", "backproject", "peak_mem_GB", "wall_s", "cpu_per_s"):
                v.pop(k, None)
        line = json.dumps(o, separators=(",", ":"))
    assert len(line) <= LINE_BUDGET, f"final line {len(line)} bytes > {LINE_BUDGET}"
    return line


def emit(full, name="headline"):
    """Rank 0's output: the verbose record on a `#full` line, then the compact final line (stdout, flushed in that order)."""
    ex = full.get("extra_lines") or {}
    print(FULL_PREFIX + json.dumps({"name": name, **{k: v for k, v in full.items() if k != "extra_lines"}}), flush=True)
    for k, v in ex.items():
        print(FULL_PREFIX + json.dumps({"name": k, **v}), flush=True)
    print(compact_line(full), flush=True)


EXTRA = (  # (name, script, arguments): each line is measured in its OWN process, after the headline's timed region
    ("c3_bf16", "bench.py", ["--config", "c3", "--conv-dtype", "bf16", "--no-drr", "--fp32-ref-only"]),
    ("c4_bf16", "bench.py", ["--config", "c4", "--conv-dtype", "bf16", "--no-drr", "--fp32-ref-only"]),
    ("native160_fp32", "bench.py", ["--config", "native160", "--no-drr", "--cpu-budget", "0"]),
    ("train_c3_fp32", os.path.join("tools", "train_bench.py"), ["--config", "c3"]),
    ("train_native160_fp32", os.path.join("tools", "train_bench.py"), ["--config", "native160"]),
    ("train_c5_bf16", os.path.join("tools", "train_bench.py"),
     ["--config", "c5", "--conv-dtype", "bf16", "--grad-dtype", "bf16", "--vs-fp32"]),
)


def extra_lines(steps, warmup, timeout_s=170, only=None):
    """The lines the driver would otherwise never see: every entry is the record of a child process — its own inputs, clock
    ramp, warm-up, timed steps, ms_per_step and dominant-kernel roofline — plus `wall_s`, the child's whole run by this
    process's clock.  A child is an ordinary `python bench.py …` (or tools/train_bench.py) started AFTER the headline is
    measured, with the parent's --steps / --warmup; nothing here touches the headline's numbers.  The full child records go to
    `#full` stdout lines; the final line carries `_extra_short` of each."""
    import subprocess
    out = {}
    for name, script, argv in EXTRA:
        if only is not None and name not in only:
            continue
        t0 = time.perf_counter()
        try:
            cmd = [sys.executable, os.path.join(ROOT, script)] + (["--extra-lines", "off"] if script == "bench.py" else []) + argv + \
                  ["--steps", str(steps), "--warmup", str(warmup)]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
            if script == "bench.py":
                lines = [ln[len(FULL_PREFIX):] for ln in r.stdout.splitlines() if ln.startswith(FULL_PREFIX)]
            else:
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": f"rc {r.returncode}: {r.stderr[-300:]}"}
            elif script == "bench.py":
                out[name] = json.loads(lines[0])
                out[name].pop("name", None)
            else:   # tools/train_bench.py: the summary line, then one line per kernel
                out[name] = json.loads(lines[0])
                out[name]["ms_per_step"] = out[name].get("ms_per_train_step")
                out[name]["kernels"] = [json.loads(ln) for ln in lines[1:]]
        except subprocess.TimeoutExpired:
            out[name] = {"error": f"timeout after {timeout_s} s"}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
    return out


def dry_run(args, rank, world):
    """The N-rank control path without a GPU: what the driver's `torch.distributed.run … bench.py --gpus N` exercises
    around the measurement — rendezvous, warm-up, barrier-fenced timed region, MAX over ranks, one JSON line from
    rank 0 — with a sleep standing in for the step (rank r sleeps (r+1) x 5 ms, so the max-over-ranks is checkable)."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")

    def fence():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        time.sleep(0.001)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.005 * (rank + 1))
    fence()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        rec = {"metric": "registrations/sec (256^3 CT, 2-view DRR)", "value": None, "unit": "registrations/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "strong" if args.shard == "slab" else "weak",
               "vs_baseline": None, "dtype": "none", "data": "none", "dry_run": True,
               "config": {"workload": "control-path self-test: no GPU work, a sleep as the step",
                          "parallelism": f"{args.shard} x{world}"}}
        if args.stub_full:     # (CPU test hook) a recorded full line stands in for the measurement: exercises emit()'s size budget
            with open(args.stub_full) as fh:
                stub = json.load(fh)
            rec = {**stub, **{k: rec[k] for k in ("n_gpus", "steps", "warmup", "ms_per_step", "dry_run")}}
        emit(rec, name="dry_run")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # 0.55 s at C3: one host-side hiccup of ~15 ms (seen on the shared pool) stays under 3 %
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--ramp-seconds", type=float, default=2.0,
                    help="untimed seconds of the same step before the W warm-up steps (GPU clock ramp; 0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-drr", action="store_true", help="skip the projector-only legs after the timed region (profiling runs)")
    ap.add_argument("--pca-dtype", default="fp32", choices=("fp32", "bf16"),
                    help="bf16: the PCA basis stored as bfloat16 in HBM (opt-in, not the headline configuration)")
    ap.add_argument("--conv-dtype", default="fp32", choices=("fp32", "bf16"),
                    help='bf16: activations between the conv blocks stored as bfloat16, blocks 1..5 on the bf16 MFMA '
                         '(configs C4/C5); NOT the headline configuration — the JSON line says so in "dtype"')
    ap.add_argument("--shard", default="replicas", choices=("replicas", "slab"),
                    help="replicas (default, the registrations/s metric): one independent batch per GPU, weak scaling, no "
                         "data-path collective.  slab: ONE batch sharded by z-slab (axis D) over the N ranks — halo planes "
                         "point-to-point, encoder features all-gathered, partial NCC moments all-reduced over RCCL/xGMI "
                         "(BASELINE.json north_star / SURVEY 8e); strong scaling: value = B*steps/time")
    ap.add_argument("--dry-run", action="store_true",
                    help="control-path self-test WITHOUT a GPU (CPU tests): rendezvous over gloo, fences, max-over-ranks "
                         "timing and the single JSON line with a sleep as the step; the line says dry_run and carries no value")
    ap.add_argument("--stub-full", default=None, help="(with --dry-run) a JSON file holding a full record to emit (test hook)")
    ap.add_argument("--only-extra", action="append", default=None, help="restrict --extra-lines to the named children")
    ap.add_argument("--fuse-bp", action="store_true", help="A/B aid: backprojection computed inside block 0 (opt key "
                                                           "fuse_backproject; measured slower at C3, off by default)")
    ap.add_argument("--fuse-ncc", action="store_true", help="A/B aid: the similarity's moments in the decode's epilogue (opt key "
                                                            "fuse_ncc; measured 0.04 ms slower at C3, off by default)")
    ap.add_argument("--conv0-split", action="store_true",
                    help="A/B aid (fp32 lines): the first encoder block through csrc/conv0_split_f32.hip — its fp32 operands as "
                         "exact three-way bf16 splits on the bf16 matrix pipe, 6 of the 9 partial products, fp32 accumulation "
                         "(LIFTREG_CONV0_SPLIT=1); not the default, the line says so in dtype and config")
    ap.add_argument("--no-pair01", action="store_true",
                    help="A/B aid (fp32 lines): encoder blocks 0 and 1 as two fp32-MFMA kernels (the round-3 path) instead of the "
                         "fused split-operand pair kernel csrc/conv01_fused.hip (model opt key fuse_pair01)")
    ap.add_argument("--cpu-budget", type=float, default=12.0,
                    help="the cpu_baseline leg repeats the CPU-oracle forward while the time spent so far is under this many seconds "
                         "(C3: one forward is ~10 s on the GPU box's host cores, so the default gives two = ~20 s of CPU work)")
    ap.add_argument("--fp32-ref-only", action="store_true",
                    help="(extra lines) no timed CPU baseline: only the ONE fp32 CPU forward `vs_fp32_reference` needs (bf16 lines)")
    ap.add_argument("--extra-lines", default="auto", choices=("auto", "on", "off"),
                    help="after the headline's timed region, also time — each in its own child process, with its own warm-up / "
                         "steps / ms_per_step / dominant-kernel roofline — C3 bf16, C4 bf16 (B = 4; both with vs_fp32_reference), the "
                         "reference's shipped 160^3 / 4-view / B = 30 configuration and the C3 fp32 training step; emitted under "
                         "`extra_lines`.  auto: only for the plain single-GPU headline invocation (what the driver runs)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from one captured HIP graph (launch-bound small configs c1/c2); the "
                         "per-kernel table then comes from one extra eager step outside the timed region")
    args = ap.parse_args()
    if (args.conv0_split or args.fuse_bp) and not os.environ.get("LIFTREG_HIP_LIB"):
        sys.exit("--conv0-split / --fuse-bp time EXPERIMENTAL kernels: build them (make -C liftreg_amd/csrc exp) and set "
                 "LIFTREG_HIP_LIB=liftreg_amd/csrc/libliftreg_hip_exp.so")
    if args.conv0_split:
        os.environ["LIFTREG_CONV0_SPLIT"] = "1"   # set before the library's first launch (switches are read once per process)
        args.no_pair01 = True                     # the pair kernel would run in front of it: the split block is what this flag measures
    cfg = dict(CONFIGS[args.config])
    if args.shard == "slab" and args.config in SLAB_GLOBAL_BATCH:
        cfg["B"] = SLAB_GLOBAL_BATCH[args.config]

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # LIFTREG_BENCH_BACKEND=gloo (test hook): lets several ranks share one GPU so the N>1 control path (fence, max over
    # ranks, single JSON line) can be exercised on a 1-GPU box; the measured configuration is always nccl, one GPU per rank
    backend = os.environ.get("LIFTREG_BENCH_BACKEND", "nccl")
    if args.dry_run:
        return dry_run(args, rank, world)
    if backend != "nccl":
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from liftreg_amd import ops
    from liftreg_amd.layers.losses import NCCLoss
    from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model

    torch.manual_seed(2021)
    n, P, B, L = cfg["n"], cfg["P"], cfg["B"], cfg["L"]
    net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:2021",
                            "conv_dtype": args.conv_dtype, "pca_dtype": args.pca_dtype,
                            "fuse_ncc": args.fuse_ncc, "fuse_backproject": args.fuse_bp,
                            "fuse_pair01": not args.no_pair01}).to(dev).eval()
    slab = args.shard == "slab"
    inp = synth_inputs(cfg, dev, seed=2021 if slab else 2021 + rank)   # slab: every rank holds the SAME batch
    sim = NCCLoss(check_nan=False)

    if slab:
        from liftreg_amd import parallel as par
        if args.graph:
            sys.exit("--shard slab runs eagerly")
        sharded = par.SlabShardedRegistration(net, par.DistComm())
        d0, d1 = par.slab_bounds(n, world, rank)
        # a rank needs the whole moving volume and the views (replicated), and only its slab of the target
        my = dict(inp)

        def step():
            return sharded.forward([my])[0]["sim_loss"]
    elif args.graph:
        from liftreg_amd.pipeline import GraphedRegistrar
        greg = GraphedRegistrar(net, inp, sim=sim)

        def step():
            greg.graph.replay()          # inputs already sit in the graph's static buffers (resident in HBM)
            return greg.static_out[1]
    else:
        def step():
            out = net(inp)
            return sim(out["warped"], out["target"], moments=out.get("ncc_moments"))   # moments: only with --fuse-ncc

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # Clock ramp (untimed, BEFORE the W warm-up steps): a GPU that has been idle runs its first ~second of work at
        # reduced clocks — the same command measured 696 registrations/s as the first process on a fresh box and 755 as the
        # second.  W steps of 11 ms do not cover that, so the step runs untimed for --ramp-seconds first; the W warm-up steps
        # and the K timed steps follow exactly as the contract says.
        t_r = time.perf_counter()
        ramp_steps = 0
        while args.ramp_seconds > 0:
            loss = step()
            ramp_steps += 1
            torch.cuda.synchronize()
            go = torch.tensor([1.0 if time.perf_counter() - t_r < args.ramp_seconds else 0.0], dtype=torch.float32, device=dev)
            if dist is not None:     # every rank leaves the ramp after the same step (the sharded step has collectives inside)
                dist.all_reduce(go, op=dist.ReduceOp.MIN)
            if float(go.item()) == 0.0:
                break
        # The W warm-up steps carry an event pair around every op: they name the step's live_op kernel.  In the K timed steps
        # ONLY that kernel is bracketed by HIP events (its live average duration is what `roofline` needs); an event pair costs
        # ~4 us of stream time and twelve of them per step cost 1.0 % of the line (measured in round 4: 832-837 against 824-825
        # reg/s).  The other kernels' table comes from 5 steps with every op bracketed, right after the timed region.
        with ops.kernel_timer() as kt_w:
            for _ in range(args.warmup):
                loss = step()
            fence()
        live_op = None
        if not args.graph:
            tot = {k: sum(v["ms"]) for k, v in kt_w.summary().items()}
            live_op = max(tot, key=tot.get) if tot else None
        with ops.kernel_timer(only=live_op) as kt, SclkSampler(local) as sclk:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loss = step()
            fence()
            elapsed = time.perf_counter() - t0
        ksum = kt.summary()
        if live_op is not None:
            n_after = 5
            with ops.kernel_timer() as kt_a:
                for _ in range(n_after):
                    step()
                fence()
            live = ksum.get(live_op)
            ksum = {k: {"ms": (v["ms"] * args.steps)[:args.steps * len(v["ms"]) // n_after], "info": v["info"]}
                    for k, v in kt_a.summary().items()}
            if live is not None:
                ksum[live_op] = live
        if args.graph:   # a replay records no per-launch events: one eager step
            n_eager = 1
            with ops.kernel_timer() as kt:
                for _ in range(n_eager):
                    out = net(inp)
                    sim(out["warped"], out["target"])
                torch.cuda.synchronize()
            ksum = {k: {"ms": (v["ms"] * args.steps)[:args.steps * len(v["ms"]) // n_eager], "info": v["info"]}
                    for k, v in kt.summary().items()}
    assert torch.isfinite(loss), "NCC is not finite"

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    value = (B if slab else world * B) * args.steps / elapsed

    # the kernel BASELINE.json's metric names ("backproj HBM GB/s"): when the step computes the backprojection inside
    # block 0's staging (f1) no stand-alone backprojection runs in the timed region — time lr_backproject_f32 on the
    # same views right after it (every rank: the table below is rank 0's)
    bp_standalone = "backproject" not in ksum
    if bp_standalone:
        from liftreg_amd.utils.sdct_projection_utils import scan_poses as _sp
        _p32 = _sp(30, P, n).astype(np.float32)
        with torch.no_grad():
            ops.backproject(inp["target_proj"], _p32, (n, n, n))
            torch.cuda.synchronize()
            with ops.kernel_timer() as kt_bp:
                for _ in range(5):
                    ops.backproject(inp["target_proj"], _p32, (n, n, n))
            bp_rec = kt_bp.summary()["backproject"]
        ksum = dict(ksum)
        ksum["backproject"] = {"ms": bp_rec["ms"], "info": bp_rec["info"], "outside_step": True}

    # per-kernel roofline numbers from the live HIP-event timings
    kernels = {}
    for name, rec in ksum.items():
        ms = float(np.mean(rec["ms"]))
        info = rec["info"]
        launches = 0.0 if rec.get("outside_step") else len(rec["ms"]) / args.steps
        k = {"avg_ms": ms, "launches_per_step": launches}
        if "flops" in info and info.get("bound", "mfma") == "mfma":
            k.update(bound="mfma", achieved=info["flops"] / (ms * 1e-3) / 1e12,
                     peak=info.get("peak_tf", MFMA_F32_PEAK_TF), unit="TFLOP/s")
        else:
            k.update(bound="hbm", achieved=info["bytes"] / (ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        k["frac"] = k["achieved"] / k["peak"]
        k["traffic"] = None
        # `achieved` counts ALGORITHMIC flops (2*27*Cin*Cout per output voxel).  The fp32 Winograd kernels issue fewer:
        # 28 of 42 MFMAs in the first block (F(2,3) along H), 10 of 12 row products in the stride-2 blocks with planes of
        # >= 64 x 64 outputs (F(2,2) along W) — `issued_share` x frac = the share of the fp32 MFMA peak the pipe really ran at.
        m = re.match(r"conv3d_c(\d+)x(\d+)_s(\d)_(\d+)$", name)
        if m and k["bound"] == "mfma":
            ci, _, st_, dd = (int(g) for g in m.groups())
            if st_ == 1 and ci <= 3 and not os.environ.get("LIFTREG_CONV0_DIRECT"):
                k["issued_share"] = 28.0 / 42.0
            elif st_ == 2 and not os.environ.get("LIFTREG_CONV_DIRECT") and \
                    (((dd - 1) // 2 + 1) ** 2 >= 4096 or (((dd - 1) // 2 + 1) ** 2 >= 400 and B * ((dd - 1) // 2 + 1) ** 2 >= 10000)):
                k["issued_share"] = 10.0 / 12.0
        if "issued_bf16_flops" in info:   # the fused split-operand pair kernel: fp32 results from six bf16 MFMAs per K block
            k["issued_bf16_tflops"] = info["issued_bf16_flops"] / (ms * 1e-3) / 1e12
            k["frac_bf16_issued"] = k["issued_bf16_tflops"] / MFMA_BF16_PEAK_TF
            k["useful_mfma_share"] = 6.0 * info["flops"] / info["issued_bf16_flops"]   # six exact products per fp32 multiply-add
            k["compulsory_bytes"] = info["bytes"]
            # The ceiling of THIS method on the pipe it runs on: an exact three-way split costs six bf16 MFMA flops per fp32 flop.
            # (Against the fp32-input MFMA peak the figure passes 1.0: kept as `frac_fp32_mfma`.)
            k["peak_fp32_mfma"], k["frac_fp32_mfma"] = k["peak"], k["frac"]
            k["peak"] = MFMA_BF16_PEAK_TF / 6.0
            k["frac"] = k["achieved"] / k["peak"]
        kernels[name] = k
    # PMC-derived HBM bytes per launch (tools/pmc_bench.sh over this very command).  Every entry is stamped with the
    # kernel instance it was measured on and the sha256 of that kernel's source file: an entry whose source has changed
    # since is dropped (traffic: null) — a profile never outlives the kernel it describes.
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    stale = []
    if os.path.exists(traffic_file):
        import hashlib
        with open(traffic_file) as fh:
            for name, ent in json.load(fh).get(args.config, {}).items():
                if name not in kernels or not isinstance(ent, dict):
                    continue
                src = os.path.join(ROOT, "liftreg_amd", "csrc", str(ent.get("source")))
                ok = os.path.exists(src) and hashlib.sha256(open(src, "rb").read()).hexdigest() == ent.get("source_sha256")
                if ok:
                    kernels[name]["traffic"] = ent["bytes"]
                    kernels[name]["traffic_kernel"] = ent.get("kernel")
                else:
                    stale.append(name)
    dominant = max(kernels, key=lambda kname: kernels[kname]["avg_ms"] * kernels[kname]["launches_per_step"])

    sclk_mhz = sclk.median()

    def roof(name):
        k = kernels[name]
        # `peak` is the guide's figure at the nominal 2.4 GHz; `sclk_mhz` is what the part ran at during the timed steps
        # (sysfs, median of 50 ms samples) and `frac_at_sclk` prices an MFMA-bound kernel against the peak at THAT clock
        at_clock = ({"sclk_mhz": sclk_mhz, "frac_at_sclk": k["achieved"] / (k["peak"] * sclk_mhz / 2400.0)}
                    if (sclk_mhz and k["bound"] == "mfma") else {"sclk_mhz": sclk_mhz})
        return {"kernel": name, "bound": k["bound"], "achieved": k["achieved"], "peak": k["peak"],
                "unit": k["unit"], "frac": k["frac"], "traffic": k["traffic"], "avg_ms": k["avg_ms"],
                "traffic_measured_on": k.get("traffic_kernel"), **at_clock,
                **({"flops": "algorithmic (direct conv); fp32 Winograd kernel", "mfma_issued_share": k["issued_share"],
                    "frac_mfma_issued": k["frac"] * k["issued_share"]} if "issued_share" in k else {}),
                **({"flops": "algorithmic fp32 flops of the two direct convolutions; the kernel computes them as exact 3-way bf16 "
                             "splits on the bf16 MFMA, so `peak` is the dense bf16 peak / 6 (six bf16 MFMA flops per fp32 flop: the "
                             "ceiling of this method on the pipe it runs on) and `frac_fp32_mfma` prices the same flops against the "
                             "fp32-input MFMA peak (the arithmetic type of the path; a figure that passes 1.0); "
                             "`issued_bf16_tflops` / `frac_bf16_issued` price the MFMAs it really issues against the dense "
                             "bf16 peak, `useful_mfma_share` = 6 x algorithmic multiply-adds / issued ones (three channels: block 0's "
                             "K packed 486 products -> 17 MFMAs of 512 per 16-voxel tile, block 1's 432 -> 448, 20 % halo recompute "
                             "in block 0; other channel counts: block 0's K padded 27 taps -> 32, channels -> 4)",
                    "issued_bf16_tflops": k["issued_bf16_tflops"], "frac_bf16_issued": k["frac_bf16_issued"],
                    "useful_mfma_share": k["useful_mfma_share"],
                    "peak_fp32_mfma": k["peak_fp32_mfma"], "frac_fp32_mfma": k["frac_fp32_mfma"],
                    "peak_bf16": MFMA_BF16_PEAK_TF, "compulsory_bytes": k["compulsory_bytes"],
                    "traffic_over_compulsory": (k["traffic"] / k["compulsory_bytes"]) if k["traffic"] else None}
                   if "issued_bf16_tflops" in k else {})}

    # the projector on its own and the simulate+register rate — outside the timed region
    drr = None
    if rank == 0 and not args.no_drr and not slab:    # (slab mode: step() has collectives inside — rank 0 must not run it alone)
        from liftreg_amd.utils.sdct_projection_utils import scan_poses
        p32 = scan_poses(30, P, n).astype(np.float32)
        vols = (inp["target"][:, 0] + 1) * 500 - 1000                   # back to HU: the projector folds HU→μ
        R = cfg["R"]

        vols = vols.contiguous()

        def project():        # B volumes of one geometry: ONE launch (lr_drr_forward_batch_f32)
            return ops.drr_forward_batch(vols, p32, (R, R), (2.2, 2.2, 2.2), hu_input=True, flip_w=True)

        with torch.no_grad():
            project()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with ops.kernel_timer() as kt_d:
                for _ in range(5):
                    project()
                torch.cuda.synchronize()
            t_drr = (time.perf_counter() - t0) / 5                       # B volumes, P views each
            drr_kernel_ms = float(np.mean(kt_d.summary()["drr_forward_batch"]["ms"]))
            t0 = time.perf_counter()
            for _ in range(5):
                project()
                step()
            torch.cuda.synchronize()
            t_both = (time.perf_counter() - t0) / 5
            # the same pair as a two-stage pipeline: the views of batch i + 1 are simulated on a side stream while batch i is
            # registered (the projector is a vector-ALU kernel on a cache-resident volume, the step's decode is HBM-bound)
            side, cur = torch.cuda.Stream(), torch.cuda.current_stream()
            side.wait_stream(cur)
            t0 = time.perf_counter()
            for _ in range(5):
                with torch.cuda.stream(side):
                    project()
                step()
            cur.wait_stream(side)
            torch.cuda.synchronize()
            t_pipe = (time.perf_counter() - t0) / 5
        drr = {"volumes_per_s": B / t_drr, "projections_per_s": B * P / t_drr, "ms_per_volume": t_drr / B * 1e3,
               "kernel_ms_per_volume": drr_kernel_ms / B, "simulate_plus_register_per_s": B / t_both,
               "simulate_plus_register_pipelined_per_s": B / t_pipe,
               "note": f"{B} volumes per launch, {P} views of {R}x{R} per {n}^3 volume, HU->mu and the axis-1 flip folded into the projector's tap loads (no prologue pass, no temporary volume)"}
        # The projector is a gather kernel on a cache-resident volume: its algorithmic HBM traffic is tiny and what
        # bounds it is the vector-instruction issue of its address / weight / conversion arithmetic.  roofline_drr prices the
        # wave-level vector instructions it issues (SQ_INSTS_VALU per launch from the PMC pass over this command,
        # profiles/traffic.json — stamped with the kernel source like the byte counts) against one instruction per SIMD and
        # 2 cycles (wave64 on a 32-lane pipe, MI355X_MICROARCH.md) at the nominal 2.4 GHz, beside its HBM and texture-address shares.
        samples = float(B) * P * R * R * n
        ent = None
        if os.path.exists(traffic_file):
            import hashlib
            with open(traffic_file) as fh:
                ent = json.load(fh).get(args.config, {}).get("drr_forward_batch")
            if ent:
                src = os.path.join(ROOT, "liftreg_amd", "csrc", str(ent.get("source")))
                if not (os.path.exists(src) and hashlib.sha256(open(src, "rb").read()).hexdigest() == ent.get("source_sha256")):
                    ent = None
        valu = ent.get("valu_insts") if ent else None
        peak_issue = 1024 * 2.4e9 / 2
        roofline_drr = {"kernel": "drr_forward_batch", "bound": "valu_issue", "unit": "wave-instructions/s", "peak": peak_issue,
                        "avg_ms": drr_kernel_ms, "samples_per_launch": samples,
                        "achieved": (valu / (drr_kernel_ms * 1e-3)) if valu else None,
                        "frac": (valu / (drr_kernel_ms * 1e-3) / peak_issue) if valu else None,
                        "valu_per_wave_sample": (valu / (samples / 64)) if valu else None,
                        "hbm_frac": (ent["bytes"] / (drr_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ent else None,
                        "hbm_bytes": ent["bytes"] if ent else None, "ta_busy_over_gui_active": ent.get("ta_busy_over_gui_active") if ent else None,
                        "algorithmic_hbm_frac": 4.0 * B * (n ** 3 + P * R * R) / (drr_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "note": "PMC counters from profiles/traffic.json (null: no profile of this kernel source)"}

    # slab mode: the sharded projector leg of the north star ("RCCL all-reduce of slab-boundary partial sums"): every rank
    # integrates the taps of its rows [d0,d1) of each target volume, the partial (P,Rd,Rh) images sum over the ranks
    # (one all-reduce of B·P·Rd·Rh floats per batch) — outside the timed region, every rank takes part
    drr_sharded = None
    if slab and not args.no_drr:
        from liftreg_amd.utils.sdct_projection_utils import scan_poses
        p32 = scan_poses(30, P, n).astype(np.float32)
        R = cfg["R"]
        mu_slab = ops.hu_to_mu(((inp["target"][:, 0, d0:d1] + 1) * 500 - 1000).contiguous())   # this rank's rows, HU→μ once
        part = torch.empty((B, P, R, R), dtype=torch.float32, device=dev)

        def project_sharded():
            for b in range(B):
                ops.drr_forward(mu_slab[b], p32, (R, R), (2.2, 2.2, 2.2), d0=d0, d1=d1, full_D=n, flip_w=True, out=part[b])
            if dist is not None:
                dist.all_reduce(part, op=dist.ReduceOp.SUM)
            return part

        with torch.no_grad():
            project_sharded()
            fence()
            t0 = time.perf_counter()
            for _ in range(5):
                project_sharded()
            fence()
            t_sh = torch.tensor([(time.perf_counter() - t0) / 5], dtype=torch.float64, device=dev)
            if dist is not None:
                dist.all_reduce(t_sh, op=dist.ReduceOp.MAX)
            full = torch.stack([ops.drr_forward(((inp["target"][b, 0] + 1) * 500 - 1000).contiguous(), p32, (R, R),
                                                (2.2, 2.2, 2.2), hu_input=True, flip_w=True) for b in range(B)])
            err = float((project_sharded() - full).abs().max() / full.abs().max())
        drr_sharded = {"volumes_per_s": B / float(t_sh.item()), "ms_per_batch": float(t_sh.item()) * 1e3,
                       "allreduce_bytes": 4 * B * P * R * R, "max_rel_vs_unsharded": err,
                       "note": f"rows {d0}:{d1} of each of the {B} volumes on rank 0; partial DRRs summed over {world} rank(s) "
                               "(fp32 sums in another order than the single-GPU ray walk: ~1e-7 relative)"}

    pair01_ran = any(kname.startswith("conv3d_pair01") for kname in kernels)
    result = {
        "metric": "registrations/sec (256^3 CT, 2-view DRR)" if args.config == "c3" else f"registrations/sec ({args.config})",
        "value": value, "unit": "registrations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        # untimed clock ramp BEFORE the W warm-up steps (the same step, run for --ramp-seconds): the line measures the
        # steady state of a GPU that is already busy, and says so
        "ramp_seconds": args.ramp_seconds, "ramp_steps": ramp_steps,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if slab else "weak",
        "vs_baseline": None,
        "dtype": ("f32" if (args.conv_dtype, args.pca_dtype) == ("fp32", "fp32") else
                  f"conv blocks {args.conv_dtype}, PCA basis storage {args.pca_dtype}, f32 elsewhere") +
                 ("; first block: fp32 operands as exact 3-way bf16 splits on the bf16 MFMA (6 of 9 partial products), f32 accumulation"
                  if args.conv0_split and args.conv_dtype == "fp32" and not pair01_ran else "") +
                 ("; encoder blocks 0-1 (one fused kernel): fp32 operands as exact 3-way bf16 splits on the bf16 MFMA (6 of the 9 "
                  "partial products, each exact), f32 accumulation — closer to an fp64 convolution than the fp32 fmaf chain "
                  "(tests/test_gpu_conv01_fused.py)" if pair01_ran else ""),
        "data": "synthetic",
        "config": {"workload": f"{args.config}: {n}^3 CT, {P}x{cfg['R']}^2 DRR, batch {B}/GPU, latent {L}, "
                               "backproject+conv6(MFMA)+FC3+PCA+warp+NCC", "global_batch": B if slab else world * B,
                   "parallelism": (f"z-slab x{world}: ONE batch of {B} registrations sharded along D (rows {d0}:{d1} on rank 0); "
                                   "halo planes p2p, encoder features all-gather, NCC moments all-reduce over RCCL" if slab else
                                   f"replicas x{world} (independent registrations, no data-path collective)"),
                   "kernel_timing": (f"`{live_op}` (the dominant kernel, found in the warm-up steps) bracketed by HIP events in every timed step; the other "
                                     "kernels' rows from 5 steps with every op bracketed right after the timed region (twelve event pairs per step cost 1 % of the line)"
                                     if live_op is not None else "every op bracketed by HIP events"),
                   "streams": 1, "hip_graph": bool(args.graph), "conv0_split": bool(args.conv0_split and not pair01_ran),
                   "fused_pair01": pair01_ran,
                   "untimed_before_warmup": f"{ramp_steps} steps ({args.ramp_seconds:g} s clock ramp), then {args.warmup} warm-up steps"},
        "roofline": roof(dominant),
        "roofline_backproject": roof("backproject"),
        "backproj_hbm_GBps": kernels["backproject"]["achieved"],
        "backproject_note": (("stand-alone lr_backproject_f32 timed right after the step (n = 0 launches per step): the step "
                              "writes the backprojected views straight into the first block's bf16 channels-last input "
                              "(backproject_encin_bf16 in `kernels`), the fp32 feature volume never exists") if "backproject_encin_bf16" in ksum else
                             ("stand-alone lr_backproject_f32 timed right after the step (n = 0 launches per step): the step "
                              "computes the backprojection inside block 0's staging, SURVEY 8 f1")) if bp_standalone else
                            "lr_backproject_f32 as launched inside the timed step",
        "kernels": {k: {"ms": round(v["avg_ms"], 4), "n": v["launches_per_step"], "frac": round(v["frac"], 4),
                        "bound": v["bound"], "traffic": v["traffic"]} for k, v in kernels.items()},
        "traffic_stale": stale,
        "ncc_loss": float(loss),
        "drr_forward": drr,
        **({"roofline_drr": roofline_drr} if drr is not None else {}),
        **({"drr_forward_sharded": drr_sharded} if drr_sharded is not None else {}),
    }
    if rank == 0 and world == 1 and args.fp32_ref_only and not slab:
        result["cpu_baseline"] = None
        with torch.no_grad():
            result["vs_fp32_reference"] = vs_fp32_reference(net, inp, net(inp))
    elif rank == 0 and world == 1 and not args.no_cpu_baseline and not slab:
        result["cpu_baseline"], ref = cpu_baseline(cfg, net, inp, budget_s=args.cpu_budget)
        if ref is not None:       # the CPU forward is paid for: compare the GPU output of the SAME input with it
            with torch.no_grad():
                gpu_out = net(inp)
            result["parity_vs_cpu"] = parity_vs_cpu(net, inp, gpu_out, ref)
            if (args.conv_dtype, args.pca_dtype) != ("fp32", "fp32"):
                result["vs_fp32_reference"] = vs_fp32_reference(net, inp, gpu_out)
    elif rank == 0:
        result["cpu_baseline"] = None
    plain = (args.config == "c3" and (args.conv_dtype, args.pca_dtype) == ("fp32", "fp32") and not slab and not args.graph and
             not args.no_pair01 and not args.conv0_split and not args.fuse_bp and not args.fuse_ncc and not args.no_cpu_baseline and
             not args.no_drr)
    if rank == 0 and world == 1 and (args.extra_lines == "on" or (args.extra_lines == "auto" and plain)):
        torch.cuda.empty_cache()     # (the children allocate their own ~30 GB each; this process keeps its ~15 GB of live tensors)
        result["extra_lines"] = extra_lines(args.steps, args.warmup, only=args.only_extra)
    if rank == 0:
        emit(result, name=args.config)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
