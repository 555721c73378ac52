#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc CSVs of a `tools/pmc_bench.sh` run (PMC passes over bench.py itself) into
  * a per-kernel counter summary (stdout → summary.txt), one block per (kernel template instance, grid size), and
  * traffic.json: HBM bytes per launch for every op of the bench step = (2·FETCH_SIZE + WRITE_SIZE)·1024, i.e. with the
    gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports exactly half of a
    16-B/lane streaming read; WRITE_SIZE is exact), each entry stamped with the demangled kernel name it was measured
    on, the kernel's source file and that file's sha256.  bench.py drops (traffic: null) any entry whose source file no
    longer hashes to the stamp — a profile can never outlive the kernel it describes.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "liftreg_amd", "csrc")
CONFIG_N = {"c1": 64, "c2": 128, "c3": 256, "c4": 256, "native160": 160}


def short_name(k):
    k = re.sub(r"^void\s+", "", k).replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z0-9_:]+(<[^()]*>)?)", k)
    return m.group(1) if m else k[:80]


def source_of(kernel):
    base = re.match(r"[A-Za-z0-9_]+", kernel).group(0)
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):      # the file that DEFINES the kernel, not one that mentions it
        if re.search(r"\bvoid\s+" + re.escape(base) + r"\s*\(", open(f).read()):
            return os.path.basename(f)
    return None


def sha256(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def op_table(cfg, P, bf16):
    """(bench op name, kernel regex, rank among same-regex groups by descending grid size)."""
    n = CONFIG_N[cfg]
    ops = [("drr_forward_batch", r"^drr_forward_fast_kernel", 0), ("backproject", r"^backproject_(tiled_)?kernel", 0), ("pca_warp", r"^pca_warp_kernel<.*, false(, (true|false))?>$", 0),
           ("pca_warp_ncc", r"^pca_warp_kernel<.*, true>$", 0), ("ncc_moments", r"^ncc_moments_kernel", 0),
           (f"conv3d_c{P + 1}x16_s1_{n}", r"^(conv3d_planar_kernel|conv0_split_f32_kernel<.*, 3, false, false>)", 0), (f"conv3d_bp_c{P + 1}x16_s1_{n}", r"^conv0_pc_kernel<.*, true>$", 0),
           (f"conv3d_c16x32_s2_{n}", r"^conv3d_rows_wlds_kernel<2, 1,", 0),
           (f"conv3d_pair01_c{P + 1}x16x32_{n}", r"^conv01_fused_kernel", 0)]
    # stride-2 blocks: planes of >= 64 x 64 outputs run the persistent Winograd rows kernel (one grid size for all of them:
    # told apart by rank only if there are several), smaller ones the direct rows kernel
    if bf16:   # --conv-dtype bf16: the first block (all channels at once for > 3 of them, else the channel-pass kernel), then the row kernels
        ops = ops[:4] + [(f"conv3d_bf16_c{P + 1}x16_s1_{n}", r"^conv0_cl_bf16_kernel<.*, false>$" if P + 1 > 3 else r"^(conv0_split_f32_kernel<.*, 1, true, (true|false)>|conv0_bf16_kernel)", 0),
                         (f"conv3d_bf16_c{P + 1}x16_s1_{n}_clin", r"^conv0_cl_bf16_kernel<.*, true>$", 0),
                         ("backproject_encin_bf16", r"^backproject_encin_bf16_kernel", 0),
                         (f"conv3d_bf16_c16x32_s2_{n}", r"^(conv3d_march_s2_bf16_kernel|conv3d_cl_rows_bf16_kernel<2, 4, false>)", 0)]
        # 32 -> 32 blocks: the z-marching kernel while batch x columns >= 256 (conv3d_bf16.hip: march_pays), else the row kernel
        batch = {"c1": 1, "c2": 4, "c3": 8, "c4": 4, "c5": 4}.get(cfg, 8)
        size, rank_m, rank_r = n // 2, 0, 0
        while size >= 16:
            if batch * ((size // 2 + 3) // 4) * ((size // 2 + 7) // 8) >= 256:
                ops.append((f"conv3d_bf16_c32x32_s2_{size}", r"^conv3d_march_s2_c32_bf16_kernel", rank_m))
                rank_m += 1
            else:
                ops.append((f"conv3d_bf16_c32x32_s2_{size}", r"^conv3d_cl_rows_bf16_kernel<(1|2), 4, true>", rank_r))
                rank_r += 1
            size //= 2
        return ops
    size, rank_w, rank_d = n // 2, 0, 0
    while size >= 16:
        if (size // 2) ** 2 >= 4096 or ((size // 2) ** 2 >= 400 and {"native160": 30}.get(cfg, 8) * (size // 2) ** 2 >= 10000):
            ops.append((f"conv3d_c32x32_s2_{size}", r"^conv3d_rows_wlds_kernel<2, 2,", rank_w))
            rank_w += 1
        else:
            ops.append((f"conv3d_c32x32_s2_{size}", r"^conv3d_cl_rows_kernel<2, 2>", rank_d))
            rank_d += 1
        size //= 2
    return ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--bench-args", default="")
    a = ap.parse_args()
    cfg = (re.search(r"--config\s+(\w+)", a.bench_args) or [None, "c3"])[1]
    P = 11 if cfg == "c4" else 4 if cfg == "native160" else 2
    agg = defaultdict(lambda: defaultdict(list))       # (kernel, grid) -> counter -> values
    meta = {}
    for f in glob.glob(os.path.join(a.out, "p*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "at::native" in k or "rocclr" in k or "Cijk" in k:
                continue
            key = (short_name(k), int(row["Grid_Size"]))
            agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[key] = {"vgpr": row.get("VGPR_Count"), "lds": row.get("LDS_Block_Size"), "wg": row.get("Workgroup_Size")}
    if not agg:
        sys.exit("no counter CSVs under " + a.out)
    print(f"# PMC over: python3 bench.py {a.bench_args}   (average per dispatch; separate --pmc passes)")
    groups = {}
    for key in sorted(agg):
        cs = agg[key]
        n = max(len(v) for v in cs.values())
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        groups[key] = avg
        print(f"== {key[0]}  grid {key[1]}  ({n} dispatches; VGPR {meta[key]['vgpr']}, LDS {meta[key]['lds']} B)")
        for c in sorted(avg):
            print(f"   {c:32s} {avg[c]:18.1f}")
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            print(f"   {'HBM bytes (2*FETCH+WRITE)*1024':32s} {(2 * avg['FETCH_SIZE'] + avg['WRITE_SIZE']) * 1024:18.0f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
            # GRBM_GUI_ACTIVE sums over 8 XCDs; MFMA busy cycles sum over 1024 SIMDs' pipes (4 per CU x 256)
            print(f"   {'matrix pipe busy (of GUI active)':32s} {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg['GRBM_GUI_ACTIVE'] / 8 * 1024):18.3f}")
        if "SQ_INSTS_LDS" in avg and avg["SQ_INSTS_LDS"] > 0 and "SQ_LDS_BANK_CONFLICT" in avg:
            print(f"   {'LDS conflict cycles per LDS inst':32s} {avg['SQ_LDS_BANK_CONFLICT'] / avg['SQ_INSTS_LDS']:18.3f}")
    traffic = {}
    for op, rx, rank in op_table(cfg, P, "--conv-dtype bf16" in a.bench_args):
        cands = sorted((k for k in groups if re.search(rx, k[0])), key=lambda k: -k[1])
        if len(cands) <= rank:
            continue
        key = cands[rank]
        avg = groups[key]
        if "FETCH_SIZE" not in avg or "WRITE_SIZE" not in avg:
            continue
        src = source_of(key[0])
        traffic[op] = {"bytes": int((2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024),
                       "fetch_kib": avg["FETCH_SIZE"], "write_kib": avg["WRITE_SIZE"],
                       **({"valu_insts": int(avg["SQ_INSTS_VALU"])} if "SQ_INSTS_VALU" in avg else {}),
                       **({"ta_busy_over_gui_active": avg["TA_BUSY_avr"] / avg["GRBM_GUI_ACTIVE"]} if avg.get("GRBM_GUI_ACTIVE") and "TA_BUSY_avr" in avg else {}),   # (both as rocprofv3 reports them)
                       "kernel": key[0], "grid": key[1], "source": src,
                       "source_sha256": sha256(os.path.join(CSRC, src)) if src else None}
    doc = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes over bench.py itself (tools/pmc_bench.sh: separate "
                       "--pmc runs, no tracing): (2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950 FETCH_SIZE x2 correction for "
                       "16-B/lane coalesced reads applied (MI355X_MICROARCH.md, HBM section). Every entry names the kernel "
                       "instance it was measured on and the sha256 of that kernel's source file; bench.py reports "
                       "traffic: null for an entry whose source has changed since.",
           "bench_args": a.bench_args, cfg: traffic}
    with open(os.path.join(a.out, "traffic.json"), "w") as fh:
        json.dump(doc, fh, indent=1)
    print("# traffic.json:", json.dumps({k: v["bytes"] for k, v in traffic.items()}))


if __name__ == "__main__":
    main()
