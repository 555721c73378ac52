#!/bin/bash
# Package power / sclk (rocm-smi, 0.3 s samples) under the bench lines of DESIGN.md §6·7:  bash tools/power_round.sh <out-dir>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; mkdir -p "$O"; cd "$R"
run() { n=$1; shift; bash tools/power_watch.sh "$O/$n.txt" "$@" > /dev/null 2>&1; }
run infer_c3_f32 python3 bench.py --no-cpu-baseline --no-drr --steps 300
run infer_c3_f32_conv0_split python3 bench.py --no-cpu-baseline --no-drr --steps 300 --conv0-split
run infer_c3_bf16 python3 bench.py --no-cpu-baseline --no-drr --steps 400 --conv-dtype bf16
run infer_c4_bf16 python3 bench.py --no-cpu-baseline --no-drr --steps 400 --config c4 --conv-dtype bf16
run train_c3_f32 python3 tools/train_bench.py --config c3 --steps 100 --no-kernel-table
run train_c5_bf16 python3 tools/train_bench.py --config c5 --conv-dtype bf16 --grad-dtype bf16 --steps 60 --no-kernel-table
python3 - "$O" <<'PY'
import glob, os, re, sys, statistics as st
o = sys.argv[1]
print("# tools/power_round.sh: rocm-smi samples (0.3 s) while the line runs; steady state = samples above 80 % of the run's maximum power")
print("# line | samples | package power W (median, max) | sclk MHz (median, min..max) | junction C (max) | result")
for f in sorted(glob.glob(o + "/*.txt")):
    if f.endswith("summary.txt"):
        continue
    rows = []
    for l in open(f):
        p = re.search(r"Power \(W\): ([\d.]+)", l); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz", l); t = re.search(r"junction\) \(C\): ([\d.]+)", l)
        if p and s:
            rows.append((float(p.group(1)), int(s.group(1)), float(t.group(1)) if t else 0.0))
    if not rows:
        continue
    pmax = max(r[0] for r in rows)
    ss = [r for r in rows if r[0] >= 0.8 * pmax]
    res = ""
    try:
        last = open(f + ".cmd.log").read().strip().splitlines()
        js = [l for l in last if l.startswith("{")]
        m = re.search(r'"value": ([\d.]+)', js[0] if "train" in f else js[-1]); res = m.group(1) if m else ""
    except Exception:
        pass
    print(f"{os.path.basename(f)[:-4]} | {len(ss)} | {st.median(r[0] for r in ss):.0f}, {pmax:.0f} | {st.median(r[1] for r in ss):.0f}, {min(r[1] for r in ss)}..{max(r[1] for r in ss)} | {max(r[2] for r in ss):.0f} | {res}")
PY
