"""Copy one tools/gpu_round.sh output directory (gpurun_out/<tag>) into profiles/ under the round's names and merge the
two PMC traffic files into profiles/traffic.json (development aid):  python tools/collect_round.py r03_final2 r03"""
import glob, json, os, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out", tag) + "/", os.path.join(R, "profiles") + "/"
a, b = json.load(open(O + "pmc/traffic.json")), json.load(open(O + "pmc_c4/traffic.json"))
old = json.load(open(P + "traffic.json"))
c3 = dict(a["c3"])
if os.path.exists(O + "pmc_bf16/traffic.json"):   # the bf16 variant of c3: other op names, same section
    for k, v in json.load(open(O + "pmc_bf16/traffic.json"))["c3"].items():
        c3.setdefault(k, v)
    shutil.copy(O + "pmc_bf16/summary.txt", P + rnd + "_pmc_bench_bf16_summary.txt")
json.dump({"_comment": old["_comment"], "bench_args": a["bench_args"], "c3": c3, "c4": b["c4"], "bench_args_c4": b["bench_args"]},
          open(P + "traffic.json", "w"), indent=1)
names = {"bench.json": "bench.json", "bench_bf16.json": "bench_bf16.json", "bench_c1_graph.json": "bench_c1_graph.json",
         "bench_c2_graph.json": "bench_c2_graph.json", "bench_c4_bf16.json": "bench_c4_bf16.json",
         "bench_c4_bf16_feature_volume_path.json": "bench_c4_bf16_feature_volume_path.json",
         "bench_c4_bf16_slab_x1.json": "bench_c4_bf16_slab_x1.json", "bench_slab_x1.json": "bench_slab_x1.json",
         "bench_conv0_split.json": "bench_conv0_split.json", "pmc/summary.txt": "pmc_bench_summary.txt",
         "pmc_c4/summary.txt": "pmc_bench_c4_bf16_summary.txt", "train_modes.jsonl": "train_modes.jsonl",
         "train_c3_kernels.jsonl": "train_c3_kernels.jsonl", "train_c5_bf16_kernels.jsonl": "train_c5_bf16_kernels.jsonl",
         "shard_bench.jsonl": "shard_bench.jsonl", "ab_conv0_split.txt": "ab_conv0_split.txt",
         "ab_train_algebra.txt": "ab_train_algebra.txt", "power_trace.txt": "power_trace.txt"}
for s, d in names.items():
    if os.path.exists(O + s):
        shutil.copy(O + s, P + rnd + "_" + d)
for s, d in (("fwd", "kernel_stats.csv"), ("fwd_c4", "kernel_stats_c4_bf16.csv")):
    f = glob.glob(O + s + "/**/*kernel_stats.csv", recursive=True)
    if f:
        shutil.copy(f[0], P + rnd + "_" + d)
open(P + rnd + "_pytest_gpu_tail.txt", "w").write("".join(open(O + "pytest.log").readlines()[-25:]))
for f in sorted(glob.glob(P + rnd + "_bench*.json")):
    d = json.loads(open(f).readline())
    print(os.path.basename(f), round(d["value"], 1), "reg/s", round(d["ms_per_step"], 3), "ms", "stale:", d.get("traffic_stale"))
