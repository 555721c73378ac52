#!/usr/bin/env python3
"""Which torch ops the slab-sharded forward issues beside the library's kernels (copies, fills, elementwise): torch.profiler over
one forward of 8 virtual ranks.  Development aid for liftreg_amd/parallel.py."""
import os, sys, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import shard_bench as sb
from liftreg_amd import parallel as par
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256); ap.add_argument("--world", type=int, default=8); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--views", type=int, default=2); ap.add_argument("--conv-dtype", default="fp32")
a = ap.parse_args()
dev = torch.device("cuda:0")
net, inp = sb._build(a, dev)
sh = par.SlabShardedRegistration(net, par.LocalComm(a.world))
with torch.no_grad():
    sh.forward([inp] * a.world); sh.forward([inp] * a.world)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        sh.forward([inp] * a.world)
        torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60, max_src_column_width=90))
