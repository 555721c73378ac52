"""A/B aid (GPU box): lr_pca_bwd_coef_f32 at C3 (B = 8, L = 56, M = 3 * 256^3) against the number of m-blocks."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from liftreg_amd import ops_bwd
dev = torch.device("cuda:0")
B, L, M = 8, 56, 3 * 256 ** 3
g = torch.randn(B, M, device=dev)
basis = torch.randn(L, M, device=dev)
for nblk in (128, 256, 512, 1024):
    for _ in range(3): ops_bwd.pca_bwd_coef(g, basis, nblk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops_bwd.pca_bwd_coef(g, basis, nblk)
    e1.record(); torch.cuda.synchronize()
    print(nblk, round(e0.elapsed_time(e1) / 10, 3), "ms")
