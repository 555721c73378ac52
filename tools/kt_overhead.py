import os, sys, time, torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/liftreg_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
from liftreg_amd import ops
from liftreg_amd.layers.losses import NCCLoss
from liftreg_amd.models.LiftRegDeformSubspaceBackproj import model
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["c3"]; n, P, L = cfg["n"], cfg["P"], cfg["L"]
torch.manual_seed(2021)
net = model([n, n, n], {"drr_feature_num": P, "latent_dim": L, "pca_path": "synthetic:2021"}).to(dev).eval()
inp = bench.synth_inputs(cfg, dev); sim = NCCLoss(check_nan=False)
def step():
    out = net(inp); return sim(out["warped"], out["target"])
with torch.no_grad():
    t_end = time.perf_counter() + 2.0
    while time.perf_counter() < t_end:
        step(); torch.cuda.synchronize()
    for rep in range(4):
        for timed in (False, True):
            for _ in range(5): step()
            torch.cuda.synchronize()
            if timed:
                with ops.kernel_timer() as kt:
                    t0 = time.perf_counter()
                    for _ in range(50): step()
                    torch.cuda.synchronize(); t = time.perf_counter() - t0
            else:
                t0 = time.perf_counter()
                for _ in range(50): step()
                torch.cuda.synchronize(); t = time.perf_counter() - t0
            print("kernel_timer" if timed else "no timer    ", round(t / 50 * 1e3, 3), "ms/step", round(8 * 50 / t, 1), "reg/s", flush=True)
