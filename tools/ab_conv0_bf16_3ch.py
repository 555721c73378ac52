import os, sys, torch
sys.path.insert(0, "/root/repo")
from liftreg_amd import ops
dev = torch.device("cuda:0")
B, C, n = 8, 3, 256
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.rand((B, C, n, n, n), generator=g, device=dev) * 2 - 1
w = torch.randn((16, C, 3, 3, 3), generator=g, device=dev) * 0.1
b = torch.randn((16,), generator=g, device=dev) * 0.1
pk = ops.conv3d_pack_weights_bf16_planar(w)
out = torch.empty((B, n, n, n, 16), dtype=torch.bfloat16, device=dev)
def run(env):
    for k in ("LIFTREG_CONV0_BF16_CL","LIFTREG_C0CL_SHAPE"): os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(3): ops.conv3d_first_bf16(x, w, b, out_layout=ops.LAYOUT_BF16_NDHWC_HPS, packed=pk, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.conv3d_first_bf16(x, w, b, out_layout=ops.LAYOUT_BF16_NDHWC_HPS, packed=pk, out=out)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10, out.clone()
t0, o0 = run({})
t1, o1 = run({"LIFTREG_CONV0_BF16_CL": "1"})
t2, o2 = run({"LIFTREG_CONV0_BF16_CL": "1", "LIFTREG_C0CL_SHAPE": "44"})
same = (o0 == o1).float().mean().item()
print(f"3-channel first block bf16 at C3: channel-pass kernel {t0:.3f} ms, z-march 1x16 {t1:.3f} ms, 4x4 {t2:.3f} ms; identical outputs {same:.5f}, max diff {(o0.float()-o1.float()).abs().max().item():.3e}")
