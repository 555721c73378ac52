"""CPU simulation (no GPU): error of a convolution whose fp32 operands are TWO-way fp16 splits (x = h0 + h1 + residual <= 2^-22 |x|;
products h0 w0, h0 w1, h1 w0, each exact in fp32, fp32 accumulation) against an fp64 evaluation, beside the exact THREE-way bf16
split of conv01_fused.hip (six products) and a plain fp32 convolution.  Evidence for DESIGN.md §9 (3 MFMAs per K block instead of 6)."""
import torch
import torch.nn.functional as F


def split_f16(t, n):
    parts, r = [], t.clone()
    for _ in range(n):
        h = r.half().float()
        parts.append(h)
        r = r - h
    return parts


def split_bf16(t, n):
    parts, r = [], t.clone()
    for _ in range(n):
        h = r.bfloat16().float()
        parts.append(h)
        r = r - h
    return parts


def conv32(x, w, stride):
    return F.conv3d(x, w, None, stride=stride, padding=1)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for name, cin, cout, stride in (("block 0 (3 -> 16, s1)", 3, 16, 1), ("block 1 (16 -> 32, s2)", 16, 32, 2)):
        x = torch.randn(1, cin, 24, 24, 24)
        if cin == 3:
            x[:, 0] = x[:, 0].abs() * 0.3
        else:
            x = F.leaky_relu(x, 0.2)
        w = torch.randn(cout, cin, 3, 3, 3) * (2.0 / (27 * cin)) ** 0.5
        ref = F.conv3d(x.double(), w.double(), None, stride=stride, padding=1)
        scale = float(ref.abs().max())
        plain = conv32(x, w, stride).double()
        xs, ws = split_f16(x, 2), split_f16(w, 2)
        y16 = (conv32(xs[0], ws[0], stride).double() + conv32(xs[0], ws[1], stride).double() + conv32(xs[1], ws[0], stride).double())
        xb, wb = split_bf16(x, 3), split_bf16(w, 3)
        yb = sum(conv32(xb[i], wb[j], stride).double() for i in range(3) for j in range(3) if i + j <= 2)

        def err(y):
            return float((y - ref).pow(2).mean().sqrt()), float((y - ref).abs().max())
        print(f"{name}: scale {scale:.2f}")
        for tag, y in (("plain fp32 convolution", plain), ("3-way bf16 split, 6 products (today)", yb), ("2-way fp16 split, 3 products", y16)):
            r, m = err(y)
            print(f"   {tag:40s} rms {r:.3e}  max {m:.3e}")


if __name__ == "__main__":
    main()
