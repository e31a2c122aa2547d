"""Practical HBM rates on the box with plain torch streaming ops (context for the roofline fractions of the HBM-bound kernels)."""
import torch, time
n = 1 << 30
x = torch.randn(n, device="cuda")   # 4 GiB
x2 = torch.randn(n, device="cuda")
y = torch.empty_like(x)
h = torch.empty(n, device="cuda", dtype=torch.float16)
def t(f, k=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
def show(name, rd, wr, s):
    print("%-28s read %5.2f  write %5.2f  total %5.2f TB/s" % (name, rd / s / 1e12, wr / s / 1e12, (rd + wr) / s / 1e12), flush=True)
B = n * 4
show("sum (read only)", B, 0, t(lambda: x.sum()))
show("max (read only)", B, 0, t(lambda: x.max()))
show("fill (write only)", 0, B, t(lambda: y.fill_(1.0)))
show("copy", B, B, t(lambda: y.copy_(x)))
show("mul scalar (out=)", B, B, t(lambda: torch.mul(x, 2.0, out=y)))
show("add two (out=)", 2 * B, B, t(lambda: torch.add(x, x2, out=y)))
show("to half (out=)", B, B // 2, t(lambda: h.copy_(x)))
show("dot (read only, 2 streams)", 2 * B, 0, t(lambda: torch.dot(x, x2)))
