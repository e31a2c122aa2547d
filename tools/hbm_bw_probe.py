import torch, time
x = torch.randn(1 << 30, device="cuda")   # 4 GiB
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
s = t(lambda: x.sum())
print("sum   read  %.2f TB/s" % (x.numel() * 4 / s / 1e12))
y = torch.empty_like(x)
s = t(lambda: y.copy_(x))
print("copy  r+w   %.2f TB/s" % (2 * x.numel() * 4 / s / 1e12))
s = t(lambda: y.fill_(1.0))
print("fill  write %.2f TB/s" % (x.numel() * 4 / s / 1e12))
s = t(lambda: torch.mul(x, 2.0, out=y))
print("scale r+w   %.2f TB/s" % (2 * x.numel() * 4 / s / 1e12))
