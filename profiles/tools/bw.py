# achievable HBM bandwidth of this box: device-to-device copy (1 read : 1 write) and fill (write only) of 4 GiB, read-only sum
import torch, time
n = 1 << 30
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
a.fill_(1.0); torch.cuda.synchronize()
def t(f, k=10):
    f(); torch.cuda.synchronize(); s = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - s) / k
print("copy  %.2f TB/s" % (2 * 4 * n / t(lambda: b.copy_(a)) / 1e12))
print("fill  %.2f TB/s" % (4 * n / t(lambda: b.fill_(2.0)) / 1e12))
print("sum   %.2f TB/s" % (4 * n / t(lambda: a.sum()) / 1e12))
print("add   %.2f TB/s (2 reads : 1 write)" % (3 * 4 * n / t(lambda: torch.add(a, b, out=b)) / 1e12))
