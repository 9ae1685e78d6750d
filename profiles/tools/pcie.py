"""profiles/tools/pcie.py: pinned-memory transfer rates of the box (what bounds the host-pointer entry points)"""
import time
import torch
n = 1 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, a, b in (("H2D", d, h), ("D2H", h, d)):
    a.copy_(b, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        a.copy_(b, non_blocking=True)
    torch.cuda.synchronize()
    print(f"{name}: {5 * n / (time.perf_counter() - t) / 1e9:.1f} GB/s (1 GiB, pinned)")
