"""Calibrates the achievable HBM copy bandwidth on this GPU with torch's own copy kernel (20 GB of traffic)."""
import torch, time
n = 10_000_000_000
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
a.random_(0, 255)
for _ in range(2): b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("torch copy 10 GB: %.3f ms  -> %.2f TB/s (read+write)" % (ms, 2 * n / ms / 1e9))
x = a.view(torch.int64)
for _ in range(2): s = x.sum()
torch.cuda.synchronize(); e0.record()
for _ in range(10): s = x.sum()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("torch sum (read only) 10 GB: %.3f ms -> %.2f TB/s" % (ms, n / ms / 1e9))
