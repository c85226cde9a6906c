"""Practical HBM bandwidth of this box: device-to-device copy and a read-only reduction at several sizes (dev tool)."""
import torch
dev = "cuda:0"
for mb in (64, 303, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    for name, fn, bytes_ in (("copy (r+w)", lambda: y.copy_(x), 8 * n), ("sum (read)", lambda: x.sum(), 4 * n), ("fill (write)", lambda: y.fill_(1.0), 4 * n)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{mb:5d} MB  {name:12s} {ms*1e3:8.1f} us  {bytes_/ms/1e9:7.2f} TB/s", flush=True)
