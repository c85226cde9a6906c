"""What a plain stream gets on this box: copy (read + write), read-only reduce, write-only fill at sizes the layers move.  dev tool"""
import torch
dev = 'cuda:0'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for mb in (38, 151, 303, 606, 2424):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    tc = t(lambda: y.copy_(x)); tr = t(lambda: x.sum()); tw = t(lambda: y.fill_(1.0)); ta = t(lambda: torch.add(x, 1.0, out=y))
    print(f"{mb:5d} MB: copy {2*mb/tc/1e3:5.2f} TB/s ({tc*1e3:6.1f} us)  add {2*mb/ta/1e3:5.2f} TB/s  read-only sum {mb/tr/1e3:5.2f} TB/s  fill {mb/tw/1e3:5.2f} TB/s")
