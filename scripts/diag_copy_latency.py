"""Which host<->device operations stall when GPU work is queued?  (dev diagnostic for this ROCm stack)"""
import time, torch
dev = torch.device('cuda', 0)
a = torch.randn(5, 16, device=dev)
big = torch.randn(64, 1024, 1024, device=dev)
h = torch.randn(5, 16); hp = h.pin_memory(); d = torch.empty(5, 16, device=dev); pin = torch.empty(5, 16).pin_memory()
def busy():
    for _ in range(4): big.mul_(1.0001)
def t(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
print('queued work alone (4 x mul_ 256MB)      : %8.1f us' % t(busy))
def f1(): busy(); a.cpu()
print('work + .cpu()                            : %8.1f us' % t(f1))
def f2(): busy(); pin.copy_(a, non_blocking=True); torch.cuda.current_stream().synchronize()
print('work + pinned D2H + stream.synchronize() : %8.1f us' % t(f2))
def f3(): busy(); torch.cuda.current_stream().synchronize()
print('work + stream.synchronize()              : %8.1f us' % t(f3))
def f4(): busy(); torch.cuda.synchronize()
print('work + torch.cuda.synchronize()          : %8.1f us' % t(f4))
def f5(): busy(); e = torch.cuda.Event(); e.record(); e.synchronize()
print('work + event.synchronize()               : %8.1f us' % t(f5))
def f5b():
    busy(); e = torch.cuda.Event(); e.record()
    while not e.query(): pass
print('work + event.query() spin                : %8.1f us' % t(f5b))
def f6(): busy(); h.to(dev, non_blocking=True)
print('work + pageable H2D non_blocking         : %8.1f us' % t(f6))
def f7(): busy(); h.to(dev)
print('work + pageable H2D blocking             : %8.1f us' % t(f7))
def f8(): busy(); d.copy_(hp, non_blocking=True)
print('work + pinned H2D non_blocking           : %8.1f us' % t(f8))
def f9(): busy(); pin.copy_(a, non_blocking=True); e = torch.cuda.Event(); e.record()
print('work + pinned D2H (no wait)              : %8.1f us' % t(f9))
def f10():
    busy(); pin.copy_(a, non_blocking=True); e = torch.cuda.Event(); e.record()
    while not e.query(): pass
print('work + pinned D2H + event.query() spin   : %8.1f us' % t(f10))
