"""D2H bandwidth of this box from page-locked memory: one copy vs the same bytes split over several streams (copy engines)."""
import time, torch
n = 10 * 1024 * 1024
d = torch.empty(n, dtype=torch.uint8, device='cuda')
h = torch.empty(n, dtype=torch.uint8).pin_memory()
s1, s2, s3, s4 = (torch.cuda.Stream() for _ in range(4))
def one():
    h.copy_(d, non_blocking=True); torch.cuda.synchronize()
def parts(k, streams):
    c = n // k
    for i in range(k):
        with torch.cuda.stream(streams[i]):
            h[i * c:(i + 1) * c].copy_(d[i * c:(i + 1) * c], non_blocking=True)
    torch.cuda.synchronize()
for name, f in (('one copy', one), ('two halves, two streams', lambda: parts(2, [s1, s2])), ('four quarters, four streams', lambda: parts(4, [s1, s2, s3, s4]))):
    for _ in range(5): f()
    t = time.perf_counter()
    for _ in range(50): f()
    dt = (time.perf_counter() - t) / 50
    print('%s: %.1f us, %.1f GB/s' % (name, dt * 1e6, n / dt / 1e9))
