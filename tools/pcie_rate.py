"""Raw device -> pinned host copy rate of this box (what bounds the end-to-end path): torch pinned tensors, 256 MiB copies."""
import time, torch
n = 256 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda")
h = torch.empty(n, dtype=torch.uint8).pin_memory()
for mode in ("one stream", "two streams, halves"):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter()
        if mode == "one stream":
            h.copy_(d, non_blocking=True)
        else:
            s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
            with torch.cuda.stream(s1):
                h[: n // 2].copy_(d[: n // 2], non_blocking=True)
            with torch.cuda.stream(s2):
                h[n // 2:].copy_(d[n // 2:], non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("D2H pinned, %s: %.1f GB/s" % (mode, n / best / 1e9), flush=True)
best = 1e9
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print("H2D pinned: %.1f GB/s" % (n / best / 1e9))
