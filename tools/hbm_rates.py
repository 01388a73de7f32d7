"""Read-only / write-only / copy rates of HBM on this box (torch kernels over 1 GiB, past the 256 MB Infinity Cache): what the
write side of the synthesis pass can expect.  python tools/hbm_rates.py"""
import torch, time
n = 1 << 28  # floats: 1 GiB
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
gb = n * 4 / 1e12
tw = t(lambda: a.fill_(1.0)); print("write-only  %.2f TB/s" % (gb / tw))
tr = t(lambda: a.sum()); print("read-only   %.2f TB/s" % (gb / tr))
tc = t(lambda: b.copy_(a)); print("copy        %.2f TB/s (read + write)" % (2 * gb / tc))
tm = t(lambda: torch.add(a, 1.0, out=b)); print("add out     %.2f TB/s (read + write)" % (2 * gb / tm))
# a 2:3 read:write mix like the synthesis pass (38 MB read, 50 MB written): two reads fused with three writes is not a torch op;
# approximated by interleaving kernels on two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
c = torch.empty(n, dtype=torch.float32, device="cuda")
def mix():
    with torch.cuda.stream(s1): a.fill_(2.0)
    with torch.cuda.stream(s2): c.sum()
for _ in range(3): mix()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): mix()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("fill || sum on two streams: %.2f TB/s combined" % (2 * gb / dt))
