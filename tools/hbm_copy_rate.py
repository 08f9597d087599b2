import torch, time
n = 152_158_273
a = torch.empty(n, dtype=torch.int64, device="cuda"); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): b.copy_(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("copy of %.2f GB: %.3f ms -> %.2f TB/s read+write" % (n * 8 / 1e9, dt * 1e3, 2 * n * 8 / dt / 1e12))
t0 = time.perf_counter()
for _ in range(20): b.fill_(1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("fill: %.3f ms -> %.2f TB/s write" % (dt * 1e3, n * 8 / dt / 1e12))
t0 = time.perf_counter()
for _ in range(20): s = a.sum()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("sum: %.3f ms -> %.2f TB/s read" % (dt * 1e3, n * 8 / dt / 1e12))
