import torch, time
n = 670_794_900  # floats = 2.68 GB
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: y.copy_(x)), ("add", lambda: torch.add(x, 1.0, out=y))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    moved = n * 4 * (1 if name == "fill" else 2)
    print(f"{name}: {dt*1e3:.3f} ms  {moved/dt/1e12:.2f} TB/s")
