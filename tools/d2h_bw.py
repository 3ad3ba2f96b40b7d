import torch, time
x = torch.empty(1024*1024*1024//4, dtype=torch.float32, device="cuda")  # 1 GiB
h = torch.empty_like(x, device="cpu").pin_memory()
for _ in range(2): h.copy_(x, non_blocking=True); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5): h.copy_(x, non_blocking=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print(f"D2H pinned 1 GiB: {1.0737/dt:.1f} GB/s")
s2=torch.cuda.Stream(); h2=torch.empty_like(h).pin_memory(); x2=torch.empty_like(x)
t0=time.perf_counter()
for _ in range(5):
    h.copy_(x, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(x2, non_blocking=True)
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
print(f"two concurrent D2H streams: {2*1.0737/dt:.1f} GB/s total")
