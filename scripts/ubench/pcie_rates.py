import torch, time
d = torch.empty(800*1024*1024, dtype=torch.uint8, device="cuda")
h = torch.empty(800*1024*1024, dtype=torch.uint8).pin_memory()
for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
    fn(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    print(name, "%.1f GB/s" % (3*h.numel()/(time.perf_counter()-t0)/1e9))
# chunks of 16.6 MB
n=48; ch=16588800
hs=[torch.empty(ch,dtype=torch.uint8).pin_memory() for _ in range(n)]
ds=[torch.empty(ch,dtype=torch.uint8,device="cuda") for _ in range(n)]
for i in range(n): hs[i].copy_(ds[i], non_blocking=True)
torch.cuda.synchronize()
t0=time.perf_counter()
for i in range(n): hs[i].copy_(ds[i], non_blocking=True)
torch.cuda.synchronize()
print("D2H 16.6 MB chunks, one stream: %.1f GB/s" % (n*ch/(time.perf_counter()-t0)/1e9))
s2=[torch.cuda.Stream() for _ in range(2)]
t0=time.perf_counter()
for i in range(n):
    with torch.cuda.stream(s2[i&1]): hs[i].copy_(ds[i], non_blocking=True)
torch.cuda.synchronize()
print("D2H 16.6 MB chunks, two streams: %.1f GB/s" % (n*ch/(time.perf_counter()-t0)/1e9))
# both directions at once
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
t0=time.perf_counter()
for _ in range(3):
    with torch.cuda.stream(sa): d.copy_(h, non_blocking=True)
    with torch.cuda.stream(sb): hs[0].copy_(ds[0], non_blocking=True); [hs[i].copy_(ds[i], non_blocking=True) for i in range(1,n)]
torch.cuda.synchronize()
dt=time.perf_counter()-t0
print("bidirectional: H2D %.1f GB/s + D2H %.1f GB/s" % (3*h.numel()/dt/1e9, 3*n*ch/dt/1e9))
