import torch, time
x = torch.randn(1,3,256,320)
torch.cuda.init(); torch.zeros(1, device="cuda")
for n in (16, 1024, 1<<18, 1<<20):
    t = torch.randn(n)
    for rep in range(3):
        t0=time.perf_counter()
        for _ in range(20):
            p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True); p.copy_(t); d = p.to("cuda", non_blocking=True)
        torch.cuda.synchronize()
        print(n*4, "bytes: %.3f ms per upload" % ((time.perf_counter()-t0)*1e3/20))
p = torch.empty(1<<20, pin_memory=True)
t = torch.randn(1<<20)
t0=time.perf_counter()
for _ in range(20):
    p.copy_(t); d = p.to("cuda", non_blocking=True)
torch.cuda.synchronize()
print("reused pinned 4MB: %.3f ms" % ((time.perf_counter()-t0)*1e3/20))
t0=time.perf_counter()
for _ in range(20):
    d = t.to("cuda")
torch.cuda.synchronize()
print("pageable 4MB .to: %.3f ms" % ((time.perf_counter()-t0)*1e3/20))
