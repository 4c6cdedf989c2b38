import torch, time
x = torch.randn((65536, 3120), device='cuda')
y = torch.empty_like(x)
for name, fn, nbytes in [("sum", lambda: x.sum(), x.numel()*4), ("copy", lambda: y.copy_(x), x.numel()*8), ("rowsum", lambda: x.sum(dim=1), x.numel()*4)]:
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
    print(name, "%.3f ms  %.2f TB/s" % (dt*1e3, nbytes/dt/1e12))
