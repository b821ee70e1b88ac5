import torch, time
x = torch.empty(268435456, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
gb = x.numel()*4/1e9
print("fill  %.3f ms  %.2f TB/s" % (t(lambda: x.fill_(1.0))*1e3, gb/t(lambda: x.fill_(1.0))/1e3))
print("copy  %.3f ms  %.2f TB/s (r+w)" % (t(lambda: y.copy_(x))*1e3, 2*gb/t(lambda: y.copy_(x))/1e3))
print("sum   %.3f ms  %.2f TB/s" % (t(lambda: x.sum())*1e3, gb/t(lambda: x.sum())/1e3))
