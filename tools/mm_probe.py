import torch, time
dev = torch.device("cuda", 0)
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for M in (131072, 18414, 2652):
    for (K, N) in ((64, 128), (128, 64), (64, 64)):
        h = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        Wt = W.t().contiguous()
        g = torch.randn(M, N, device=dev)
        out = torch.empty(M, N, device=dev)
        r = {
            "F.linear(h,W)": t(lambda: torch.nn.functional.linear(h, W)),
            "F.linear(h,W,b)": t(lambda: torch.nn.functional.linear(h, W, b)),
            "mm(h,Wt)": t(lambda: torch.mm(h, Wt)),
            "addmm(b,h,Wt)": t(lambda: torch.addmm(b, h, Wt)),
            "mm(h,W.t())": t(lambda: torch.mm(h, W.t())),
            "dgrad mm(g,W)": t(lambda: torch.mm(g, W)),
            "wgrad mm(g.t(),h)": t(lambda: torch.mm(g.t(), h)),
        }
        print(f"M={M} K={K} N={N}: " + "  ".join(f"{k} {v:.0f}us" for k, v in r.items()), flush=True)
