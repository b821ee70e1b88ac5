import sys, time, cProfile, pstats, io, torch
sys.path.insert(0, '.')
import bench
from se3conv3d_amd import workloads as W
dev = torch.device("cuda:0")
levels = W.build_stack(W.WORKLOADS["headline"], dev, 0)
lv = levels[3]
for _ in range(20): bench.step([lv])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): bench.step([lv])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
