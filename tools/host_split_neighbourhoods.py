import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import se3conv3d_amd as amd
from se3conv3d_amd import workloads as W, _lib
dev = torch.device("cuda:0")
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
pts, bid = W.faust_raw_batch(dev)
clouds = W.faust_clouds(pts, bid)
calls = W.faust_network_calls(os.path.join(ROOT, "tests", "golden", "network_faust_calls.npz"))
nbhs = W.faust_neighbourhoods(clouds, calls)
caps = {k: int(nb.num_edges() * 1.25) + 64 for k, nb in nbhs.items()}
lib = _lib.load()
acc = {"t": 0.0, "n": 0}
for name in ("se3_ball_query_bounded_shared", "se3_ball_query_bounded", "se3_ball_query_grid_from_box"):
    fn = getattr(lib, name)
    def wrap(*a, _fn=fn):
        t0 = time.perf_counter(); r = _fn(*a); acc["t"] += time.perf_counter() - t0; acc["n"] += 1; return r
    setattr(lib, name, wrap)
def fresh():
    for c in clouds: amd.ops.forget_source_grids(c)
    return W.faust_neighbourhoods(clouds, calls, caps)
for _ in range(5): fresh()
torch.cuda.synchronize(); acc["t"] = 0.0; acc["n"] = 0
t0 = time.perf_counter()
for _ in range(20): fresh()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 20
print(f"neighbourhoods per step: {tot*1e3:.3f} ms wall; inside the library's C calls {acc['t']/20*1e3:.3f} ms ({acc['n']//20} calls); Python around them {(tot - acc['t']/20)*1e3:.3f} ms")
