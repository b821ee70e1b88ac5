import sys, torch
sys.path.insert(0, '.')
import se3conv3d_amd as amd
from se3conv3d_amd import workloads as W
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
levels = W.build_stack(W.WORKLOADS["headline"], dev, 0)
lv = levels[3]
cap = int(lv["e"] * 1.25) + 64
def build():
    nb = amd.pc.BQNeighborhood(lv["pc"], lv["pc"], lv["r"], p_capacity=cap)
    amd.layers._geometry_of(lv["pc"], lv["pc"], nb).transpose()
for _ in range(3): build()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    build()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=50, max_src_column_width=90))
