import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3conv3d_amd as amd
from oracle import se3conv_oracle as O
import bench
dev = torch.device("cuda:0")
levels = bench.build_stack(amd, O, dev, 0)
lv = levels[3]
def run():
    for _ in range(200):
        nb = amd.pc.BQNeighborhood(lv["pc"], lv["pc"], lv["r"])
        amd.layers._geometry_of(lv["pc"], lv["pc"], nb).transpose()
    torch.cuda.synchronize()
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
