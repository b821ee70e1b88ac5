#!/usr/bin/env python3
"""Benchmark of the PNEConvLayerRotEquiv hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU; scenes are sharded by rank, the
     data path has no collective -- only the timing barrier and a MAX all-reduce of the wall time.)

Workload (BASELINE.json metric "Mpoints/sec fwd+bwd PNEConvLayerRotEquiv (N=64k,k=32,F=2)"):
one synthetic cloud per rank, N0 = 65 536 points ~ U[0,1)^3, F = 2 random frames per point,
C = 64 -> 64 channels, K = 32 basis functions, radius for mean degree k = 32; a 4-level stack = one
same-level convolution per level of a grid-subsampled hierarchy (cell doubling, radius = 2 x cell
like tasks/SemSeg/seg_models.py:29-33).  One step = forward + backward (dX, dA, dbeta, dW) of all
four levels, neighbourhoods prebuilt ("conv-only").  value = N0 * n_gpus /
step time.  The JSON line also carries the single full-resolution layer rate, the roofline of the
dominant kernel (HIP events inside the library, on the launch stream) and a CPU baseline (the
oracle, timed on this box's host cores on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N0, K_DEG, FRAMES, CH, KB = 65536, 32, 2, 64, 32
# MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate; bf16 dense MFMA ~2.5 PF.
# In "bf16x3" every multiply costs 3 bf16 MFMA products, so frac <= 1/3 by construction there.
PEAK_MFMA_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0}
PEAK_HBM_GBPS = 8000.0          # HBM3E spec


def build_stack(amd, O, device, seed):
    torch.manual_seed(seed)
    cfg = {"pca": False, "n_frames": FRAMES, "fixed_axis": False}
    r0 = O.radius_for_degree(N0, K_DEG)
    pts = torch.rand(N0, 3, device=device)
    bid = torch.zeros(N0, dtype=torch.int32, device=device)
    pc0 = amd.pc.PointcloudRotEquiv(pts, bid, cfg)
    hier = amd.pc.PointHierarchyRotEquiv(pc0, 3, "grid_avg", grid_radii=[r0, 2 * r0, 4 * r0])
    radii = [r0, 2 * r0, 4 * r0, 8 * r0]  # radius = 2 x the cell that produced the level
    factory = amd.PNEConvLayerRotEquivFactory(9, KB, "mlp_gelu")
    levels = []
    for lvl, (pc, r) in enumerate(zip(hier.pcs_, radii)):
        nbh = hier.create_neighborhood(lvl, lvl, "ball_query", bq_radius=r)
        conv = factory.create_conv_layer(CH, CH).to(device)
        conv.norm_neigh_dist_.fill_(1.0 / r)
        conv.norm_num_neighs_.fill_(nbh.start_ids_.shape[0] / max(nbh.neighbors_.shape[0], 1))
        n = pc.pts_.shape[0]
        x = torch.randn(n * FRAMES, CH, device=device, requires_grad=True)
        g = torch.randn(n * FRAMES, CH, device=device)
        levels.append(dict(pc=pc, nbh=nbh, conv=conv, x=x, g=g, n=n, e=nbh.neighbors_.shape[0], r=r))
    return levels


def step(levels):
    for lv in levels:
        lv["x"].grad = None
        for p in lv["conv"].parameters():
            p.grad = None
        out = lv["conv"](p_pc_in=lv["pc"], p_pc_out=lv["pc"], p_in_features=lv["x"], p_neighborhood=lv["nbh"])
        out.backward(lv["g"])


class GraphedStep:
    """The step captured once into a HIP graph and replayed (launch-bound for the small levels
    otherwise: ~25 kernel launches + Python per level).  Inputs/outputs live in static buffers, so a
    replay recomputes exactly the same forward+backward on whatever the buffers hold."""

    def __init__(self, levels):
        self.levels = levels
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up on the side stream: lazy builds (transposed edge list), allocator
                step(levels)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            step(levels)

    def __call__(self):
        self.graph.replay()


def layer_flops(n, e):
    """Algorithmic FLOPs per stage of one layer (MLP counted with its bias row)."""
    ep = e * FRAMES * FRAMES
    rows = n * FRAMES
    dense = rows * 2 * CH * KB * CH
    edge = ep * (2 * 10 * KB + 2 * CH * KB)
    pg = ep * (2 * 10 * KB + 2 * CH * KB + 2 * 10 * KB)
    return {"edge_t_fwd": edge, "gemm_out": dense, "gemm_gradT": dense, "gemm_gradW": dense, "gemm_gradX": dense,
            "gemm_H": dense, "edge_t_transposed": edge, "edge_param_grad": pg, "edge_bwd": pg + ep * 2 * CH * KB}


def stage_bytes(n, e, t24=True):
    """HBM bytes each kernel of the (unfused) pipeline has to move at least: its gathered rows counted
    once per edge (uncached-gather model), its dense operands and results once.  t24: T and U are stored in the
    3-byte row format (bf16x3 path, C >= 64); grad_T always is packed 4-byte words."""
    rows = n * FRAMES
    g_bytes = 4 * rows * CH * KB  # grad_T: [rows, C, K] words
    t_bytes = (3 if t24 else 4) * rows * CH * KB  # T / U
    geom = 8 * e + 4 * n + 12 * 2 * n + 36 * 2 * rows
    w = 4 * CH * KB * CH
    # every point-edge touches its neighbour's F*C block once per pass; the two centre
    # frames of a point share that gather
    gather = geom + 4 * e * FRAMES * CH
    return {"edge_t_fwd": gather + t_bytes, "edge_t_transposed": gather + t_bytes, "edge_param_grad": gather + g_bytes,
            "gemm_out": t_bytes + w + 4 * rows * CH, "gemm_gradT": g_bytes + w + 4 * rows * CH,
            "gemm_gradX": t_bytes + w + 4 * rows * CH, "gemm_gradW": t_bytes + 4 * rows * CH + w,
            "gemm_H": g_bytes + w + 4 * rows * CH, "edge_bwd": gather + g_bytes + t_bytes,
            "prep": 3 * 8 * rows * CH + 2 * (48 + 64) * rows + 6 * w}


def layer_bytes(n, e):
    """Algorithmic HBM bytes of one layer fwd+bwd (uncached-gather model of SURVEY.md section 8d:
    every point-edge touches its neighbour's F_in*C_in block once per pass; no T, nothing E'-sized)."""
    rows = n * FRAMES
    geom = 8 * e + 4 * n + 12 * 2 * n + 36 * 2 * rows
    params = 4 * (10 * KB + CH * KB * CH)
    fwd = geom + 4 * e * FRAMES * CH + 4 * rows * CH + params
    bwd = geom + 2 * 4 * e * FRAMES * CH + 4 * rows * CH + 4 * rows * CH + 2 * params
    return fwd + bwd


def profile_level0(lib, lv, reps):
    lib.se3_profile_reset()
    lib.se3_profile_enable(1)
    for _ in range(reps):
        step([lv])
    torch.cuda.synchronize()
    lib.se3_profile_enable(0)
    buf = C.create_string_buffer(4096)
    lib.se3_profile_tags(buf, 4096)
    stages = {}
    for tag in buf.value.decode().split(","):
        if not tag:
            continue
        ms, cnt = C.c_double(0), C.c_int64(0)
        lib.se3_profile_read(tag.encode(), C.byref(ms), C.byref(cnt))
        stages[tag] = (ms.value / max(cnt.value, 1), cnt.value)
    lib.se3_profile_reset()
    return stages


def cpu_baseline(O):
    """The oracle (a port of the reference's Python path) on this box's host cores, bounded sample:
    one layer, same k / F / C / K, N = 4096 points."""
    n = 4096
    g = torch.Generator().manual_seed(0)
    pts = torch.rand(n, 3, generator=g)
    bid = torch.zeros(n, dtype=torch.int32)
    fr = O.random_frames(n, FRAMES, g)
    r = O.radius_for_degree(n, K_DEG)
    nb, ends = O.ball_query(pts, pts, bid, bid, r)
    a, b, w = O.init_parameters(9, CH, CH, KB, g)
    x = torch.randn(n * FRAMES, CH, generator=g)
    go = torch.randn(n * FRAMES, CH, generator=g)
    rho, nu = torch.tensor(1.0 / r), torch.tensor(n / nb.shape[0])
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.conv_forward_backward(pts, pts, fr, fr, nb, x, a, b, w, rho, nu, go)
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {"value": n / best / 1e6, "unit": "Mpoints/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (torch CPU) single layer fwd+bwd, N={n}, k~{nb.shape[0] / n:.1f}, F={FRAMES}, "
                      f"C={CH}, K={KB}; best of 2 after 1 warm-up ({best:.2f} s/step)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--precision", default=os.environ.get("SE3CONV_PRECISION", "bf16x3"), choices=["bf16x3", "fp32"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    import se3conv3d_amd as amd
    from oracle import se3conv_oracle as O  # cpu_baseline leg + radius helper only
    from se3conv3d_amd import _lib

    from se3conv3d_amd.sharding import gather_scene_results, job_throughput, shard_scenes

    lib = _lib.load()
    amd.set_precision(args.precision)
    # one scene (cloud) per GPU, sharded by size like a deployment would; scene id doubles as the seed
    my_scenes = shard_scenes([N0] * world, world)[rank]
    assert len(my_scenes) == 1
    levels = build_stack(amd, O, device, seed=my_scenes[0])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        _, dt = job_throughput(float(N0 * steps), dt, dist, device)  # MAX over ranks
        return dt

    if args.no_graph:
        run_stack, run_layer = (lambda: step(levels)), (lambda: step(levels[:1]))
    else:
        run_stack, run_layer = GraphedStep(levels), GraphedStep(levels[:1])
    dt_stack = timed(run_stack, args.steps, args.warmup)
    dt_layer = timed(run_layer, args.steps, max(1, args.warmup // 2))
    ms_step = dt_stack / args.steps * 1e3
    ms_layer = dt_layer / args.steps * 1e3

    stages = profile_level0(lib, levels[0], reps=5)
    fl = layer_flops(levels[0]["n"], levels[0]["e"])
    t24 = args.precision == "bf16x3" and CH >= 64 and CH % 2 == 0 and "SE3_NO_T24" not in os.environ
    sb = stage_bytes(levels[0]["n"], levels[0]["e"], t24)
    peak_tf = PEAK_MFMA_TFLOPS[args.precision]
    dom = max(stages, key=lambda t: stages[t][0]) if stages else None
    roofline = None
    if dom is not None:
        sec = stages[dom][0] * 1e-3
        tf = fl.get(dom, 0) / sec / 1e12
        gbs = sb.get(dom, 0) / sec / 1e9
        frac_mfma, frac_hbm = tf / peak_tf, gbs / PEAK_HBM_GBPS
        if frac_hbm >= frac_mfma:
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(frac_hbm, 4), "traffic": None, "algorithmic_bytes_per_launch": sb.get(dom, 0),
                        "mfma_frac": round(frac_mfma, 4)}
        else:
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(tf, 2), "peak": peak_tf, "unit": "TFLOP/s",
                        "frac": round(frac_mfma, 4), "traffic": None, "algorithmic_flops_per_launch": fl.get(dom, 0),
                        "hbm_frac": round(frac_hbm, 4)}
        try:  # measured PMC traffic of the same kernel (separate rocprofv3 --pmc passes, committed under profiles/)
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
                tr = json.load(fh).get(dom) if args.precision == "bf16x3" else None
            roofline["traffic"] = tr["hbm_bytes"] if tr else None
        except OSError:
            pass
        roofline.update({"avg_launch_ms": round(stages[dom][0], 4), "launches": stages[dom][1],
                         "stages_ms": {t: round(v[0], 4) for t, v in sorted(stages.items())}})
    # "end-to-end" number of SURVEY.md section 8d: the conv step plus the per-step neighbourhood work of the 4 levels
    # (ball query, source-major edge list for backward); it needs one host sync per level for E, so it is timed
    # eagerly beside the graph-replayed conv step
    def build_neighbourhoods():
        for lv in levels:
            nb = amd.pc.BQNeighborhood(lv["pc"], lv["pc"], lv["r"])
            # what the layer does on first use: int32 views + the source-major list for backward (a cloud against
            # itself gives a symmetric radius graph, whose source-major list is the edge list itself)
            amd.layers._geometry_of(lv["pc"], lv["pc"], nb).transpose()
    for _ in range(2):
        build_neighbourhoods()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(5):
        build_neighbourhoods()
    torch.cuda.synchronize(device)
    ms_nbh = (time.perf_counter() - t0) / 5 * 1e3
    lb = layer_bytes(levels[0]["n"], levels[0]["e"])
    hbm = {"algorithmic_bytes_per_layer": lb, "achieved": round(lb / (ms_layer * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS,
           "unit": "GB/s", "frac": round(lb / (ms_layer * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)}

    result = {
        "metric": "Mpoints/sec fwd+bwd PNEConvLayerRotEquiv (N=64k,k=32,F=2)",
        "value": round(N0 * world / (ms_step * 1e-3) / 1e6, 3),
        "unit": "Mpoints/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "bf16x3 (fp32 split into bf16 hi+lo, 3 MFMA products, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "4-level PNEConvLayerRotEquiv stack, conv-only fwd+bwd (dX,dA,dbeta,dW), one cloud per GPU",
                   "n_points": N0, "k": K_DEG, "frames": FRAMES, "channels": CH, "num_basis": KB,
                   "level_points": [lv["n"] for lv in levels], "level_edges": [lv["e"] for lv in levels],
                   "mean_degree_level0": round(levels[0]["e"] / levels[0]["n"], 2),
                   "launch": "eager" if args.no_graph else "hipGraph replay of the captured step", "sharding": "one scene per rank, no data-path collective"},
        "single_layer": {"ms_per_step": round(ms_layer, 4), "value": round(N0 * world / (ms_layer * 1e-3) / 1e6, 3),
                         "unit": "Mpoints/s", "hbm_roofline": hbm},
        "end_to_end": {"neighbourhood_ms": round(ms_nbh, 4), "ms_per_step": round(ms_step + ms_nbh, 4),
                       "value": round(N0 * world / ((ms_step + ms_nbh) * 1e-3) / 1e6, 3), "unit": "Mpoints/s",
                       "note": "conv step + ball query and the operator's geometry views (incl. source-major edge lists) of the 4 levels, rebuilt every step"},
        "roofline": roofline,
    }
    # the "trivial result gather": one checksum of the level-0 output per scene, to rank 0
    with torch.no_grad():
        lv = levels[0]
        out0 = lv["conv"](p_pc_in=lv["pc"], p_pc_out=lv["pc"], p_in_features=lv["x"], p_neighborhood=lv["nbh"])
        sums = gather_scene_results({my_scenes[0]: out0.double().sum().reshape(1)}, dist, dst=0)
    if rank == 0:
        result["scene_checksums"] = {str(k): round(float(v), 6) for k, v in sorted(sums.items())}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(O)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
