#!/usr/bin/env python3
"""Benchmark of the PNEConvLayerRotEquiv hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--precision bf16x3|fp32]

N > 1: one rank per GPU.  Under ``torch.distributed.run`` (RANK / WORLD_SIZE in the environment) this process is
a rank; started plainly it first launches ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a
CHILD process -- before anything here touches the GPU -- relays rank 0's JSON line and exits with the child's code.
Scenes are sharded by rank (se3conv3d_amd.sharding), the data path has no collective: only the timing barrier, a
MAX all-reduce of the wall time and a once-per-job gather of one checksum per scene ("scaling": "weak").

Workload (``--workload headline`` = BASELINE.json's metric "Mpoints/sec fwd+bwd PNEConvLayerRotEquiv
(N=64k,k=32,F=2)"): one synthetic cloud per rank, N0 = 65 536 points ~ U[0,1)^3, F = 2 random frames per point,
C = 64 -> 64 channels, K = 32 basis functions, radius for mean degree 32; a 4-level stack = one same-level
convolution per level of a grid-subsampled hierarchy (cell doubling, radius = 2 x cell).  One step = forward +
backward (dX, dA, dbeta, dW) of all four levels, neighbourhoods prebuilt ("conv-only").
value = level-0 points of all ranks / step time.  Other workloads: se3conv3d_amd/workloads.py.

The JSON line also carries (N = 1 only): the single full-resolution layer, the same stack in exact-fp32 arithmetic
and launched eagerly, the end-to-end step with per-step neighbourhoods, ``roofline`` of the dominant kernel on
SURVEY section 8d's algorithmic bytes (HIP events inside the library, on the launch stream; intermediates only
show up in ``traffic``), ``layer_frac`` / ``stack_frac``, and ``cpu_baseline`` (the oracle on this box's host cores).
"""
import argparse
import ctypes as C
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Mpoints/sec fwd+bwd PNEConvLayerRotEquiv (N=64k,k=32,F=2)"
# MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate; bf16 dense MFMA ~2.5 PF.
# In "bf16x3" every multiply costs 3 bf16 MFMA products, so mfma_frac <= 1/3 by construction there.
PEAK_MFMA_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0, "bf16x3_t16": 2500.0}
PEAK_HBM_GBPS = 8000.0          # HBM3E spec
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r06_traffic.json")


def library_source_sha() -> str:
    """sha256 over the kernel sources the library is built from: what a committed PMC measurement (profiles/*_traffic.json,
    written by tools/pmc_traffic.py with the same function) is valid for."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "se3conv3d_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


# ------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def mark(msg: str) -> None:
    """Progress marks on stderr (SE3_BENCH_VERBOSE=1): which leg a run was in when it died."""
    if os.environ.get("SE3_BENCH_VERBOSE"):
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def launch_ranks(args, argv) -> int:
    """Parent of an N-rank run: no GPU call has happened in this process (``import torch`` does not initialise
    HIP), the ranks are children of ``torch.distributed.run``.  Their stdout is relayed line by line."""
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this pool
    # host threads per rank: the ranks share this box's cores (torch.distributed.run would pin 1 thread per rank; the
    # ranks set their own count from the cores they may use, run_rank)
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, usable_cores() // args.gpus))))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


# ------------------------------------------------------------------------------------------ the step
def step(levels):
    from se3conv3d_amd import ops as _ops

    for lv in levels:
        # a training step builds its clouds anew (the task scripts' create_hierarchy): the geometry records a cloud's
        # convolutions share are rebuilt once per step here too, inside the timed region
        _ops.invalidate_prepared(lv["pc"])
        lv["x"].grad = None
        for p in lv["conv"].parameters():
            p.grad = None
        out = lv["conv"](p_pc_in=lv["pc"], p_pc_out=lv["pc"], p_in_features=lv["x"], p_neighborhood=lv["nbh"])
        out.backward(lv["g"])


class GraphedStep:
    """The step captured once into a HIP graph and replayed (launch-bound for the small levels otherwise).
    Inputs/outputs live in static buffers, so a replay recomputes exactly the same forward+backward on whatever
    the buffers hold."""

    def __init__(self, levels, fn=step):
        self.levels = levels
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):  # warm-up on the side stream: lazy builds (transposed edge list), allocator
                fn(levels)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: a helper thread of the process (the RCCL watchdog of an N-rank run) may touch the runtime
        # while this thread captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            fn(levels)

    def __call__(self):
        self.graph.replay()


def step_two_clouds(rec, nbh=None):
    """Forward + backward of one level-to-level convolution (workloads.build_down_up)."""
    from se3conv3d_amd import ops as _ops

    if rec.get("own_clouds", True):  # (a network's call list shares its clouds between calls: its driver invalidates them once per step)
        _ops.invalidate_prepared(rec["pc_in"])
        _ops.invalidate_prepared(rec["pc_out"])
    rec["x"].grad = None
    for p in rec["conv"].parameters():
        p.grad = None
    out = rec["conv"](p_pc_in=rec["pc_in"], p_pc_out=rec["pc_out"], p_in_features=rec["x"],
                      p_neighborhood=rec["nbh"] if nbh is None else nbh)
    out.backward(rec["g"])


def profile_level(lib, lv, reps, fn=None):
    """Per-stage launch times of one level: HIP events the library records on its own launch stream."""
    lib.se3_profile_reset()
    lib.se3_profile_enable(1)
    for _ in range(reps):
        step([lv]) if fn is None else fn()
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
        # (average per launch, launches, total per step)
        stages[tag] = (ms.value / max(cnt.value, 1), cnt.value, ms.value / reps)
    lib.se3_profile_reset()
    return stages


# ------------------------------------------------------------------------------------------ CPU baseline
def usable_cores() -> int:
    """Host cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box hands
    a job a share of a larger host; torch with one thread per LOGICAL core of the host then thrashes -- measured 70x
    slower than 8 threads on a 256-thread host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                        n = min(n, max(1, q // int(fh.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline():
    """the oracle (a port of the reference's Python path, kind "port") on this box's host
    cores -- threads = all cores and 8 (MAX_NUM_THREADS of tasks/SemSeg/train_dfaust_rot.py:17); configuration 1
    (N=1024, k=16, F=1, C=32) and an N=8192 proxy of the headline shape (k=32, F=2, C=64; the full 65 536-point
    layer takes minutes); forward and backward timed separately; median of up to 5 runs inside a time budget (one warm-up run for configuration 1, none for the proxy)."""
    from oracle import se3conv_oracle as O  # the checker, used here as the reported CPU baseline only
    from se3conv3d_amd.workloads import radius_for_degree

    def case(n, k, f, c, threads, budget_s, warm):
        torch.set_num_threads(threads)
        g = torch.Generator().manual_seed(0)
        pts = torch.rand(n, 3, generator=g)
        bid = torch.zeros(n, dtype=torch.int32)
        fr = O.random_frames(n, f, g)
        r = radius_for_degree(n, k)
        nb, _ = O.ball_query(pts, pts, bid, bid, r)
        a, b, w = O.init_parameters(9, c, c, 32, g)
        x = torch.randn(n * f, c, generator=g)
        go = torch.randn(n * f, c, generator=g)
        rho, nu = torch.tensor(1.0 / r), torch.tensor(n / nb.shape[0])
        fwd, bwd = [], []
        t_start = time.perf_counter()
        for it in range(warm + 5):
            xs = x.clone().requires_grad_(True)
            ps = [t.clone().requires_grad_(True) for t in (a, b, w)]
            t0 = time.perf_counter()
            out = O.conv_forward(pts, pts, fr, fr, nb, xs, *ps, rho, nu)
            t1 = time.perf_counter()
            out.backward(go)
            t2 = time.perf_counter()
            if it >= warm:
                fwd.append(t1 - t0), bwd.append(t2 - t1)
            if fwd and time.perf_counter() - t_start > budget_s:
                break
        mf, mb = statistics.median(fwd), statistics.median(bwd)
        return {"n": n, "k_achieved": round(nb.shape[0] / n, 2), "frames": f, "channels": c, "threads": threads,
                "fwd_s": round(mf, 4), "bwd_s": round(mb, 4), "runs": len(fwd),
                "mpoints_per_s": round(n / (mf + mb) / 1e6, 6)}

    all_cores = usable_cores()
    prev = torch.get_num_threads()
    cases = []
    for threads in (all_cores, 8):
        cases.append(dict(case(1024, 16, 1, 32, threads, 3.0, 1), config="config 1"))
        cases.append(dict(case(8192, 32, 2, 64, threads, 15.0, 0), config="headline proxy"))
    torch.set_num_threads(prev)
    best = max((c for c in cases if c["config"] == "headline proxy"), key=lambda c: c["mpoints_per_s"])
    return {"value": best["mpoints_per_s"], "unit": "Mpoints/s", "cores": best["threads"], "kind": "port",
            "sample": "oracle (torch CPU port of the reference's Python path) single layer, fwd and bwd timed separately, "
                      f"median of up to 5 runs inside a 15 s budget; value = headline proxy N=8192, k~{best['k_achieved']}, F=2, C=64, K=32 "
                      f"at {best['threads']} threads ({best['fwd_s']} s fwd + {best['bwd_s']} s bwd); usable host cores (affinity / cgroup quota): {all_cores} of {os.cpu_count()} logical",
            "cases": cases}


# ------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    from se3conv3d_amd import workloads as W
    from se3conv3d_amd.sharding import gather_scene_results, job_throughput, shard_scenes  # gather: dry run only

    spec = W.WORKLOADS[args.workload]
    n0 = spec["points"] * spec["clouds"]
    dist = None
    # host side of a rank: building its scene and launching kernels.  N ranks share the box's usable cores (affinity /
    # cgroup quota), at most 8 threads each (MAX_NUM_THREADS of the reference's task scripts): 8 ranks never oversubscribe
    cpu_threads = max(1, min(8, usable_cores() // world))
    torch.set_num_threads(cpu_threads)
    # SE3_BENCH_FORCE_DIST=1: a one-rank job still goes through the process group (RCCL init, barrier, MAX reduce, result
    # gather, graph capture next to a live communicator) -- the rehearsal of the N-rank path a one-GPU box can run
    if world > 1 or os.environ.get("SE3_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # one scene (cloud / batch) per GPU, sharded by size like a deployment would; scene id doubles as the seed
    my_scenes = shard_scenes([n0] * world, world)[rank]
    assert len(my_scenes) == 1

    if args.dry_run:
        # CPU rehearsal of the N-rank protocol (tests/test_bench_launch.py): gloo, no GPU, no library call --
        # sharding, barrier, MAX-over-ranks timing and the result gather run exactly as in the real path
        device = torch.device("cpu")
        if os.environ.get("SE3_BENCH_FAIL_RANK") == str(rank):
            raise SystemExit(3)  # rehearsal hook: a rank that dies must fail the parent (tests/test_bench_launch.py)
        if dist is not None:
            dist.init_process_group("gloo")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            torch.rand(1024, 3).sum()
        dt = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        _, dt = job_throughput(float(n0 * args.steps), dt, dist, device)
        sums = gather_scene_results({my_scenes[0]: torch.tensor([float(my_scenes[0])])}, dist, dst=0)
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "dry_run": True,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "config": {"workload": args.workload, "cpu_threads_per_rank": cpu_threads},
                              "scenes": sorted(sums),
                              "scene_checksums": {str(k): float(v[0]) for k, v in sorted(sums.items())}}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.share_gpu:  # rehearsal on a box with fewer GPUs than ranks: ranks share devices, gloo carries the barrier
        local_rank %= torch.cuda.device_count()
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    reduce_device = torch.device("cpu") if args.share_gpu else device
    if dist is not None:
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    import se3conv3d_amd as amd
    from se3conv3d_amd import _lib

    lib = _lib.load()
    amd.set_precision(args.precision)
    levels = W.build_stack(spec, device, seed=my_scenes[0], order=args.point_order)
    frames = spec["frames"]

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
        _, dt = job_throughput(float(n0 * steps), dt, dist, reduce_device)  # MAX over ranks
        return dt

    def timed_events(fn, steps):
        """SURVEY 8d protocol beside the wall clock: one HIP event pair per step on the launch stream (graph replays and
        eager launches both run on torch's current stream), median and minimum over the steps."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize()
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        return {"median_ms": round(statistics.median(ms), 4), "min_ms": round(ms[0], 4), "steps": steps,
                "method": "HIP events per step on the launch stream"}

    def streaming_rates_gbps():
        """What this chip streams, two ways (1 GiB buffers, best of 6, HIP events): a device-to-device copy (1 GiB read +
        1 GiB written -- the mix a layer's launches have) and a write-only fill (the fastest stream this chip sustains:
        no read / write turn-arounds; an upper bound no mixed stream reaches)."""
        a = torch.empty(1 << 28, dtype=torch.float32, device=device)
        b = torch.empty_like(a)

        def best_ms(fn):
            best = 1e9
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best

        copy = 2 * (1 << 30) / (best_ms(lambda: b.copy_(a)) * 1e-3) / 1e9
        fill = (1 << 30) / (best_ms(lambda: b.fill_(1.0)) * 1e-3) / 1e9
        del a, b
        return copy, fill

    launch = "eager" if args.no_graph else "hipGraph replay of the captured step"
    if args.no_graph:
        run_stack = lambda: step(levels)
    else:
        try:
            run_stack = GraphedStep(levels)
        except RuntimeError as exc:
            if world == 1:
                raise
            # an N-rank run must not die on a capture refused next to a live process group: this rank launches eagerly.
            # A stream left in capture mode makes the synchronisation itself raise: then the rank fails (non-zero exit,
            # which fails the job) rather than timing something undefined
            try:
                torch.cuda.synchronize()
            except RuntimeError as exc2:
                raise SystemExit(f"rank {rank}: graph capture failed ({str(exc)[:120]}) and the device cannot be "
                                 f"synchronised afterwards ({str(exc2)[:120]})")
            run_stack = lambda: step(levels)
            launch = f"eager on rank {rank} (graph capture failed: {str(exc)[:120]})"
    mark("stack")
    dt_stack = timed(run_stack, args.steps, args.warmup)
    ms_step = dt_stack / args.steps * 1e3
    mpts = lambda ms: round(n0 * world / (ms * 1e-3) / 1e6, 3)
    dtype = {"fp32": "f32", "bf16x3": "bf16x3 (fp32 split into bf16 hi+lo, 3 MFMA products, fp32 accumulate)",
             "bf16x3_t16": "bf16x3 arithmetic, row-sized intermediates T / U in 2.25-byte block floating point (16-bit mantissas, "
                           "one exponent per 4 channels)"}[args.precision]
    result = {
        "metric": METRIC, "value": mpts(ms_step), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "timing": {"value_from": "wall clock around the K steps between barriers + synchronize, MAX over ranks (the driver's contract)"},
        "config": {"workload": f"{args.workload}: {spec['note']}; conv-only fwd+bwd (dX,dA,dbeta,dW), one scene per GPU",
                   "n_points": n0, "clouds_per_gpu": spec["clouds"], "k": spec["degree"], "frames": frames,
                   "channels": spec["widths"], "num_basis": W.NUM_BASIS,
                   "level_points": [lv["n"] for lv in levels], "level_edges": [lv["e"] for lv in levels],
                   "mean_degree_level0": round(levels[0]["e"] / levels[0]["n"], 2), "point_order": args.point_order,
                   "launch": launch, "cpu_threads_per_rank": cpu_threads,
                   "sharding": "one scene per rank, no data-path collective" + (" (REHEARSAL: ranks share GPUs)" if args.share_gpu else "")},
    }

    if world == 1:
        lv0 = levels[0]
        mark("single layer / eager / stages")
        result["timing"]["hip_events"] = timed_events(run_stack, max(args.steps, 50))
        run_layer = (lambda: step(levels[:1])) if args.no_graph else GraphedStep(levels[:1])
        ms_layer = timed(run_layer, args.steps, max(1, args.warmup // 2)) / args.steps * 1e3
        ev_layer = timed_events(run_layer, max(args.steps, 50))
        ms_eager = timed(lambda: step(levels), args.steps, max(1, args.warmup // 2)) / args.steps * 1e3

        prof_reps = 5
        stages = profile_level(lib, lv0, reps=prof_reps)
        fl = W.layer_flops(lv0["n"], lv0["e"], frames, lv0["c"])
        own = W.stage_owned_bytes(lv0["n"], lv0["e"], frames, lv0["c"])
        peak_tf = PEAK_MFMA_TFLOPS[args.precision]
        core = {t: v for t, v in stages.items() if t in own}
        roofline = None
        if core:
            dom = max(core, key=lambda t: core[t][2])
            sec = core[dom][2] * 1e-3  # the stage's launches of one step together (one launch unless the rows are sliced)
            gbs, tf = own[dom] / sec / 1e9, fl.get(dom, 0) / sec / 1e12
            # measured PMC traffic of the same kernel: separate rocprofv3 --pmc passes of an EARLIER run, committed under
            # profiles/ with the hash of the kernel sources they were taken with -- reported only while that still matches
            traffic, traffic_source = None, "none (no PMC file for this workload / precision)"
            try:
                with open(TRAFFIC_JSON) as fh:
                    tj = json.load(fh)
                if args.precision == "bf16x3" and args.workload == "headline":
                    if tj.get("_library_source_sha") == library_source_sha():
                        tr = tj.get(dom)
                        traffic = tr["hbm_bytes"] if tr else None
                        traffic_source = f"{os.path.relpath(TRAFFIC_JSON, ROOT)} (external: rocprofv3 --pmc passes, not this run; kernel sources {tj['_library_source_sha']})"
                    else:
                        traffic_source = (f"stale: {os.path.relpath(TRAFFIC_JSON, ROOT)} was measured with kernel sources "
                                          f"{tj.get('_library_source_sha')}, this tree is {library_source_sha()}")
            except OSError:
                traffic_source = "none (profiles/ file missing)"
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                        "algorithmic_bytes_per_launch": own[dom],
                        "traffic_over_algorithmic": round(traffic / own[dom], 3) if traffic else None,
                        "mfma_frac": round(tf / peak_tf, 4), "avg_launch_ms": round(core[dom][0], 4),
                        "launches": core[dom][1], "launches_per_step": core[dom][1] // prof_reps, "stage_ms_per_step": round(core[dom][2], 4),
                        "note": "achieved = SURVEY 8d bytes this launch owns (geometry + gathered rows; no row-sized "
                                "intermediates) / its HIP-event time; intermediates appear in traffic only",
                        "stages_ms": {t: round(v[2], 4) for t, v in sorted(stages.items())},
                        "stages_note": "per step: every launch of the stage in one forward + backward of the layer summed (prep runs in both passes)"}
        # what the library says it moves through memory for this shape (3-byte rows / 4-byte words / nothing)
        shp = _lib.Se3Shape(lv0["n"], lv0["n"], lv0["e"], frames, frames, lv0["c"], lv0["c"], W.NUM_BASIS,
                            _lib.PRECISIONS[args.precision])
        per_el = tuple(int(lib.se3conv_intermediate_row_bytes(C.byref(shp), w)) / (lv0["c"] * W.NUM_BASIS) for w in range(3))
        moved = W.stage_moved_bytes(lv0["n"], lv0["e"], frames, lv0["c"], per_el)
        lb = W.layer_bytes(lv0["n"], lv0["e"], frames, lv0["c"])
        sb = sum(W.layer_bytes(lv["n"], lv["e"], frames, lv["c"]) for lv in levels)
        result["roofline"] = roofline
        result["single_layer"] = {"ms_per_step": round(ms_layer, 4), "value": mpts(ms_layer), "unit": "Mpoints/s",
                                  "hip_events": ev_layer, "algorithmic_bytes": lb,
                                  "intermediate_bytes_per_element": dict(zip(("T", "U", "grad_T"), (round(b, 3) for b in per_el))),
                                  "least_bytes_with_intermediates": sum(moved.values())}
        # Yardsticks of the decomposition as built.  Every launch moves its owned bytes plus the row-sized intermediates
        # (T, U, grad_T) once each way -- the edge phase and its contraction cannot share a CU (the contraction's weight
        # operand is 512 KB of split bf16 against 160 KB of LDS).  Of those bytes the neighbour-row gathers
        # (3 x 4*E*F*C) are NOT HBM traffic: the gathered table (33 MB at the headline shape) sits in the memory-side cache.
        # So the bound that matters is `hbm_bytes` (intermediates + every other tensor once) at the rate this chip streams;
        # `least_bytes_with_intermediates` (gathers priced as if they were HBM reads) is kept beside it for continuity.
        mark("streaming rates")
        rate, fill_rate = streaming_rates_gbps()
        hbm_layer = sum(W.stage_hbm_bytes(lv0["n"], lv0["e"], frames, lv0["c"], per_el).values())
        gather_layer = sum(W.stage_gather_bytes(lv0["n"], lv0["e"], frames, lv0["c"]).values())
        least_stack = hbm_stack = 0
        for lv in levels:
            shp_l = _lib.Se3Shape(lv["n"], lv["n"], lv["e"], frames, frames, lv["c"], lv["c"], W.NUM_BASIS,
                                  _lib.PRECISIONS[args.precision])
            pe = tuple(int(lib.se3conv_intermediate_row_bytes(C.byref(shp_l), w)) / (lv["c"] * W.NUM_BASIS) for w in range(3))
            least_stack += sum(W.stage_moved_bytes(lv["n"], lv["e"], frames, lv["c"], pe).values())
            hbm_stack += sum(W.stage_hbm_bytes(lv["n"], lv["e"], frames, lv["c"], pe).values())
        ms_at = lambda nbytes, gbps: nbytes / (gbps * 1e9) * 1e3
        ceil_layer_ms, ceil_stack_ms = ms_at(sum(moved.values()), rate), ms_at(least_stack, rate)
        result["design_ceiling"] = {
            "streaming_rate_GBps": round(rate, 1), "rate_method": "1 GiB device-to-device copy, 2 GiB moved, best of 6 (HIP events)",
            "layer": {"hbm_bytes": hbm_layer, "cache_served_gather_bytes": gather_layer,
                      "hbm_bound_ms": round(ms_at(hbm_layer, rate), 4),
                      "achieved_over_hbm_bound": round(ms_at(hbm_layer, rate) / ms_layer, 4),
                      "least_bytes_with_intermediates": sum(moved.values()), "ms": round(ceil_layer_ms, 4),
                      "value": mpts(ceil_layer_ms), "achieved_over_ceiling": round(ceil_layer_ms / ms_layer, 4)},
            "stack": {"hbm_bytes": hbm_stack, "hbm_bound_ms": round(ms_at(hbm_stack, rate), 4),
                      "achieved_over_hbm_bound": round(ms_at(hbm_stack, rate) / ms_step, 4),
                      "value_at_hbm_bound": mpts(ms_at(hbm_stack, rate)),
                      "least_bytes_with_intermediates": least_stack, "ms": round(ceil_stack_ms, 4),
                      "value": mpts(ceil_stack_ms), "achieved_over_ceiling": round(ceil_stack_ms / ms_step, 4)},
            "write_only_rate": {"rate_GBps": round(fill_rate, 1), "rate_method": "1 GiB fill, best of 6 (HIP events)",
                                "layer_hbm_bound_ms": round(ms_at(hbm_layer, fill_rate), 4),
                                "stack_hbm_bound_ms": round(ms_at(hbm_stack, fill_rate), 4),
                                "note": "the same HBM bytes at the fastest stream this chip sustains (write-only)"},
            "note": "hbm_bytes = T, U, grad_T once per producer and per consumer (bytes per element as the library reports "
                    "them) + every other tensor once; the gathers are served by the memory-side cache and priced at zero "
                    "here.  achieved_over_hbm_bound is the honest distance to what this decomposition could reach; "
                    "achieved_over_ceiling (gathers priced as HBM reads) overstates it and is kept for comparison with "
                    "round 3.  BASELINE.json's 50 Mpoints/s is 1.31 ms per step"}
        result["layer_frac"] = round(lb / (ms_layer * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)
        result["stack_frac"] = round(sb / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)
        result["stack_algorithmic_bytes"] = sb
        result["eager"] = {"ms_per_step": round(ms_eager, 4), "value": mpts(ms_eager),
                           "note": "same step launched from Python every iteration (no graph replay)"}

        # The same step with its four levels on four streams inside ONE captured graph (fork at the start, join at the end).
        # The stack's levels do not feed each other in this benchmark, so this is legal here -- and it is NOT `value`: in the
        # reference's encoder consecutive convolutions are dependent.  What it shows is what independent convolutions gain
        # from a scheduler that overlaps them (the three lateral convolutions of FPNDecoder.py:118-131 are independent, so are
        # the parameter-gradient branches of any two layers): the small levels' launch latencies hide under level 0.
        mark("levels_concurrent")
        lv_streams = [torch.cuda.Stream() for _ in levels[1:]]

        from se3conv3d_amd import layers as _layers, ops as _ops

        def raw_step(lv):
            """forward + backward of one level through the raw operator calls (no autograd engine: its worker thread and a
            multi-stream capture do not mix on this runtime -- the module-level step on side streams made capture_end crash)"""
            conv = lv["conv"]
            geom = _layers._geometry_of(lv["pc"], lv["pc"], lv["nbh"])
            with torch.no_grad():
                out, t_save = _ops.se3conv_forward(geom, lv["x"], conv.proj_axes_, conv.proj_biases_, conv.conv_weights_,
                                                   conv.norm_neigh_dist_, conv.norm_num_neighs_, save_t=True)
                return _ops.se3conv_backward(geom, lv["x"], conv.proj_axes_, conv.proj_biases_, conv.conv_weights_,
                                             conv.norm_neigh_dist_, conv.norm_num_neighs_, t_save, lv["g"])

        def step_levels_concurrent(lvls):
            cur = torch.cuda.current_stream()
            for st in lv_streams:
                st.wait_stream(cur)
            keep = []
            for lv, st in zip(lvls[1:], lv_streams):
                with torch.cuda.stream(st):
                    keep.append(raw_step(lv))
            keep.append(raw_step(lvls[0]))
            for st in lv_streams:
                cur.wait_stream(st)
            for outs in keep:  # results produced on a side stream are used (and freed) on the launch stream
                for t in outs:
                    if t is not None:
                        t.record_stream(cur)

        try:
            if args.no_extra:
                raise RuntimeError("skipped (--no-extra)")
            run_cc = (lambda: step_levels_concurrent(levels)) if args.no_graph else GraphedStep(levels, fn=step_levels_concurrent)
            ms_cc = timed(run_cc, args.steps, max(1, args.warmup // 2)) / args.steps * 1e3
            result["levels_concurrent"] = {"ms_per_step": round(ms_cc, 4), "value": mpts(ms_cc), "unit": "Mpoints/s",
                                           "note": "NOT the headline: the four (mutually independent) levels of the step on four streams "
                                                   "of one captured graph; what a scheduler gains on independent convolutions"}
        except RuntimeError as exc:
            result["levels_concurrent"] = {"error": str(exc)[:200]}
            torch.cuda.synchronize()

        mark("forward_only")
        # inference: the forward pass of the same stack alone (eval mode, no autograd graph, T in the workspace)
        def forward_only(lvls):
            with torch.no_grad():
                for lv in lvls:
                    lv["conv"](p_pc_in=lv["pc"], p_pc_out=lv["pc"], p_in_features=lv["x"], p_neighborhood=lv["nbh"])

        try:
            if args.no_extra:
                raise RuntimeError("skipped (--no-extra)")
            for lv in levels:
                lv["conv"].eval()
            run_fwd = (lambda: forward_only(levels)) if args.no_graph else GraphedStep(levels, fn=forward_only)
            ms_fwd = timed(run_fwd, args.steps, max(1, args.warmup // 2)) / args.steps * 1e3
            result["forward_only"] = {"ms_per_step": round(ms_fwd, 4), "value": mpts(ms_fwd), "unit": "Mpoints/s",
                                      "note": "forward pass of the same stack alone (inference: eval mode, torch.no_grad)"}
        except RuntimeError as exc:  # an extra leg must not take the line down with it ...
            result["forward_only"] = {"error": str(exc)[:200]}
            try:  # ... but the later legs must not be timed on a stream a refused capture left half open
                torch.cuda.synchronize()
            except RuntimeError as exc2:
                raise SystemExit(f"forward_only leg failed ({str(exc)[:120]}) and the device cannot be synchronised "
                                 f"afterwards ({str(exc2)[:120]}): the remaining legs would time something undefined")
        finally:
            for lv in levels:
                lv["conv"].train()

        # "end-to-end" number of SURVEY.md section 8d: the conv step plus the per-step neighbourhood work of the levels.
        # Ball queries go into capacity-bounded edge buffers (1.25 x the known edge count; the edge count stays on the
        # device), so the whole thing -- 4 ball queries + 4 x (forward + backward) -- has no host synchronisation and
        # replays as ONE captured graph; the overflow flags are read once after the timed region.
        mark("end_to_end")
        caps = [int(lv["e"] * 1.25) + 64 for lv in levels]
        flags = []
        extra = not args.no_extra

        def e2e_step(lvls):
            flags.clear()
            for lv, cap in zip(lvls, caps):
                amd.ops.forget_source_grids(lv["pc"])  # (a step's clouds are new: no source grid of the step before, see faust_step)
                nb = amd.pc.BQNeighborhood(lv["pc"], lv["pc"], lv["r"], p_capacity=cap)
                flags.append(nb.edge_info_)
                step([dict(lv, nbh=nb)])

        # the same work scheduled the way a pipeline would: the neighbourhoods depend on the points only, so the builds
        # of levels 1.. (small, latency-bound launches) run on a second stream underneath level 0's convolution
        bq_stream = torch.cuda.Stream()

        def e2e_step_overlapped(lvls):
            flags.clear()
            for lv in lvls:
                amd.ops.forget_source_grids(lv["pc"])
            cur = torch.cuda.current_stream()
            bq_stream.wait_stream(cur)
            nbs = [amd.pc.BQNeighborhood(lvls[0]["pc"], lvls[0]["pc"], lvls[0]["r"], p_capacity=caps[0])]
            with torch.cuda.stream(bq_stream):
                for lv, cap in zip(lvls[1:], caps[1:]):
                    nb = amd.pc.BQNeighborhood(lv["pc"], lv["pc"], lv["r"], p_capacity=cap)
                    for t in (nb.neighbors_i32_, nb.start_ids_, nb.edge_info_, nb.sources_i32_):
                        if t is not None:
                            t.record_stream(cur)
                    nbs.append(nb)
            flags.extend(nb.edge_info_ for nb in nbs)
            step([dict(lvls[0], nbh=nbs[0])])
            cur.wait_stream(bq_stream)
            step([dict(lv, nbh=nb) for lv, nb in zip(lvls[1:], nbs[1:])])

        def check_flags():
            assert all(int(f[1]) == 0 for f in flags), "ball query overflowed its edge buffer"
            assert [int(f[0]) for f in flags] == [lv["e"] for lv in levels], "bounded ball query found a different edge count"

        if extra:
            run_e2e = (lambda: e2e_step(levels)) if args.no_graph else GraphedStep(levels, fn=e2e_step)
            ms_e2e = timed(run_e2e, args.steps, max(1, args.warmup // 2)) / args.steps * 1e3
            check_flags()
            run_e2e_ov = (lambda: e2e_step_overlapped(levels)) if args.no_graph else GraphedStep(levels, fn=e2e_step_overlapped)
            ms_e2e_ov = timed(run_e2e_ov, args.steps, max(1, args.warmup // 2)) / args.steps * 1e3
            check_flags()
            ms_e2e_eager = timed(lambda: e2e_step(levels), max(5, args.steps // 4), 2) / max(5, args.steps // 4) * 1e3
            result["end_to_end"] = {"ms_per_step": round(ms_e2e, 4), "value": mpts(ms_e2e), "unit": "Mpoints/s",
                                    "neighbourhood_ms": round(ms_e2e - ms_step, 4), "eager_ms_per_step": round(ms_e2e_eager, 4),
                                    "overlapped": {"ms_per_step": round(ms_e2e_ov, 4), "value": mpts(ms_e2e_ov),
                                                   "neighbourhood_ms": round(ms_e2e_ov - ms_step, 4),
                                                   "note": "levels 1-3 build their neighbourhoods on a second stream under level 0's "
                                                           "convolution (they depend on the points only); same graph, same results"},
                                    "note": "ball query of every level (capacity-bounded edge buffers, no host sync) + the conv step, "
                                            "one captured graph, level by level on one stream; eager_ms_per_step = the same launched from Python"}

        # The convolutions BETWEEN two levels: the encoder's first down-convolution (level 0 -> 1) and
        # the decoder's last up-convolution (level 1 -> 0) of this workload's hierarchy (and of dfaust_f2's, the
        # reference's headline task), forward + backward.  N_in != N_out and a non-symmetric edge relation: backward reads
        # the source-major copy of the edge list.  Two timings each, both one captured graph: `conv_only` (neighbourhood and
        # its transposed copy prebuilt, like the headline), and `with_neighbourhood` -- the ball query (capacity-bounded, no
        # host sync) and se3_csr_transpose inside the step, as the task scripts pay them (they rebuild the hierarchy every
        # step, tasks/SemSeg/train_dfaust_rot.py:108-158).  Not part of `value`.
        mark("down_up")
        def down_up_leg(wname):
            wspec = W.WORKLOADS[wname]
            # (Round 4: next to a live RCCL communicator the captured level 1 -> 0 step faulted on replay -- the memset nodes of
            # rocPRIM's one-sweep radix sort inside the transposition, on the HIP runtime PyTorch ships; the library issues no
            # hipMemsetAsync any more and the leg is captured whether or not a process group exists)
            eager_leg = args.no_graph
            mark(f"down_up {wname}: build")
            recs = W.build_down_up(wspec, device, seed=my_scenes[0], order=args.point_order)
            leg = {}
            reps_t = max(10, args.steps // 2)
            for rec in recs:
                cap = int(rec["e"] * 1.25) + 64
                held = []

                def with_nbh(_lv=None, rec=rec, cap=cap, held=held):
                    amd.ops.forget_source_grids(rec["pc_in"])
                    amd.ops.forget_source_grids(rec["pc_out"])
                    nb = amd.pc.BQNeighborhood(rec["pc_in"], rec["pc_out"], rec["r"], p_capacity=cap)
                    held[:] = [nb]
                    step_two_clouds(rec, nb)

                conv_only = lambda _lv=None, rec=rec: step_two_clouds(rec)
                mark(f"down_up {wname} {rec['name']}: conv_only")
                run_c = conv_only if eager_leg else GraphedStep(None, fn=conv_only)
                ms_c = timed(run_c, reps_t, 3) / reps_t * 1e3
                mark(f"down_up {wname} {rec['name']}: with_neighbourhood")
                run_n = with_nbh if eager_leg else GraphedStep(None, fn=with_nbh)
                ms_n = timed(run_n, reps_t, 3) / reps_t * 1e3
                assert int(held[0].edge_info_[1]) == 0 and int(held[0].edge_info_[0]) == rec["e"], "bounded two-cloud ball query"
                mark(f"down_up {wname} {rec['name']}: stage times")
                stages = profile_level(lib, None, 5, fn=lambda rec=rec: step_two_clouds(rec))
                ab = W.layer_bytes_two_clouds(rec["n_in"], rec["n_out"], rec["e"], rec["f"], rec["f"], rec["c_in"], rec["c_out"])
                leg[rec["name"]] = {
                    "n_in": rec["n_in"], "n_out": rec["n_out"], "edges": rec["e"], "c_in": rec["c_in"], "c_out": rec["c_out"],
                    "frames": rec["f"], "radius": round(rec["r"], 5), "launch": "eager" if eager_leg else "hipGraph replay",
                    "conv_only_ms": round(ms_c, 4), "with_neighbourhood_ms": round(ms_n, 4),
                    "neighbourhood_and_transpose_ms": round(ms_n - ms_c, 4),
                    "algorithmic_bytes": ab, "layer_frac": round(ab / (ms_c * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                    "layer_frac_with_neighbourhood": round(ab / (ms_n * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4),
                    "stages_ms": {t: round(v[2], 4) for t, v in sorted(stages.items())}}
            return leg

        try:
            if not extra:
                raise RuntimeError("skipped (--no-extra)")
            result["down_up"] = {w: down_up_leg(w) for w in dict.fromkeys([args.workload, "dfaust_f2"])}
            result["down_up"]["note"] = ("level 0 -> 1 down-convolution (radius of level 0) and level 1 -> 0 up-convolution (radius of "
                                         "level 1) of the workload's hierarchy, fwd+bwd; layer_frac on SURVEY 8d's bytes for N_in != N_out")
        except RuntimeError as exc:
            result["down_up"] = {"error": str(exc)[:200]}
            torch.cuda.synchronize()

        # Everything the library owns of one DFaust training step (BASELINE config 2): what the task script
        # rebuilds every step -- create_hierarchy (grid sub-samples, 16-NN + PCA frames on every level, the output cloud;
        # tasks/SemSeg/train_dfaust_rot.py:108-158) and the twelve neighbourhoods of the network with the source-major lists
        # backward reads -- then the network's 21 convolution calls forward + backward (call list recorded from the reference,
        # tests/golden/network_faust_calls.npz) and the row-wise glue of its eight ResNetFormer blocks (the blocks with their
        # convolution taken out: batch norm, skip + drop path, the two dense layers).  The geometry runs eagerly -- the
        # hierarchy build reads each level's size back (one host synchronisation per level) -- the convolutions and the glue
        # as captured graphs; `ms` is their sum.  Loss, optimizer and data loading are the task script's.  Not part of `value`.
        mark("faust_step")

        def faust_step_leg():
            fixture = os.path.join(ROOT, "tests", "golden", "network_faust_calls.npz")
            bodies = 32
            pts_raw, bid_raw = W.faust_raw_batch(device, bodies)
            calls = W.faust_network_calls(fixture)
            clouds = W.faust_clouds(pts_raw, bid_raw)
            nbhs = W.faust_neighbourhoods(clouds, calls)
            caps = {k: int(nb.num_edges() * 1.25) + 64 for k, nb in nbhs.items()}
            reps_g = 10
            ms_clouds = timed(lambda: W.faust_clouds(pts_raw, bid_raw), reps_g, 2) / reps_g * 1e3

            def step_neighbourhoods():
                for c in clouds:  # a step's clouds are new objects: only the queries of ONE step share a source grid
                    amd.ops.forget_source_grids(c)
                return W.faust_neighbourhoods(clouds, calls, caps)

            ms_nbhs = timed(step_neighbourhoods, reps_g, 2) / reps_g * 1e3
            recs = W.build_faust_network_convs(device, fixture, bodies=bodies)
            for r in recs:
                r["own_clouds"] = False
            rec_clouds = {id(c): c for r in recs for c in (r["pc_in"], r["pc_out"])}

            def all_convs(_lv=None):
                for c in rec_clouds.values():
                    amd.ops.invalidate_prepared(c)
                for r in recs:
                    step_two_clouds(r)

            reps_c = max(10, args.steps // 2)
            run_convs = all_convs if args.no_graph else GraphedStep(None, fn=all_convs)
            ms_convs = timed(run_convs, reps_c, 3) / reps_c * 1e3

            class NoConv(torch.nn.Module):  # the block's convolution is in `convolutions` already
                def forward(self, p_pc_in, p_pc_out, p_in_features, p_neighborhood):
                    return p_in_features

            fac = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu")
            glue = []
            for level, width in ((1, 32), (2, 64), (3, 128), (4, 256)):  # NUM_BLOCKS [2,2,2,2], NUM_FEATURES [32,64,128,256]
                pc = clouds[level]
                rows = pc.pts_.shape[0] * 2
                for _ in range(2):
                    blk = amd.ResNetFormer(width, width, fac, amd.BatchNormPC, 0.1).to(device)
                    blk.spatial_conv_ = NoConv()
                    blk.train()
                    glue.append((blk, pc, torch.randn(rows, width, device=device, requires_grad=True),
                                 torch.randn(rows, width, device=device)))

            def all_glue(_lv=None):
                for blk, pc, x, g in glue:
                    x.grad = None
                    blk.zero_grad(set_to_none=True)
                    blk(pc, x, None).backward(g)

            run_glue = all_glue if args.no_graph else GraphedStep(None, fn=all_glue)
            ms_glue = timed(run_glue, reps_c, 3) / reps_c * 1e3
            n_lv0 = int(clouds[0].pts_.shape[0])
            total = ms_clouds + ms_nbhs + ms_convs + ms_glue
            return {"ms": round(total, 3), "value": round(n_lv0 / total / 1e3, 3), "unit": "Mpoints/s (level-0 points)",
                    "bodies": bodies, "points_per_level": [int(c.pts_.shape[0]) for c in clouds],
                    "parts_ms": {"hierarchy_and_frames": round(ms_clouds, 3), "neighbourhoods": round(ms_nbhs, 3),
                                 "convolutions": round(ms_convs, 3), "block_glue": round(ms_glue, 3)},
                    "share": {"geometry": round((ms_clouds + ms_nbhs) / total, 3), "convolutions": round(ms_convs / total, 3),
                              "block_glue": round(ms_glue / total, 3)},
                    "note": "one DFaust training step as far as the library owns it: create_hierarchy + 12 neighbourhoods (eager: "
                            "one host read-back per level) + 21 convolutions fwd+bwd + glue of 8 ResNetFormer blocks (captured "
                            "graphs); the sum of the four parts"}

        try:
            if not extra or args.workload != "headline":
                raise RuntimeError("skipped")
            result["faust_step"] = faust_step_leg()
        except RuntimeError as exc:
            result["faust_step"] = {"error": str(exc)[:200]}
            torch.cuda.synchronize()

        mark("fp32 leg")
        if not args.no_fp32 and not args.no_extra and args.precision != "fp32":
            amd.set_precision("fp32")
            run32 = (lambda: step(levels)) if args.no_graph else GraphedStep(levels)
            ms32 = timed(run32, max(3, args.steps // 2), 2) / max(3, args.steps // 2) * 1e3
            amd.set_precision(args.precision)
            result["fp32_mode"] = {"ms_per_step": round(ms32, 4), "value": mpts(ms32), "unit": "Mpoints/s",
                                   "note": "same stack with every contraction on v_mfma_f32_32x32x2_f32 (exact fp32 products)"}

        # The third arithmetic mode: bf16x3 products with T / U in the 2.25-byte block format.  Timed
        # like the headline (same stack, graph replay) with its error measured at FULL size: every output and gradient of
        # the level-0 layer against the exact-fp32 mode's (itself within 1.5e-6 of the fp64 oracle), next to the default
        # mode's own distance -- the measured cost in accuracy of the bytes it saves.  Opt-in: not part of `value`.
        mark("t16 leg")
        if args.precision == "bf16x3" and not args.no_t16 and not args.no_extra:
            def layer_results(prec):
                amd.set_precision(prec)
                lv = levels[0]
                lv["x"].grad = None
                for p in lv["conv"].parameters():
                    p.grad = None
                out = lv["conv"](p_pc_in=lv["pc"], p_pc_out=lv["pc"], p_in_features=lv["x"], p_neighborhood=lv["nbh"])
                out.backward(lv["g"])
                return [out.detach().double()] + [t.grad.detach().double().clone() for t in
                                                  (lv["x"], lv["conv"].proj_axes_, lv["conv"].proj_biases_, lv["conv"].conv_weights_)]

            try:
                ref32 = layer_results("fp32")
                rel = lambda got: [float((a - b).norm() / b.norm()) for a, b in zip(got, ref32)]
                err_t16, err_def = rel(layer_results("bf16x3_t16")), rel(layer_results("bf16x3"))
                del ref32
                amd.set_precision("bf16x3_t16")
                run16 = (lambda: step(levels)) if args.no_graph else GraphedStep(levels)
                ms16 = timed(run16, args.steps, max(1, args.warmup // 2)) / args.steps * 1e3
                run16l = (lambda: step(levels[:1])) if args.no_graph else GraphedStep(levels[:1])
                ms16l = timed(run16l, args.steps, max(1, args.warmup // 2)) / args.steps * 1e3
                st16 = profile_level(lib, levels[0], reps=5)
                shp16 = _lib.Se3Shape(lv0["n"], lv0["n"], lv0["e"], frames, frames, lv0["c"], lv0["c"], W.NUM_BASIS,
                                      _lib.PRECISIONS["bf16x3_t16"])
                names = ("out", "dX", "dA", "dbeta", "dW")
                result["t16_mode"] = {
                    "ms_per_step": round(ms16, 4), "value": mpts(ms16), "unit": "Mpoints/s",
                    "single_layer_ms": round(ms16l, 4),
                    "row_bytes": {k: int(lib.se3conv_intermediate_row_bytes(C.byref(shp16), w)) for w, k in enumerate(("T", "U", "grad_T"))},
                    "rel_err_vs_fp32_mode": dict(zip(names, (float(f"{e:.3g}") for e in err_t16))),
                    "default_mode_rel_err_vs_fp32_mode": dict(zip(names, (float(f"{e:.3g}") for e in err_def))),
                    "worst_rel_err": float(f"{max(err_t16):.3g}"),
                    "stages_ms": {t: round(v[2], 4) for t, v in sorted(st16.items())},
                    "note": "same stack, bf16x3 products, T / U rows as 16-bit mantissas with one exponent per 4 channels (2.25 B per "
                            "element instead of 3); errors = ||x - x_fp32mode|| / ||x_fp32mode|| of the full-size level-0 layer"}
            except RuntimeError as exc:
                result["t16_mode"] = {"error": str(exc)[:200]}
                torch.cuda.synchronize()
            finally:
                amd.set_precision(args.precision)

    mark("result gather")
    # the "trivial result gather": one checksum of the level-0 output per scene.  A fixed-size record per rank, so it
    # travels as one tensor all-gather (RCCL over xGMI; gloo in the rehearsal) rather than through pickled objects.
    with torch.no_grad():
        lv = levels[0]
        out0 = lv["conv"](p_pc_in=lv["pc"], p_pc_out=lv["pc"], p_in_features=lv["x"], p_neighborhood=lv["nbh"])
        rec = torch.stack((torch.tensor(float(my_scenes[0]), dtype=torch.float64, device=device), out0.double().sum()))
        if dist is not None:
            rec = rec.to(reduce_device)
            parts = [torch.empty_like(rec) for _ in range(world)]
            dist.all_gather(parts, rec)
        else:
            parts = [rec]
        sums = {int(p[0].item()): float(p[1].item()) for p in parts}
    if rank == 0:
        assert sorted(sums) == list(range(world)), "every scene exactly once"
        result["scene_checksums"] = {str(k): round(float(v), 6) for k, v in sorted(sums.items())}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="headline", choices=["headline", "scannet150k_f1", "dfaust_f2", "dfaust_f4"])
    ap.add_argument("--point-order", default="random", choices=["random", "morton"],
                    help="row order of the synthetic points: as drawn, or sorted along a Z-order curve per scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the exact-fp32 leg")
    ap.add_argument("--no-extra", action="store_true", help="A/B runs: only the stack, the single layer and its stage times (no forward_only / "
                    "end_to_end / down_up / t16 / fp32 legs)")
    ap.add_argument("--no-t16", action="store_true", help="skip the leg of the third arithmetic mode (T / U in the 2.25-byte block format)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo rehearsal of the N-rank protocol (no GPU work)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: ranks may share a GPU (rank %% device count), gloo instead of RCCL")
    ap.add_argument("--precision", default=os.environ.get("SE3CONV_PRECISION", "bf16x3"), choices=["bf16x3", "fp32", "bf16x3_t16"])
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
