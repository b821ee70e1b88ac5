"""CPU, world_size 2 over gloo: the N>1 path of the benchmark / deployment -- scenes sharded by rank, no
data-path collective, results gathered to rank 0, throughput = sum of units / max time.  The per-scene
"result" here is the oracle's forward on a tiny cloud, so the gather is checked against a single-process run."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import se3conv_oracle as O
from se3conv3d_amd.sharding import gather_scene_results, job_throughput, shard_scenes

SCENES = [96, 40, 64, 128, 33]  # points per scene (ragged on purpose)


def scene_result(scene_id: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(100 + scene_id)
    n = SCENES[scene_id]
    pts = torch.rand(n, 3, generator=g)
    bid = torch.zeros(n, dtype=torch.int32)
    fr = O.random_frames(n, 2, g)
    nb, _ = O.ball_query(pts, pts, bid, bid, 0.35)
    a, b, w = O.init_parameters(9, 4, 4, 32, torch.Generator().manual_seed(7))  # replicated parameters
    x = torch.randn(n * 2, 4, generator=g)
    return O.conv_forward(pts, pts, fr, fr, nb, x, a, b, w, torch.tensor(1 / 0.35), torch.tensor(0.1))


def _worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_scenes(SCENES, world)[rank]
        local = {s: scene_result(s) for s in mine}
        merged = gather_scene_results(local, dist, dst=0)
        rate, tmax = job_throughput(float(sum(SCENES[s] for s in mine)), 1.0 + rank, dist)
        if rank == 0:
            out_q.put(({k: v.clone() for k, v in merged.items()}, rate, tmax, mine))
        else:
            assert merged is None
            out_q.put((None, rate, tmax, mine))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_scenes_is_a_balanced_partition():
    for world in (1, 2, 3, 8):
        parts = shard_scenes(SCENES, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(SCENES)))
        loads = [sum(SCENES[i] for i in p) for p in parts]
        assert max(loads) - min(l for l in loads if l > 0 or world <= len(SCENES)) <= max(SCENES)
    assert shard_scenes([5, 5, 5, 5], 2) == [[0, 2], [1, 3]]  # equal sizes -> i mod G
    assert shard_scenes([], 4) == [[], [], [], []]
    with pytest.raises(ValueError):
        shard_scenes([1], 0)


def test_single_process_helpers_need_no_process_group():
    local = {0: torch.ones(2)}
    assert gather_scene_results(local) == local
    assert job_throughput(10.0, 2.0) == (5.0, 2.0)


@pytest.mark.timeout(120)
def test_two_ranks_gloo_gather_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    merged = next(g[0] for g in got if g[0] is not None)
    assert sorted(merged) == list(range(len(SCENES)))
    for s in range(len(SCENES)):
        assert torch.equal(merged[s], scene_result(s)), f"scene {s} differs between sharded and single-process runs"
    # throughput: all units / slowest rank (rank 1 reported 2.0 s)
    for _, rate, tmax, _mine in got:
        assert tmax == 2.0 and rate == sum(SCENES) / 2.0
    assert sorted(i for g in got for i in g[3]) == list(range(len(SCENES)))
