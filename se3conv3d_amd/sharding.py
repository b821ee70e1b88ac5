"""Scene sharding across the GPUs of a node.

The operator has no cross-scene dependence: grid keys carry the batch id
(custom_ops/ball_query/grid_utils.cuh:79-93 of the reference), so whole scenes are the unit of
distribution.  One process per GPU, parameters replicated, no collective on the data path; the only
communication is a result gather / the benchmark's timing reduction (RCCL over xGMI when the
process group is "nccl", gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch


def shard_scenes(scene_sizes: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-first bin packing of scenes (by point count) onto ``world_size`` ranks --
    the same balancing idea as the reference's ScanNetMaxPtsSampler (data_sets/loaders/ScanNet.py:447-503),
    applied across GPUs instead of across batches.  Deterministic: ties go to the lower rank, and every
    rank's list is returned in ascending scene id.  With equal sizes this is ``i mod G``."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    loads = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    order = sorted(range(len(scene_sizes)), key=lambda i: (-int(scene_sizes[i]), i))
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += int(scene_sizes[i])
    return [sorted(v) for v in out]


def gather_scene_results(local: Dict[int, torch.Tensor], dist=None, dst: int = 0):
    """The "trivial result gather": every rank contributes ``{scene id: tensor}``; rank ``dst`` receives
    the merged dict (other ranks get None).  Tensors are moved through CPU pickles, which is fine for a
    once-per-job gather; nothing here runs inside the timed region."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(local)
    payload = {k: v.detach().cpu() for k, v in local.items()}
    bucket = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(payload, bucket, dst=dst)
    if dist.get_rank() != dst:
        return None
    merged: Dict[int, torch.Tensor] = {}
    for part in bucket:
        for k, v in part.items():
            if k in merged:
                raise RuntimeError(f"scene {k} was computed by two ranks")
            merged[k] = v
    return merged


def job_throughput(units_local: float, seconds_local: float, dist=None, device=None):
    """Whole-job rate: units summed over ranks / slowest rank's time (MAX all-reduce)."""
    if dist is None or not dist.is_initialized():
        return units_local / seconds_local, seconds_local
    t = torch.tensor([seconds_local], dtype=torch.float64, device=device)
    u = torch.tensor([units_local], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / float(t.item()), float(t.item())
