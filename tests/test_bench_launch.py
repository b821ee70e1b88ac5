"""`bench.py --gpus N` started plainly must launch N ranks itself (as child processes, before any GPU call in the
parent) and relay rank 0's JSON line.  Rehearsed here on CPU: `--dry-run` runs the same sharding, barrier,
MAX-over-ranks timing and result gather over gloo, without the HIP library."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], capture_output=True, text=True,
                       timeout=280, env=env, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


@pytest.mark.timeout(300)
def test_gpus_2_forks_two_ranks_and_reports_them():
    p, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout  # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["dry_run"] is True
    assert rec["scaling"] == "weak" and rec["scenes"] == [0, 1]  # one scene per rank, gathered to rank 0
    assert rec["config"]["workload"] == "headline"


@pytest.mark.timeout(300)
def test_gpus_8_dry_run_one_scene_per_rank():
    """BASELINE configuration 5 (ScanNet scenes sharded by scene over 8 GPUs, no collectives) as far as a box without
    GPUs can rehearse it: the parent forks 8 ranks, every rank gets exactly one scene, the 8 per-scene results reach
    rank 0 through the one gather of the job, and the ranks split the host's cores instead of taking 8 threads each."""
    p, lines = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run", "--workload", "scannet150k_f1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "weak" and rec["config"]["workload"] == "scannet150k_f1"
    assert rec["scenes"] == list(range(8))                       # every scene exactly once
    assert sorted(rec["scene_checksums"]) == [str(i) for i in range(8)]
    assert [rec["scene_checksums"][str(i)] for i in range(8)] == [float(i) for i in range(8)]  # rank r sent scene r's record
    cores = len(os.sched_getaffinity(0))
    assert 1 <= rec["config"]["cpu_threads_per_rank"] <= max(1, cores // 8)


@pytest.mark.timeout(120)
def test_single_rank_dry_run_needs_no_launcher():
    p, lines = _run(["--dry-run", "--steps", "2", "--workload", "scannet150k_f1"])
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(lines[-1])
    assert rec["n_gpus"] == 1 and rec["config"]["workload"] == "scannet150k_f1"


@pytest.mark.timeout(300)
def test_failing_rank_fails_the_parent():
    # --gpus 2 under a WORLD_SIZE of 1 is a usage error inside the rank; started plainly the parent launches two
    # ranks and a rank that exits non-zero must surface as a non-zero exit of the parent
    p, _ = _run(["--gpus", "2", "--steps", "1", "--dry-run"], {"SE3_BENCH_FAIL_RANK": "1"})
    assert p.returncode != 0


def test_workloads_match_the_baseline_configurations():
    sys.path.insert(0, ROOT)
    from se3conv3d_amd import workloads as W

    h = W.WORKLOADS["headline"]
    assert (h["points"], h["frames"], h["degree"], h["widths"][0]) == (65536, 2, 32, 64)
    assert W.WORKLOADS["scannet150k_f1"]["frames"] == 1 and W.WORKLOADS["scannet150k_f1"]["fixed_axis"] == 2
    assert W.WORKLOADS["dfaust_f2"]["pca"] and W.WORKLOADS["dfaust_f2"]["clouds"] == 32
    assert W.WORKLOADS["dfaust_f4"]["frames"] == 4
    # 3.30 GB algorithmic per headline layer, and the per-launch split sums to it
    n, e = 65536, 2048498
    own = W.stage_owned_bytes(n, e, 2, 64)
    assert sum(own.values()) == W.layer_bytes(n, e, 2, 64) == 3304053280
    assert own["edge_param_grad"] == 1076491152
