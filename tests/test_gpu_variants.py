"""Kernel variants that are selected by environment variables (read once per process) are exercised by
re-running a slice of the parity suite in a child process with the variable set:

  SE3_NO_PAIR=1     single-wavefront edge kernel instead of the wave-pair kernel for C = 64
  SE3_PG_SINGLE=1   one row per wavefront in the parameter-gradient kernel (what odd frame counts use)
  SE3_NO_T24=1      T and U as packed hi/lo words instead of the 3-byte row format (what C < 64 always uses)
  SE3_OVERLAP=1     backward branches on two streams at every size (default since round 5: never -- the fork lost its A/B,
                    profiles/r05_no_fork_ab.txt)
  SE3_BWD_BRANCH_ORDER=1  backward kernels branch by branch instead of writers first
  SE3_DX_PATH=1     feature gradient edge-major (edge_dx.hip) wherever it is implemented (the default decides by the bytes
                    either form moves -- down-convolutions and sparse levels only -- so the rest of the suite runs the U form)
  SE3_EDGE_STREAM=1 the chunk-stream forms of the edge kernel (round 6: resident workgroups, the chunk pipeline running across
                    item boundaries; the wave-pair form for 64-channel rows and the single-wavefront form for 32-channel
                    rows, two frames per item) at every size -- by default they take levels of 4 096 items and up, so the
                    small parity cases would never reach them
  SE3_NN_KG=2       the dense products over 3-byte rows of under-filled levels with two k groups per workgroup (round 5; lost
                    its A/B, profiles/r05_nn_kgroups_ab.txt)

(The row-sliced schedule of round 5 -- SE3_SLICE_MB, SE3_SLICE_STREAMS -- lost its A/B at every slice size,
profiles/r05_slice_ab.txt, and was removed in round 6.  The merged backward kernel, the fused edge + contraction kernel and the chunk-stream kernels of rounds 1-2 lost their A/B
measurements -- profiles/r02_levels_fused.txt, r02_stream_kernel_ab.txt, r03_merged_backward_and_stash_ab.txt -- and were
removed in round 3; so were round 3's in-kernel reductions, profiles/r03_in_kernel_reduction_ab.txt.)

One child at a time; each child is an ordinary `pytest -m gpu` run over the golden / random-shape / headline
tests of tests/test_gpu_parity.py.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# every switch acts on the split-bf16 kernels only: the children run the default arithmetic mode's cases of the slice
# (round 5: eight children over all three modes were 318 s of the 547 s suite)
SLICE = "(golden or random_shapes or headline_subset or features_only or empty_rows) and bf16x3 and not t16"
# independent switches share a child
VARIANTS = ["SE3_NO_PAIR,SE3_PG_SINGLE", "SE3_NO_T24", "SE3_BWD_BRANCH_ORDER,SE3_NN_KG=2", "SE3_OVERLAP",
            "SE3_DX_PATH=1,SE3_EDGE_STREAM=1"]


@pytest.mark.gpu
@pytest.mark.parametrize("var", VARIANTS)
def test_variant_passes_parity_slice(var):
    env = dict(os.environ)
    for part in var.split(","):
        name, _, value = part.partition("=")
        env[name] = value or "1"
    proc = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
         "-k", SLICE, "-p", "no:cacheprovider"],
        cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (proc.stdout + proc.stderr)[-2000:]
    assert proc.returncode == 0, f"{var}: parity slice failed\n{tail}"
    assert " passed" in proc.stdout, tail
