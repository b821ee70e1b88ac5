"""CPU: the C-ABI library builds, loads and exports every symbol include/se3conv.h declares.
No compute entry point is called here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "se3conv.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^(?:int|int64_t|size_t|const char\*)\s+(se3\w+)\s*\(", text, flags=re.M)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ["se3conv_fwd", "se3conv_bwd", "se3_ball_query_count", "se3_ball_query_store", "se3_feat_basis_proj",
                 "se3_feat_basis_proj_grad", "se3_compute_keys", "se3_rot_tensors", "se3_csr_transpose"]:
        assert must in syms


def test_library_exports_every_declared_symbol(built_library):
    from se3conv3d_amd import _lib

    lib = C.CDLL(built_library)
    syms = declared_symbols()
    assert set(syms) == set(_lib.SIGNATURES), "ctypes signature table and header disagree"
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in se3conv.h but not exported"
    header_version = int(re.search(r"#define SE3_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert _lib.load().se3_abi_version() == _lib.ABI_VERSION == header_version   # the binding refuses a library of another version


def test_host_side_argument_checks(built_library):
    from se3conv3d_amd import _lib

    lib = _lib.load()
    null = C.c_void_p(0)
    assert lib.se3_compute_keys(null, null, null, null, null, -1, null, null) == -1
    assert lib.se3_compute_keys(null, null, null, null, null, 4, null, null) == -1
    assert lib.se3_ball_query_count(null, null, null, null, null, null, 0.0, 1, 1, null, 0, null, null) == -1
    bad = _lib.Se3Shape(10, 10, 10, 0, 1, 8, 8, 32)  # f_in = 0
    assert lib.se3conv_fwd_workspace_bytes(C.byref(bad), 1) == 0
    k16 = _lib.Se3Shape(10, 10, 10, 1, 1, 8, 8, 16)  # K != 32: slices of 32 inside the library -- a valid shape,
    args = [null] * 12 + [C.byref(k16), null, null, null, 0, null]  # null arguments are what is wrong here
    assert lib.se3conv_fwd(*args) == -1
    assert lib.se3conv_fwd_workspace_bytes(C.byref(k16), 1) > 0 and lib.se3conv_bwd_workspace_bytes(C.byref(k16), 1, 1, 0) > 0
    huge = _lib.Se3Shape(1 << 31, 10, 10, 2, 1, 8, 8, 32)  # row ids are int32 inside the kernels
    args = [C.c_void_p(16)] * 12 + [C.byref(huge), C.c_void_p(16), null, C.c_void_p(16), 1 << 62, null]
    assert lib.se3conv_fwd(*args) == -2
    assert lib.se3_feat_basis_proj(null, null, null, null, 0, 0, 0, 8, 12, null, null) == -2
    assert b"workspace" in lib.se3_error_string(-3)
    ok = _lib.Se3Shape(1000, 1000, 16000, 2, 2, 64, 64, 32)
    assert lib.se3conv_fwd_workspace_bytes(C.byref(ok), 0) > 1000 * 2 * 64 * 32 * 4
    assert lib.se3conv_bwd_workspace_bytes(C.byref(ok), 1, 1, 1) > 0
    assert lib.se3_ball_query_workspace_bytes(1000, 1000) > 0


def test_intermediate_format_query(built_library):
    """Host-only entry: which row format the operator picks for T / U / grad_T of a shape (bench.py's traffic model)."""
    from se3conv3d_amd import _lib

    lib = _lib.load()
    q = lambda shp, which: lib.se3conv_intermediate_bytes_per_element(C.byref(shp), which)
    headline = _lib.Se3Shape(65536, 65536, 2_000_000, 2, 2, 64, 64, 32, _lib.PRECISIONS["bf16x3"])
    assert [q(headline, w) for w in range(3)] == [3, 3, 4]
    exact = _lib.Se3Shape(65536, 65536, 2_000_000, 2, 2, 64, 64, 32, _lib.PRECISIONS["fp32"])
    assert [q(exact, w) for w in range(3)] == [4, 4, 4]
    narrow = _lib.Se3Shape(4096, 4096, 60_000, 1, 1, 32, 32, 32, _lib.PRECISIONS["bf16x3"])  # single-wavefront kernel, one channel per lane
    assert [q(narrow, w) for w in range(3)] == [3, 3, 4]
    between = _lib.Se3Shape(4096, 4096, 60_000, 1, 1, 48, 48, 32, _lib.PRECISIONS["bf16x3"])  # two channels per lane: packed words
    assert [q(between, w) for w in range(3)] == [4, 4, 4]
    assert q(headline, 3) < 0 and q(_lib.Se3Shape(10, 10, 10, 0, 1, 8, 8, 32, 1), 0) < 0
    # exact bytes per row; the third arithmetic mode keeps T / U of 64-channel rows (even frame count) in the 2.25-byte block format
    rb = lambda shp, which: lib.se3conv_intermediate_row_bytes(C.byref(shp), which)
    assert [rb(headline, w) for w in range(3)] == [64 * 32 * 3, 64 * 32 * 3, 64 * 32 * 4]
    t16 = _lib.Se3Shape(65536, 65536, 2_000_000, 2, 2, 64, 64, 32, _lib.PRECISIONS["bf16x3_t16"])
    assert [rb(t16, w) for w in range(3)] == [64 * 72, 64 * 72, 64 * 32 * 4] and [q(t16, w) for w in range(3)] == [2, 2, 4]   # grad_T: opt-in (SE3_T16_GT)
    wide16 = _lib.Se3Shape(4096, 4096, 60_000, 2, 2, 128, 256, 32, _lib.PRECISIONS["bf16x3_t16"])   # c_out = 256: grad_T by the tiled GEMM, packed words
    assert [rb(wide16, w) for w in range(3)] == [128 * 72, 256 * 72, 128 * 32 * 4]
    wide16b = _lib.Se3Shape(4096, 4096, 60_000, 2, 2, 128, 64, 32, _lib.PRECISIONS["bf16x3_t16"])
    assert [rb(wide16b, w) for w in range(3)] == [128 * 72, 64 * 72, 128 * 32 * 4]
    narrow16 = _lib.Se3Shape(4096, 4096, 60_000, 1, 1, 32, 32, 32, _lib.PRECISIONS["bf16x3_t16"])   # not implemented there: as bf16x3
    assert [rb(narrow16, w) for w in range(3)] == [32 * 32 * 3, 32 * 32 * 3, 32 * 32 * 4]
    odd16 = _lib.Se3Shape(4096, 4096, 60_000, 1, 1, 64, 64, 32, _lib.PRECISIONS["bf16x3_t16"])      # F = 1 at 64 channels: single-wavefront kernel
    assert [rb(odd16, w) for w in range(3)] == [64 * 32 * 3, 64 * 32 * 3, 64 * 32 * 4]


def test_workspace_queries_of_degenerate_and_wide_shapes(built_library):
    """Host-only arithmetic behind the workspace queries (row-range counts of the weight-gradient GEMM): zero rows (a
    division by the range count once took the process down with SIGFPE) and wide layers on many rows, whose ranges must
    stay within reach of one launch's 32-bit operand offsets (512 -> 256 channels beyond ~32 k rows)."""
    from se3conv3d_amd import _lib

    lib = _lib.load()
    assert lib.se3_linear_wgrad_workspace_bytes(0, 8, 16) == 8 * 16 * 4          # one (empty) range
    assert lib.se3_linear_wgrad_workspace_bytes(1, 5, 32) >= 5 * 32 * 4
    for prec in ("bf16x3", "fp32"):
        empty = _lib.Se3Shape(0, 0, 0, 2, 2, 64, 64, 32, _lib.PRECISIONS[prec])
        assert lib.se3conv_bwd_workspace_bytes(C.byref(empty), 1, 1, 0) > 0
        wide = _lib.Se3Shape(21000, 21000, 250_000, 2, 2, 512, 256, 32, _lib.PRECISIONS[prec])
        small = _lib.Se3Shape(2000, 2000, 25_000, 2, 2, 512, 256, 32, _lib.PRECISIONS[prec])
        # 42 000 rows of 16 384 values: at least 2 row ranges (one range = 32 703 rows at most), so > 1x the weight count more
        w_bytes = 512 * 32 * 256 * 4
        assert lib.se3conv_bwd_workspace_bytes(C.byref(wide), 0, 1, 1) - lib.se3conv_bwd_workspace_bytes(C.byref(small), 0, 1, 1) >= w_bytes


def test_header_lists_every_environment_switch():
    """The header promises that ALL process-wide state is listed there: every getenv in the library's sources must be
    named in it (suffix variants are listed as `NAME (+ _SUFFIX)`)."""
    import glob

    header = open(HEADER).read()
    names = set()
    for path in glob.glob(os.path.join(ROOT, "se3conv3d_amd", "csrc", "*")):
        names |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(path).read()))
    assert names, "no switches found: the scan is broken"
    for name in sorted(names):
        base = [b for b in re.findall(r"[A-Z0-9_]{5,}", header) if name == b or (name.startswith(b) and name[len(b):].startswith("_"))]
        assert base, f"{name} is read from the environment but not listed in include/se3conv.h"
        if name not in header:
            assert name[len(max(base, key=len)):] in header, f"suffix of {name} not listed"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from se3conv3d_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.Se3LibraryError):
        _lib.load()


def test_cpu_tensors_are_rejected(built_library):
    """The product path has no CPU fallback: CPU tensors raise instead of being computed elsewhere."""
    import torch

    from se3conv3d_amd import ops

    pts = torch.rand(8, 3)
    bid = torch.zeros(8, dtype=torch.int32)
    with pytest.raises(ValueError):
        ops.ball_query(pts, pts, bid, bid, 0.5)


def test_no_product_import_of_oracle():
    """Nothing under se3conv3d_amd/ may import the oracle (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "se3conv3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("CPU oracle", ""), f


def test_library_issues_no_memset_calls():
    """A memset NODE of a captured graph faults on replay next to a live RCCL communicator on the HIP runtime PyTorch 2.10
    ships: the library zeroes buffers by kernels (launch_fill_words) and keeps its sorts off
    rocPRIM's one-sweep radix sort, which issues hipMemsetAsync per pass."""
    csrc = os.path.join(ROOT, "se3conv3d_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")):
            continue
        code = "\n".join(line.split("//")[0] for line in open(os.path.join(csrc, name)).read().splitlines())
        assert "hipMemsetAsync(" not in code and "hipMemset(" not in code, f"{name} calls hipMemset*"
        if name != "geometry.hip":
            assert "DeviceRadixSort" not in code, f"{name} sorts through hipcub::DeviceRadixSort directly"
    geo = open(os.path.join(csrc, "geometry.hip")).read()
    # the only direct use is inside sort_pairs_no_scratch (sizes up to rocPRIM's merge-sort limit)
    assert geo.count("hipcub::DeviceRadixSort::SortPairs(") == 2, "radix sorts must go through sort_pairs_no_scratch"
