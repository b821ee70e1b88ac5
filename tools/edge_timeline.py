#!/usr/bin/env python3
"""Where a wavefront of the edge kernels spends its cycles: runs one full-resolution layer forward +
backward on the TIMELINE build of the library (s_memtime stamps at the phase boundaries of `edge_t_pair_bf16_kernel` and of
the pair form of `edge_param_grad_bf16_v2_kernel`, edge_bf16.hip) and prints the median duration of every segment over the
wavefronts, as cycles and as a share of the wavefront's life.

    SE3_LIB_SUFFIX=_tl SE3_CXXFLAGS=-DSE3_TIMELINE=1 python -m se3conv3d_amd.build      # build container
    SE3_LIB_SUFFIX=_tl python tools/edge_timeline.py [--workload headline] > profiles/r06_edge_timeline.txt   # GPU box

The stamped build is for SHARES, not for run time (every stamp is a scheduling barrier with an lgkmcnt(0))."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

WORDS = 64


def med(a):
    return float(np.median(a)) if len(a) else float("nan")


def report_edge_t(name, rec):
    rec = rec[rec[:, 10] != 0]  # records that were written
    if not len(rec):
        print(f"{name}: no records")
        return
    r = rec.astype(np.int64)
    life = (r[:, 9] - r[:, 0]) & 0xffffffff
    print(f"== {name}: {len(r)} wavefronts; life (entry -> stores issued) median {med(life):.0f} cycles, mean {life.mean():.0f}; "
          f"stores retired {med((r[:, 10] - r[:, 9]) & 0xffffffff):.0f} later")
    slot_occupancy(r[:, 11], r[:, 12], r[:, 0], r[:, 10])
    if int(r[:, 14].max()) > 8:  # resident workgroups (the chunk-stream kernel): entry / exit stamps and chunk counts only
        by_x = {}
        for x in range(8):
            m = r[:, 12] == x
            if m.any():
                t0 = r[m, 0].min()
                end = (r[m, 10] - t0) & 0xffffffff
                start = (r[m, 0] - t0) & 0xffffffff
                by_x[x] = (start, end, r[m, 14])
        for x, (start, end, ch) in sorted(by_x.items()):
            print(f"   XCD {x}: {len(end)} wavefronts; entered within {np.percentile(start, 99):.0f} cycles (99 %); finished at "
                  f"{np.percentile(end, 1):.0f} / {np.median(end):.0f} / {end.max()} cycles (1 % / median / last); chunks per wavefront "
                  f"{ch.min()} .. {ch.max()} (mean {ch.mean():.1f}); cycles per chunk {np.median((end - start) / ch):.0f}")
        return
    for nch in sorted(set(r[:, 14].tolist())):
        sel = r[r[:, 14] == nch]
        if len(sel) < 200 or nch > 5:
            continue
        lf = (sel[:, 9] - sel[:, 0]) & 0xffffffff
        tot = med(lf)
        rows = []

        def seg(label, a, b):
            d = (b - a) & 0xffffffff
            rows.append((label, med(d), float(d.mean())))

        seg("entry: arguments, MLP weights -> LDS, barrier", sel[:, 0], sel[:, 1])
        seg("item set-up issued (division, s_load of the row extents, centre record load)", sel[:, 1], sel[:, 2])
        seg("WAIT row extents", sel[:, 2], sel[:, 3])
        seg("neighbour-id loads issued (chunks 0, 1)", sel[:, 3], sel[:, 4])
        seg("WAIT ids of chunk 0", sel[:, 4], sel[:, 5])
        seg("record load of chunk 0 issued", sel[:, 5], sel[:, 6])
        prev = sel[:, 6]
        for j in range(int(nch)):
            c = sel[:, 16 + 8 * j: 24 + 8 * j]
            seg(f"chunk {j}: ids -> rows, bpermute, 16 feature gathers + next record issued", prev, c[:, 1])
            seg(f"chunk {j}: WAIT this chunk's record", c[:, 1], c[:, 2])
            seg(f"chunk {j}: descriptor, split, kernel-MLP MFMA, result in", c[:, 2], c[:, 3])
            seg(f"chunk {j}: GELU, hi/lo split, publish to LDS", c[:, 3], c[:, 4])
            seg(f"chunk {j}: BARRIER (partner wavefront)", c[:, 4], c[:, 5])
            seg(f"chunk {j}: WAIT feature words of k-step 0", c[:, 5], c[:, 6])
            seg(f"chunk {j}: LDS reads, fragments, 12 aggregation MFMAs issued", c[:, 6], c[:, 7])
            prev = c[:, 7]
        seg("loop exit", prev, sel[:, 7])
        seg("WAIT accumulators (MFMA drain)", sel[:, 7], sel[:, 8])
        seg("pack 3-byte rows + 32 stores issued", sel[:, 8], sel[:, 9])
        print(f"-- {int(nch)} chunks: {len(sel)} wavefronts, median life {tot:.0f} cycles, mean n_total {sel[:, 15].mean():.1f}")
        s_med = sum(x[1] for x in rows)
        waits = sum(x[1] for x in rows if "WAIT" in x[0] or "BARRIER" in x[0])
        for label, m, mean in rows:
            print(f"   {m:8.0f} cyc  {100 * m / s_med:5.1f} %   (mean {mean:7.0f})  {label}")
        print(f"   sum of medians {s_med:.0f} = {100 * s_med / tot:.0f} % of the median life; waits + barriers {100 * waits / s_med:.1f} %")


def slot_occupancy(hw_id, xcc, t_start, t_end):
    """How full the SIMDs were: wavefronts grouped by the SIMD they ran on (HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13;
    XCC_ID), per SIMD the wavefront-cycles over the span from its first entry to its last exit, and the idle time of a
    wavefront slot between one wavefront's end and the next one's entry."""
    simd = (xcc << 20) | (hw_id & 0xfff0)
    order = np.argsort(simd, kind="stable")
    simd, ts, te, wid = simd[order], t_start[order], t_end[order], (hw_id[order] & 0xf)
    bounds = np.flatnonzero(np.diff(simd)) + 1
    conc, gaps, per_simd, ends = [], [], [], []
    for lo, hi in zip(np.r_[0, bounds], np.r_[bounds, len(simd)]):
        s0 = ts[lo]
        a = ((ts[lo:hi] - s0 + 2 ** 31) & 0xffffffff) - 2 ** 31  # wrap-safe, relative to one wavefront of the SIMD
        b = ((te[lo:hi] - s0 + 2 ** 31) & 0xffffffff) - 2 ** 31
        span = b.max() - a.min()
        conc.append((b - a).sum() / span)
        if hi - lo <= 8:  # resident workgroups: when the SIMD's wavefronts finish, as fractions of its span
            ends.append(np.sort((b - a.min()) / span)[:4] if hi - lo >= 4 else np.full(4, np.nan))
        per_simd.append(hi - lo)
        w = wid[lo:hi]
        for slot in np.unique(w):
            m = w == slot
            o = np.argsort(a[m])
            g = a[m][o][1:] - b[m][o][:-1]
            gaps.extend(g[g > -1000].tolist())
    gaps = np.array(gaps)
    print(f"   SIMDs seen {len(conc)}, wavefronts per SIMD {np.mean(per_simd):.1f}; resident wavefronts per SIMD (wavefront-cycles / span): "
          f"mean {np.mean(conc):.2f}, 10 % / 90 % {np.percentile(conc, 10):.2f} / {np.percentile(conc, 90):.2f}")
    if ends:
        e = np.nanmean(np.array(ends), axis=0)
        print(f"   wavefronts of a SIMD finish at {e[0]:.2f} / {e[1]:.2f} / {e[2]:.2f} / {e[3]:.2f} of its span (mean over SIMDs)")
    if len(gaps):
        print(f"   a slot between two wavefronts: idle median {np.median(gaps):.0f} cycles, mean {gaps.mean():.0f}, 90 % {np.percentile(gaps, 90):.0f}")


PG_SEGS = [
    "entry: arguments, MLP weights -> LDS, barrier", "item set-up issued", "WAIT row extents",
    "centre record, ids, 32 grad_T row loads issued", "WAIT ids", "WAIT grad_T words (+ record load issued)",
    "grad_T fragments built, parked in LDS", "BARRIER image complete", "chunk: ids -> rows, 8 feature loads issued",
    "chunk: WAIT this chunk's record", "chunk: descriptor, next record issued, splits, descriptor image", "chunk: kernel MLP + GELU' (both frames)",
    "chunk: WAIT feature words", "chunk: fragments, gphi / d[A;beta] MFMAs, wave barrier", "BARRIER item done",
    "accumulator drain + workgroup reduction"]


def report_pg(rec):
    rec = rec[rec[:, 19] != 0]
    if not len(rec):
        print("edge_param_grad: no records")
        return
    r = rec.astype(np.int64)
    life = (r[:, 19] - r[:, 18]) & 0xffffffff
    print(f"== edge_param_grad_bf16_v2 (pair form): {len(r)} wavefronts; life median {med(life):.0f} cycles (min {life.min()}, max {life.max()}); "
          f"items / wavefront {r[:, 16].mean():.1f}, chunks / wavefront {r[:, 17].mean():.1f}")
    slot_occupancy(r[:, 20], r[:, 21], r[:, 18], r[:, 19])
    tot = r[:, :16].sum(axis=1)
    waits = 0.0
    for i, label in enumerate(PG_SEGS):
        share = 100 * r[:, i].sum() / tot.sum()
        per = r[:, i].sum() / max(1, r[:, 17].sum() if label.startswith("chunk") else r[:, 16].sum())
        unit = "chunk" if label.startswith("chunk") else "item"
        if i in (0, 15):
            per, unit = r[:, i].mean(), "wavefront"
        if "WAIT" in label or "BARRIER" in label:
            waits += share
        print(f"   {share:5.1f} %  {per:8.0f} cyc / {unit:9s}  {label}")
    print(f"   accounted {100 * tot.sum() / life.sum():.1f} % of the wavefronts' life; waits + barriers {waits:.1f} %")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="headline")
    ap.add_argument("--precision", default="bf16x3")
    args = ap.parse_args()
    import se3conv3d_amd as amd
    from se3conv3d_amd import _lib, workloads as W

    lib = _lib.load()
    if not hasattr(lib, "se3_timeline_set"):
        sys.exit("this library has no stamps: build and select the timeline variant (see the docstring)")
    lib.se3_timeline_set.restype = C.c_int
    lib.se3_timeline_set.argtypes = [C.c_void_p, C.c_uint32]
    amd.set_precision(args.precision)
    levels = W.build_stack(dict(W.WORKLOADS[args.workload]), torch.device("cuda", 0), seed=0, n_levels=1)
    lv = levels[0]
    cap = int(lv["n"]) * 4 + 4096
    buf = torch.zeros(3 * cap * WORDS, dtype=torch.int32, device="cuda")
    for _ in range(3):
        bench.step(levels[:1])
    torch.cuda.synchronize()
    assert lib.se3_timeline_set(buf.data_ptr(), cap) == 0
    bench.step(levels[:1])
    torch.cuda.synchronize()
    lib.se3_timeline_set(None, 0)
    rec = buf.cpu().numpy().view(np.uint32).reshape(3, cap, WORDS)
    print(f"# {args.workload}: level 0, N = {lv['n']}, E = {lv['e']}, precision {args.precision}; stamped build (shares, not run time)")
    prof = bench.profile_level(lib, lv, 3)
    print("# stage times of the STAMPED build (ms): " + ", ".join(f"{k} {v[0]:.3f}" for k, v in prof.items()))
    report_edge_t("edge_t_pair_bf16 forward", rec[0])
    report_edge_t("edge_t_pair_bf16 transposed", rec[1])
    report_pg(rec[2])


if __name__ == "__main__":
    main()
