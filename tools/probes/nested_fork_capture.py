#!/usr/bin/env python3
"""Probe: a fork from a forked stream inside a HIP graph capture, with torch streams / events only (no library call).
origin -> A -> B (B forked from A), joined B -> A -> origin, end capture.  Variants: events created before / inside the capture,
events that were already recorded eagerly before the capture (what a library that reuses its fork / join events does)."""
import faulthandler
import sys

import torch

faulthandler.enable()
mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
x = torch.zeros(1 << 20, device="cuda")
A, B = torch.cuda.Stream(), torch.cuda.Stream()
fork, join = torch.cuda.Event(), torch.cuda.Event()


def step():
    cur = torch.cuda.current_stream()
    A.wait_stream(cur)
    with torch.cuda.stream(A):
        x.add_(1.0)
        if mode in ("fresh", "flat"):
            f, j = torch.cuda.Event(), torch.cuda.Event()
        else:
            f, j = fork, join
        if mode != "flat":  # "flat": the control -- one level of fork only
            f.record(A)
            B.wait_event(f)
            with torch.cuda.stream(B):
                x[: 1 << 10].mul_(2.0)
            j.record(B)
            A.wait_event(j)
        x.add_(1.0)
    cur.wait_stream(A)


if mode == "reused":
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()  # the events get an eager record first
    torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    step()
print(mode, "captured", flush=True)
g.replay()
torch.cuda.synchronize()
print(mode, "replayed ok", float(x[0]), flush=True)
