#!/usr/bin/env python3
"""Static instruction mix per kernel of a HIP source (hipcc -S, gfx950): a quick look at what a kernel's
inner loop is made of.  usage: tools/isa_mix.py se3conv3d_amd/csrc/edge_bf16.hip [name-filter]"""
import collections, re, subprocess, sys, os

def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = "/tmp/isa_mix.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-I" + root + "/include",
                    "-I" + root + "/se3conv3d_amd/csrc", "-S", "--cuda-device-only", "-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    name, mix, ops = None, None, None
    def flush():
        if name and filt in name and sum(mix.values()) > 50:
            print(name)
            print("  ", dict(mix))
            print("  ", ops.most_common(14))
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush()
            name, mix, ops = m.group(1), collections.Counter(), collections.Counter()
            continue
        if name is None or not line.startswith("\t"):
            continue
        t = line.strip()
        if not t or t[0] in ".;":
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"): mix["mfma"] += 1
        elif op.startswith("v_"): mix["valu"] += 1; ops[op] += 1
        elif op.startswith("ds_"): mix["lds"] += 1
        elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): mix["vmem"] += 1
        elif op.startswith("s_waitcnt"): mix["wait"] += 1
        elif op.startswith("s_"): mix["salu"] += 1
    flush()

if __name__ == "__main__":
    main()
