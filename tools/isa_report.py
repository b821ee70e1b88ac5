#!/usr/bin/env python3
"""Per-kernel compile report of a HIP source with the flags the library ships with (se3conv3d_amd/build.py): VGPRs,
spilled registers, scratch, LDS, waves per SIMD, and -- from the assembly -- instruction, branch and scratch-access
counts.  The round-2 findings of DESIGN.md section 4.2 (a branch per guarded load, a 64-bit division per item) were read
off exactly this.      usage: tools/isa_report.py se3conv3d_amd/csrc/edge_bf16.hip [name-filter] [extra -D flags ...]"""
import collections, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se3conv3d_amd import build  # the shipped FLAGS

def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    extra = sys.argv[3:]
    flags = [f for f in build.FLAGS if f not in ("-Wall",)] + extra
    asm = "/tmp/isa_report.s"
    res = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                          "-o", asm, src], capture_output=True, text=True)
    if res.returncode:
        sys.exit(res.stderr[-2000:])
    usage, cur = collections.OrderedDict(), None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = usage.setdefault(m.group(1), {})
        m = re.search(r"remark:\s+(VGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    counts, name = {}, None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            counts[name] = collections.Counter()
            continue
        if line.startswith(".Lfunc_end"):  # (not s_endpgm: the exit block may be laid out in front of the loop body)
            name = None
            continue
        if name is None or not line.startswith("\t"):
            continue
        op = line.strip().split()[0] if line.strip() else ""
        if not op or op[0] in ".;":
            continue
        c = counts[name]
        c["instructions"] += 1
        c["branches"] += op.startswith("s_cbranch") or op == "s_branch"
        c["scratch"] += op.startswith("scratch_")
        c["mfma"] += op.startswith("v_mfma")
        c["waitcnt"] += op == "s_waitcnt"
    print(f"{'kernel':70s} {'VGPR':>5s} {'spill':>5s} {'LDS':>6s} {'w/SIMD':>6s} {'instr':>6s} {'branch':>6s} {'scratch':>7s} {'mfma':>5s}")
    for k, u in usage.items():
        if filt not in k:
            continue
        c = counts.get(k, {})
        short = re.sub(r"^_ZN3se312_GLOBAL__N_1\d+", "", k)[:70]
        print(f"{short:70s} {u.get('VGPRs', 0):5d} {u.get('VGPRs Spill', 0):5d} {u.get('LDS Size', 0):6d} {u.get('Occupancy', 0):6d} "
              f"{c.get('instructions', 0):6d} {c.get('branches', 0):6d} {c.get('scratch', 0):7d} {c.get('mfma', 0):5d}")

if __name__ == "__main__":
    main()
