#!/usr/bin/env python3
"""The top of a rocprofv3 `*_kernel_stats.csv` (--kernel-trace --stats --output-format csv) in one screen: total kernel
time, then per kernel name (cut to 100 characters) calls, average duration, share.   usage: tools/kernel_stats_top.py <csv>"""
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))
tot=sum(int(r[2]) for r in rows[1:])
print("total ns",tot)
for r in rows[1:45]:
    print(f"{r[0][:100]:100s} calls {r[1]:>5s} avg {float(r[3])/1e3:8.1f} us  {r[4]:>6s}%")
