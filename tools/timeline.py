#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run: for the last `n` kernels, start offset, duration, the idle gap in
front of each kernel (no kernel of the process running), and the total busy / idle time of that stretch.
usage: tools/timeline.py <dir with *_kernel_trace.csv> [n_last_kernels]"""
import csv, glob, sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", "")))
rows.sort()
rows = rows[-n:]
t0 = rows[0][0]
busy_end = rows[0][0]
idle = 0
for s, e, name, q in rows:
    gap = s - busy_end
    if gap > 0:
        idle += gap
    print("%10.3f ms  %9.3f ms  gap %8.3f  q%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, gap / 1e6, q, name))
    busy_end = max(busy_end, e)
print("span %.3f ms, idle %.3f ms" % ((busy_end - t0) / 1e6, idle / 1e6))
