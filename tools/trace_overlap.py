#!/usr/bin/env python3
"""Developer tool: from a rocprofv3 --kernel-trace CSV, the timeline of the LAST bench step of the barycentric
configuration: per kernel name start/end relative to the step, total busy time and the time two kernels overlap.
usage: python tools/trace_overlap.py <dir-with-*_kernel_trace.csv> [n_kernels_from_end]"""
import csv
import glob
import sys

d = sys.argv[1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-tail:]
t0 = int(rows[0]["Start_Timestamp"])
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?'):>3}  {r['Kernel_Name'][:70]}")
    ev += [(s, 1), (e, -1)]
ev.sort()
busy = over = 0
depth = 0
last = 0
for t, dlt in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        over += t - last
    depth += dlt
    last = t
print(f"span {(ev[-1][0])/1e3:.1f} us, busy {busy/1e3:.1f} us, >=2 kernels in flight {over/1e3:.1f} us")
