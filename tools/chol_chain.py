#!/usr/bin/env python3
"""Developer tool: the per-panel chain of the factorisation from a rocprofv3 --kernel-trace CSV of `tools/chol_time.py N`:
average duration of the diagonal-block kernel, the row solve and the updates by K, launch gaps, and the diag-to-diag
period at the K = 128 nodes of the recursion (the number VERDICT r3 item 1 asks for).
usage: python tools/chol_chain.py <dir-with-*_kernel_trace.csv>"""
import csv, glob, re, sys
from collections import defaultdict

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last factorisation = kernels after the last chol_zero_info_kernel
last0 = max(i for i, r in enumerate(rows) if "chol_zero_info" in r["Kernel_Name"])
seq = []
for r in rows[last0:]:
    nm = r["Kernel_Name"]
    if "chol_diag_writeback" in nm:
        seq.append(("wb", int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        break
    key = "diag" if "chol_diag128" in nm else "trsm" if "chol_trsm" in nm else "gemm" if "gemm_minus" in nm else nm[:24]
    seq.append((key, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
dur = defaultdict(list)
for k, s, e in seq:
    dur[k].append((e - s) / 1e3)
for k, v in dur.items():
    print(f"{k:8s} launches {len(v):5d}  avg {sum(v)/len(v):8.2f} us  min {min(v):8.2f}  max {max(v):8.2f}  total {sum(v)/1e3:8.3f} ms")
gaps = [(seq[i + 1][1] - seq[i][2]) / 1e3 for i in range(len(seq) - 1)]
print(f"launch gaps: avg {sum(gaps)/len(gaps):.2f} us, total {sum(gaps)/1e3:.3f} ms over {len(gaps)} boundaries")
# panel p (0-based) is followed by the update with K = 128 * (lowest set bit of p + 1): the recursion's post-order
idx = [i for i, (k, _, _) in enumerate(seq) if k == "diag"]
byk = defaultdict(list)
for p, (a, b) in enumerate(zip(idx, idx[1:])):
    mid = [seq[i][0] for i in range(a + 1, b)]
    if mid != ["trsm", "gemm"]:
        continue
    K = 128 * ((p + 1) & -(p + 1))
    byk[K].append(((seq[b][1] - seq[a][1]) / 1e3, (seq[a][2] - seq[a][1]) / 1e3, (seq[a + 1][2] - seq[a + 1][1]) / 1e3,
                   (seq[a + 2][2] - seq[a + 2][1]) / 1e3))
for K in sorted(byk):
    per = byk[K]
    n = len(per)
    print(f"K = {K:5d} nodes: {n:3d}  diag-to-diag {sum(p[0] for p in per)/n:8.2f} us = diag {sum(p[1] for p in per)/n:6.2f} + trsm {sum(p[2] for p in per)/n:6.2f}"
          f" + update {sum(p[3] for p in per)/n:8.2f}")
span = (seq[-1][2] - seq[0][1]) / 1e3
print(f"factorisation span {span/1e3:.3f} ms")
