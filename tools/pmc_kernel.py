#!/usr/bin/env python3
"""Developer tool: per-kernel means of the counters of ONE rocprofv3 --pmc pass.
usage: python tools/pmc_kernel.py <dir> [kernel-name-substring ...]"""
import collections, csv, glob, re, sys
d = sys.argv[1]
subs = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
        if subs and not any(s in name for s in subs):
            continue
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, c in sorted(acc.items()):
    n = max(len(v) for v in c.values())
    print(f"{name}  ({n} dispatches)")
    for k, v in sorted(c.items()):
        print(f"    {k:28s} mean {sum(v)/len(v):14.1f}")
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        tot = sum(wc) / len(wc)
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM"):
            if k in c:
                print(f"    {k + ' / WAVE_CYCLES':38s} {sum(c[k])/len(c[k])/tot:6.3f}")
