#!/bin/bash
# developer tool: counters of bary_walk_kernel on C5, one counter per pass (with --kernel-trace only)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc_list.txt 2>&1
for K in "$@"; do
  mkdir -p $R/gpurun_out/pmcw/$K
  timeout -k 10 200 rocprofv3 --pmc $K --kernel-trace -d $R/gpurun_out/pmcw/$K -o p --output-format csv -- python3 $R/bench.py --config C5 --only --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcw/$K/run.log 2>&1 || echo "pmc $K FAILED"
  python3 - "$R/gpurun_out/pmcw/$K" "$K" <<'PY'
import csv, glob, sys, collections
d, k = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == k:
            acc[r["Kernel_Name"].split("(")[0][:30]].append(float(r["Counter_Value"]))
for n, v in acc.items():
    if "bary" in n or "cell_" in n or "unsort" in n:
        print(f"{k:34s} {n:32s} launches {len(v)}  per-launch {sum(v) / len(v):.4g}")
PY
done
