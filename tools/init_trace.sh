#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/init_trace.sh C2   -> timeline of the last init of the bench config
R=$GRAFT_REPO_ROOT
C=${1:-C2}
mkdir -p $R/gpurun_out/r04/itrace_$C
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace -d $R/gpurun_out/r04/itrace_$C -o t --output-format csv -- python3 $R/bench.py --config $C --only --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r04/itrace_$C/run.log 2>&1
python3 - <<PY > $R/gpurun_out/r04/init_timeline_$C.txt
import csv, glob
f = sorted(glob.glob("$R/gpurun_out/r04/itrace_$C/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fills = [i for i, r in enumerate(rows) if "rbf_fill" in r["Kernel_Name"]]
start = fills[-1]
t0 = int(rows[start]["Start_Timestamp"])
agg = {}
prev_end = t0
out = []
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0][:44]
    gap = (s - prev_end) / 1e3
    prev_end = e
    a = agg.setdefault(nm, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += max(gap, 0.0)
    if "rbf_eval" in nm or "tl_coarse" in nm: break
print("from the last fill to the first sweep kernel: %.1f us" % ((prev_end - t0) / 1e3))
for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-46s x%4d  busy %9.1f us   idle-before %8.1f us" % (nm, a[0], a[1], a[2]))
PY
cat $R/gpurun_out/r04/init_timeline_$C.txt
rm -rf $R/gpurun_out/r04/itrace_$C
