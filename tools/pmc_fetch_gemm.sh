set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmcq/a $R/gpurun_out/pmcq/b
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcq/a -o p --output-format csv -- python3 $R/tools/bench_gemm_one.py > $R/gpurun_out/pmcq/a/run.log 2>&1
GSL_SINTERP_NO_SUPERTILE=1 timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcq/b -o p --output-format csv -- python3 $R/tools/bench_gemm_one.py > $R/gpurun_out/pmcq/b/run.log 2>&1
for d in a b; do echo "== $d"; cat $R/gpurun_out/pmcq/$d/run.log | tail -2; python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/pmcq/$d/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "streamk" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
            print(r["Kernel_Name"][:60], r["Counter_Value"], "KB raw -> x2 =", float(r["Counter_Value"])*2*1024/1e9, "GB")
PY
done
