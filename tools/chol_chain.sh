#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/chol_chain.sh N [outfile]
R=$GRAFT_REPO_ROOT
N=${1:-4096}
OUT=${2:-$R/gpurun_out/r04/chol_chain_$N.txt}
case "$OUT" in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $R/gpurun_out/r04/trace_$N
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace -d $R/gpurun_out/r04/trace_$N -o t --output-format csv -- python3 $R/tools/chol_time.py $N > $R/gpurun_out/r04/trace_$N/run.log 2>&1
python3 $R/tools/chol_chain.py $R/gpurun_out/r04/trace_$N > $OUT 2>&1
rm -rf $R/gpurun_out/r04/trace_$N/*/*.csv.bak
cat $OUT
