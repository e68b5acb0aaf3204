#!/bin/bash
# Round-end profiling on the GPU box: kernel stats per config + HBM counter passes.
# usage (on the box, from the repo root):  bash tools/profile_all.sh
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in C2 C3 C4 C5; do
  mkdir -p $R/gpurun_out/ks_$C
  timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_$C -o ks --output-format csv -- python3 $R/bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/ks_$C/run.log 2>&1
  echo "stats $C done"
done
for C in C2 C3 C5; do
  for K in FETCH_SIZE WRITE_SIZE; do
    mkdir -p $R/gpurun_out/pmc/${C}_$K
    timeout -k 10 280 rocprofv3 --pmc $K --kernel-trace -d $R/gpurun_out/pmc/${C}_$K -o p --output-format csv -- python3 $R/bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc/${C}_$K/run.log 2>&1
    echo "pmc $C $K done"
  done
done
