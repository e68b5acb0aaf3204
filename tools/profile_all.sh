#!/bin/bash
# Round-end profiling on the GPU box: kernel stats per config + HBM / MFMA / VALU counter passes.
# usage (on the box, from the repo root):  bash tools/profile_all.sh
# (python3 directly after "--": the profiler's preloaded library initialises the GPU, no exec hop allowed;
#  counters in their own passes, with --kernel-trace only)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in C2 C3 C4 C5; do
  mkdir -p $R/gpurun_out/ks_$C
  timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_$C -o ks --output-format csv -- python3 $R/bench.py --config $C --only --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/ks_$C/run.log 2>&1
  echo "stats $C done"
done
for C in C2 C3 C4 C5; do
  for K in FETCH_SIZE WRITE_SIZE; do
    mkdir -p $R/gpurun_out/pmc/${C}_$K
    timeout -k 10 280 rocprofv3 --pmc $K --kernel-trace -d $R/gpurun_out/pmc/${C}_$K -o p --output-format csv -- python3 $R/bench.py --config $C --only --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc/${C}_$K/run.log 2>&1
    echo "pmc $C $K done"
  done
done
# pipe utilisation: MFMA busy of the factorisation (C3), VALU busy of the sweeps (C2 thin-plate, C5 barycentric)
for CK in "C3 SQ_VALU_MFMA_BUSY_CYCLES" "C3 GRBM_GUI_ACTIVE" "C3 SQ_ACTIVE_INST_VALU" "C3 SQ_BUSY_CYCLES" "C3 SQ_WAIT_INST_ANY" "C3 SQ_WAVE_CYCLES" "C4 SQ_ACTIVE_INST_VALU" "C4 SQ_BUSY_CYCLES" "C4 GRBM_GUI_ACTIVE" "C4 SQ_WAIT_INST_ANY" "C4 SQ_WAVE_CYCLES" "C2 SQ_ACTIVE_INST_VALU" "C2 SQ_BUSY_CYCLES" "C2 GRBM_GUI_ACTIVE" "C5 SQ_ACTIVE_INST_VALU" "C5 SQ_BUSY_CYCLES" "C5 GRBM_GUI_ACTIVE" "C5 SQ_WAIT_INST_ANY" "C5 SQ_WAVE_CYCLES"; do
  set -- $CK
  mkdir -p $R/gpurun_out/pmc/${1}_$2
  timeout -k 10 280 rocprofv3 --pmc $2 --kernel-trace -d $R/gpurun_out/pmc/${1}_$2 -o p --output-format csv -- python3 $R/bench.py --config $1 --only --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc/${1}_$2/run.log 2>&1 || echo "pmc $1 $2 FAILED"
  echo "pmc $1 $2 done"
done

# the per-panel chain of the factorisation and its levels (round 4)
for N in 4096 8192 16384; do bash $R/tools/chol_chain.sh $N $R/gpurun_out/chol_chain_$N.txt > /dev/null; echo "chain $N done"; done
rm -rf $R/gpurun_out/r04/trace_*
cd $R && python3 tools/chol_levels.py > gpurun_out/chol_levels.txt 2>&1 || echo "chol_levels FAILED"
