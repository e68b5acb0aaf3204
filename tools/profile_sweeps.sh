set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in C4 C5; do
  rm -rf $R/gpurun_out/ks_$C; mkdir -p $R/gpurun_out/ks_$C
  timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_$C -o ks --output-format csv -- python3 $R/bench.py --config $C --only --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/ks_$C/run.log 2>&1
  for K in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc/${C}_$K; mkdir -p $R/gpurun_out/pmc/${C}_$K
    timeout -k 10 280 rocprofv3 --pmc $K --kernel-trace -d $R/gpurun_out/pmc/${C}_$K -o p --output-format csv -- python3 $R/bench.py --config $C --only --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc/${C}_$K/run.log 2>&1
  done
  echo "$C done"
done
cd $R
python tools/trace_overlap.py gpurun_out/ks_C5 24 > gpurun_out/r03_C5_two_level_trace.txt
for C in C4 C5; do python bench.py --config $C --only --steps 10 --warmup 2 > gpurun_out/r03_bench_$C.json 2> gpurun_out/r03_bench_$C.err; done
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
