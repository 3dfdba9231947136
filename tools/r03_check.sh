#!/bin/bash
# One GPU-box call that refreshes a round's evidence (run through gpurun from the repo root):
#   tools/r03_check.sh <tag> [quick]
# full GPU parity suite, default bench line, 2-rank self-spawn smoke (gloo, ranks share the GPU), rocprofv3 kernel stats
# of the bench command (untruncated CSV), conv census.
set -u
TAG=${1:-r03}; QUICK=${2:-}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" | tee -a $OUT/rc.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.log
if [ -z "$QUICK" ]; then
  MG_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --batch-per-gpu 2 --no-cpu-baseline > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "bench_gloo2 rc=$?" | tee -a $OUT/rc.log
  timeout 300 python bench.py --batch-per-gpu 4 --no-cpu-baseline > $OUT/bench_bs4.json 2> $OUT/bench_bs4.err; echo "bench_bs4 rc=$?" | tee -a $OUT/rc.log
  MG_LEGACY_WEIGHTS=1 timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_legacy_weights.json 2> $OUT/bench_legacy_weights.err; echo "bench_legacy rc=$?" | tee -a $OUT/rc.log
  MG_NO_PAIR=1 timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_nopair.json 2> $OUT/bench_nopair.err; echo "bench_nopair rc=$?" | tee -a $OUT/rc.log
  timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_again.json 2> $OUT/bench_again.err; echo "bench_again rc=$?" | tee -a $OUT/rc.log
  timeout 300 python tools/conv_census.py > $OUT/conv_census.txt 2> $OUT/conv_census.err; echo "census rc=$?" | tee -a $OUT/rc.log
fi
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/rc.log
cd $R
if [ -z "$QUICK" ]; then timeout 600 bash tools/pmc_step.sh ${3:-unknown} > $OUT/pmc_step.log 2>&1; echo "pmc_step rc=$?" | tee -a $OUT/rc.log; cp gpurun_out/pmc_step/conv_traffic.json $OUT/ 2>/dev/null; rm -rf gpurun_out/pmc_step/FETCH_SIZE gpurun_out/pmc_step/WRITE_SIZE; fi
F=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$F" ] && cp "$F" $OUT/kernel_stats.csv
rm -rf $OUT/prof
tail -3 $OUT/pytest_gpu.log; cut -c1-1200 $OUT/bench.json; [ -f $OUT/bench_gloo2.json ] && cut -c1-400 $OUT/bench_gloo2.json; cat $OUT/rc.log
