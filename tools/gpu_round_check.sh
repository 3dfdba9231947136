#!/bin/bash
# One GPU-box call that refreshes the round's evidence (run through gpurun from the repo root):
#   new-kernel parity tests, input-pipeline microbench, the default bench line, rocprofv3 kernel stats of the same command.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r01b; mkdir -p $OUT
cd $R
timeout 300 python -m pytest tests/test_gpu_inputs.py -q -m gpu -x > $OUT/pytest_inputs.log 2>&1; echo "pytest_inputs rc=$?" | tee -a $OUT/rc.log
timeout 120 python tools/bench_inputs.py > $OUT/bench_inputs.json 2> $OUT/bench_inputs.err; echo "bench_inputs rc=$?" | tee -a $OUT/rc.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.log
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/rc.log
cd $R
F=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$F" ] && head -60 "$F" | cut -c1-260 > $OUT/kernel_stats_top.csv
rm -rf $OUT/prof/*kernel_trace.csv $OUT/prof/*agent_info.csv 2>/dev/null
cat $OUT/bench_inputs.json; tail -3 $OUT/pytest_inputs.log; cat $OUT/bench.json | cut -c1-1500
