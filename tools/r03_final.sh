#!/bin/bash
# Final evidence at the round's last commit: GPU suite, the bench line, rocprofv3 kernel stats, HBM traffic PMC passes (so that bench.py's
# roofline.traffic is measured on the kernel sources it runs on).     bash tools/r03_final.sh <commit>
set -u
COMMIT=${1:-unknown}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03final; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" | tee -a $OUT/rc.log
timeout 700 bash tools/pmc_step.sh $COMMIT > $OUT/pmc_step.log 2>&1; echo "pmc_step rc=$?" | tee -a $OUT/rc.log
cp gpurun_out/pmc_step/conv_traffic.json $OUT/ 2>/dev/null; cp gpurun_out/pmc_step/conv_traffic.json profiles/r03_conv_traffic.json 2>/dev/null; rm -rf gpurun_out/pmc_step/FETCH_SIZE gpurun_out/pmc_step/WRITE_SIZE
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.log
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/rc.log
cd $R
F=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats.csv; rm -rf $OUT/prof
tail -3 $OUT/pytest_gpu.log; cut -c1-400 $OUT/bench.json; cat $OUT/rc.log
