#!/bin/bash
# One GPU-box call (through gpurun from the repo root):  tools/r04_check.sh <tag>
#   full GPU parity suite, smoke(), the DEFAULT bench line (with its in-run rocprofv3 --pmc traffic passes, the fp32 configs[1] leg and the
#   CPU baseline: timed as the driver will run it), rocprofv3 kernel stats of the bench command.
set -u
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 200 python tools/inpaint_err_probe.py > $OUT/inpaint_err_probe.txt 2>&1; grep mean $OUT/inpaint_err_probe.txt
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" | tee -a $OUT/rc.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/rc.log
T0=$(date +%s); timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s" | tee -a $OUT/rc.log
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-extra > $OUT/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/rc.log
cd $R
F=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats.csv; rm -rf $OUT/prof
tail -4 $OUT/pytest_gpu.log | cut -c1-600; tail -1 $OUT/smoke.log; cut -c1-3000 $OUT/bench.json; cat $OUT/rc.log
