#!/bin/bash
# One GPU-box call:  tools/r04_quick.sh <tag>  -- kernel / trainer parity tests, a short bench line and its kernel stats (no CPU baseline).
set -u
TAG=${1:-r04q}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trainer.py tests/test_gpu_fullsize.py -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/rc.log
timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-extra > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.log
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-extra > $OUT/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/rc.log
cd $R
F=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats.csv; rm -rf $OUT/prof
tail -4 $OUT/pytest_gpu.log | cut -c1-600; cut -c1-1500 $OUT/bench.json; grep -i "stats_stage" $OUT/kernel_stats.csv | cut -c1-300; cat $OUT/rc.log
