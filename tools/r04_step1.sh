#!/bin/bash
# round-4 GPU call: statistics rework -- kernel tests, gradient probe, full-size parity, bench line
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04b; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "stats or norm or spade" > $OUT/pytest_stats.log 2>&1; echo "pytest_stats rc=$?" | tee -a $OUT/rc.log
timeout 700 python tools/grad_probe.py > $OUT/grad_probe.txt 2>&1; echo "probe rc=$?" | tee -a $OUT/rc.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -s > $OUT/pytest_fullsize.log 2>&1; echo "pytest_fullsize rc=$?" | tee -a $OUT/rc.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.log
tail -3 $OUT/pytest_stats.log; head -16 $OUT/grad_probe.txt; grep -n "passed\|failed\|bs 8\|losses\|full-width" $OUT/pytest_fullsize.log | cut -c1-1500; cut -c1-300 $OUT/bench.json
