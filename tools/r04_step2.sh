#!/bin/bash
# round-4 GPU call: attention kernel + in-painting branch, forward-error probe
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04c; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "self_attention or tr16 or probe" > $OUT/pytest_attn.log 2>&1; echo "pytest_attn rc=$?" | tee -a $OUT/rc.log
timeout 600 python -m pytest tests/test_inpaint.py -q -m gpu > $OUT/pytest_inpaint.log 2>&1; echo "pytest_inpaint rc=$?" | tee -a $OUT/rc.log
timeout 300 python tools/bench_inpaint.py > $OUT/bench_inpaint.txt 2>&1; echo "bench_inpaint rc=$?" | tee -a $OUT/rc.log
timeout 700 python tools/grad_probe.py > $OUT/grad_probe.txt 2>&1; echo "probe rc=$?" | tee -a $OUT/rc.log
KSTATS_ARGS="--inpaint-orient" timeout 500 bash tools/kstats.sh inpaint "attention|Cijk|softmax|gemm" > $OUT/kstats_inpaint.txt 2>&1; echo "kstats rc=$?" | tee -a $OUT/rc.log
tail -15 $OUT/pytest_attn.log | cut -c1-400; tail -5 $OUT/pytest_inpaint.log | cut -c1-400; cat $OUT/bench_inpaint.txt | tail -3; head -14 $OUT/grad_probe.txt; tail -8 $OUT/kstats_inpaint.txt | cut -c1-250
