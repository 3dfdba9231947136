#!/bin/bash
# round-4 GPU call: software-pipelined halo main loop A/B, in-painting branch after weight caching, multirank additions
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04d; mkdir -p $OUT; cd $R
timeout 900 python tools/ab_halo_ring.py > $OUT/halo_pipe_ab.txt 2>&1; echo "halo_ab rc=$?" | tee -a $OUT/rc.log
timeout 600 python -m pytest tests/test_inpaint.py tests/test_gpu_multirank.py -q -m gpu -x > $OUT/pytest_a.log 2>&1; echo "pytest_a rc=$?" | tee -a $OUT/rc.log
timeout 300 python tools/bench_inpaint.py > $OUT/bench_inpaint.txt 2>&1; echo "bench_inpaint rc=$?" | tee -a $OUT/rc.log
grep -v "^Network\|amdgpu.ids" $OUT/halo_pipe_ab.txt | cut -c1-330; tail -4 $OUT/pytest_a.log | cut -c1-300; tail -2 $OUT/bench_inpaint.txt | cut -c1-300
