#!/bin/bash
# Round-3 evidence in one GPU-box call (run through gpurun from the repo root):  bash tools/r03_evidence.sh <commit>
#   full GPU suite, the default bench line (+ repeat), bs 4, configs[1] (generator forward fp32 bs 4), configs[4]-style (in-painting branch),
#   one-rank RCCL with every collective forced (two communicators vs MG_DP_ONE_GROUP=1), 2-rank gloo self-spawn smoke, rocprofv3 kernel
#   stats of the bench command, conv census, HBM traffic PMC passes, MFMA counters of the halo conv, launch A/B of the glue kernels.
set -u
COMMIT=${1:-unknown}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" | tee -a $OUT/rc.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_again.json 2> $OUT/bench_again.err; echo "bench_again rc=$?" | tee -a $OUT/rc.log
timeout 300 python bench.py --batch-per-gpu 4 --no-cpu-baseline > $OUT/bench_bs4.json 2> $OUT/bench_bs4.err; echo "bench_bs4 rc=$?" | tee -a $OUT/rc.log
timeout 300 python bench.py --mode gfwd --dtype fp32 --batch-per-gpu 4 --no-cpu-baseline > $OUT/gfwd_bs4_fp32.json 2> $OUT/gfwd.err; echo "gfwd rc=$?" | tee -a $OUT/rc.log
timeout 300 python bench.py --inpaint-orient --no-cpu-baseline > $OUT/bench_inpaint.json 2> $OUT/bench_inpaint.err; echo "inpaint rc=$?" | tee -a $OUT/rc.log
MG_DP_FORCE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29741 bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_rccl1_two_groups.json 2> $OUT/bench_rccl1_two_groups.err; echo "rccl1 two groups rc=$?" | tee -a $OUT/rc.log
MG_DP_FORCE=1 MG_DP_ONE_GROUP=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29742 bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_rccl1_one_group.json 2> $OUT/bench_rccl1_one_group.err; echo "rccl1 one group rc=$?" | tee -a $OUT/rc.log
MG_DETERMINISTIC=1 timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_deterministic.json 2> $OUT/bench_deterministic.err; echo "deterministic rc=$?" | tee -a $OUT/rc.log
MG_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --batch-per-gpu 2 --no-cpu-baseline > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "bench_gloo2 rc=$?" | tee -a $OUT/rc.log
timeout 300 python tools/conv_census.py > $OUT/conv_census.txt 2> $OUT/conv_census.err; echo "census rc=$?" | tee -a $OUT/rc.log
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/rc.log
cd $R
F=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats.csv; rm -rf $OUT/prof
timeout 700 bash tools/pmc_step.sh $COMMIT > $OUT/pmc_step.log 2>&1; echo "pmc_step rc=$?" | tee -a $OUT/rc.log
cp gpurun_out/pmc_step/conv_traffic.json $OUT/ 2>/dev/null; rm -rf gpurun_out/pmc_step/FETCH_SIZE gpurun_out/pmc_step/WRITE_SIZE
timeout 600 bash tools/pmc_conv.sh > $OUT/pmc_halo.txt 2>&1; echo "pmc_conv rc=$?" | tee -a $OUT/rc.log; rm -rf gpurun_out/pmc
tail -3 $OUT/pytest_gpu.log; cut -c1-700 $OUT/bench.json; echo; for f in bench_again bench_bs4 gfwd_bs4_fp32 bench_inpaint bench_rccl1_two_groups bench_rccl1_one_group bench_deterministic bench_gloo2; do echo "$f: $(cut -c1-200 $OUT/$f.json)"; done; cat $OUT/rc.log
