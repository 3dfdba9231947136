#!/bin/bash
# One GPU-box call (through gpurun from the repo root):  tools/r06_check.sh <tag>
#   full GPU parity suite (incl. tests/test_dropin.py[hip]: the reference's own trainer from the staged archive), smoke(), the DEFAULT bench
#   line as the driver runs it (in-run rocprofv3 --pmc traffic passes, fp32 configs[1] leg, CPU baseline from the staged reference), rocprofv3
#   kernel stats of the bench command with the weight gradients on the side stream and in-stream, the bs 4 line, the conv census.
set -u
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 2400 python -m pytest tests -q -m gpu -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" | tee -a $OUT/rc.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/rc.log
T0=$(date +%s); timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 )) s" | tee -a $OUT/rc.log
timeout 600 python bench.py --batch-per-gpu 4 --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_bs4.json 2> $OUT/bench_bs4.err; echo "bench bs4 rc=$?" | tee -a $OUT/rc.log
timeout 600 python bench.py --batch-per-gpu 1 --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_bs1.json 2> $OUT/bench_bs1.err; echo "bench bs1 rc=$?" | tee -a $OUT/rc.log
MG_WGRAD_STREAM=0 timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_single_stream.json 2> $OUT/bench_single_stream.err; echo "bench single-stream rc=$?" | tee -a $OUT/rc.log
MG_DP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_rccl1.json 2> $OUT/bench_rccl1.err; echo "bench rccl1 (one rank, collectives forced) rc=$?" | tee -a $OUT/rc.log
MG_DP_FORCE=1 MG_COMM=native HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_native1.json 2> $OUT/bench_native1.err; echo "bench native comm (one rank, collectives forced through mg_allreduce_*) rc=$?" | tee -a $OUT/rc.log
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-extra > $OUT/prof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/rc.log
F=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats.csv; rm -rf $OUT/prof
MG_WGRAD_STREAM=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-extra > $OUT/prof1.log 2>&1; echo "rocprof single-stream rc=$?" | tee -a $OUT/rc.log
F=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats_single_stream.csv; rm -rf $OUT/prof
cd $R
timeout 600 python tools/conv_census.py > $OUT/conv_census.txt 2>&1; echo "census rc=$?" | tee -a $OUT/rc.log
tail -6 $OUT/pytest_gpu.log | cut -c1-600; tail -1 $OUT/smoke.log; cut -c1-3500 $OUT/bench.json; cut -c1-400 $OUT/bench_bs4.json; cut -c1-400 $OUT/bench_single_stream.json; cut -c1-400 $OUT/bench_rccl1.json; cut -c1-400 $OUT/bench_native1.json; cut -c1-400 $OUT/bench_bs1.json; cat $OUT/rc.log
