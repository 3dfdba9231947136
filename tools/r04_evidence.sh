#!/bin/bash
# Round-4 evidence in one GPU-box call (through gpurun from the repo root):  bash tools/r04_evidence.sh
#   full GPU suite + smoke + the default bench line (tools/r04_check.sh), bs 4, the in-painting branch, one-rank RCCL with every collective
#   forced (one communicator = default / MG_DP_TWO_GROUPS=1), MG_DETERMINISTIC=1, 2-rank gloo self-spawn smoke, conv census, MFMA counters of
#   the halo conv, the trainer noise probe, the statistics-free upper bound, the fp32 gradient probe, the MFMA + LDS-feed ceiling probe.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r04}; mkdir -p $OUT; cd $R
bash tools/r04_check.sh ${1:-r04} > $OUT/check.log 2>&1; tail -12 $OUT/check.log | cut -c1-400
timeout 300 python bench.py --batch-per-gpu 4 --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_bs4.json 2> $OUT/bench_bs4.err; echo "bench_bs4 rc=$?" | tee -a $OUT/rc.log
timeout 300 python bench.py --inpaint-orient --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_inpaint.json 2> $OUT/bench_inpaint.err; echo "inpaint rc=$?" | tee -a $OUT/rc.log
timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_again.json 2> $OUT/bench_again.err; echo "bench_again rc=$?" | tee -a $OUT/rc.log
MG_DP_FORCE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29741 bench.py --gpus 1 --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_rccl1_one_group.json 2> $OUT/bench_rccl1_one_group.err; echo "rccl1 one group rc=$?" | tee -a $OUT/rc.log
MG_DP_FORCE=1 MG_DP_TWO_GROUPS=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29742 bench.py --gpus 1 --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_rccl1_two_groups.json 2> $OUT/bench_rccl1_two_groups.err; echo "rccl1 two groups rc=$?" | tee -a $OUT/rc.log
MG_DETERMINISTIC=1 timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-extra > $OUT/bench_deterministic.json 2> $OUT/bench_deterministic.err; echo "deterministic rc=$?" | tee -a $OUT/rc.log
MG_BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 2 --warmup 1 --batch-per-gpu 2 --no-cpu-baseline > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "bench_gloo2 rc=$?" | tee -a $OUT/rc.log
timeout 300 python tools/conv_census.py > $OUT/conv_census.txt 2> $OUT/conv_census.err; echo "census rc=$?" | tee -a $OUT/rc.log
timeout 300 python tools/ab_stats_free.py > $OUT/ab_stats_free.txt 2>&1; echo "ab_stats_free rc=$?" | tee -a $OUT/rc.log
timeout 400 python tools/noise_probe.py > $OUT/noise_probe.txt 2>&1; echo "noise_probe rc=$?" | tee -a $OUT/rc.log
timeout 500 python tools/grad_probe.py > $OUT/grad_probe.txt 2>&1; echo "grad_probe rc=$?" | tee -a $OUT/rc.log
(hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_ceiling tools/mfma_ceiling.hip 2>/dev/null && timeout 120 /tmp/mfma_ceiling) > $OUT/mfma_ceiling.txt 2>&1; echo "mfma_ceiling rc=$?" | tee -a $OUT/rc.log
timeout 600 bash tools/pmc_conv.sh > $OUT/pmc_halo.txt 2>&1; echo "pmc_conv rc=$?" | tee -a $OUT/rc.log; rm -rf gpurun_out/pmc
for f in bench_again bench_bs4 bench_inpaint bench_rccl1_one_group bench_rccl1_two_groups bench_deterministic bench_gloo2; do echo "$f: $(cut -c1-170 $OUT/$f.json)"; done
grep -v "^Network\|amdgpu" $OUT/ab_stats_free.txt | tail -9; tail -12 $OUT/noise_probe.txt | cut -c1-250; cat $OUT/rc.log
