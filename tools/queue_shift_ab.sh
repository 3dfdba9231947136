#!/bin/bash
# How sensitive is the one-rank forced-collectives step to WHICH hardware queue torch's process-group stream / RCCL's streams land on?  K dummy streams created
# before anything else shift the map.  Arms: gradient buckets on the process group's stream (MG_DP_GRAD_SIDE=2, the default so far), in the issuing stream
# (MG_DP_GRAD_SIDE=0), and through the C ABI's communicator (MG_COMM=native, its stream probed).   GPU box: bash tools/queue_shift_ab.sh
run() { name=$1; shift; env "$@" MG_DP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline --no-traffic --no-extra --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-30s %7.2f ms/step  probe %s' % ('$name', j['ms_per_step'], [p['candidates_tried'] for p in j.get('second_stream_probe', [])]))"; }
for K in 0 1 2 3 4 5 6 7; do
  run "K=$K pg_stream" MG_BENCH_DUMMY_STREAMS=$K MG_DP_GRAD_SIDE=2
  run "K=$K in_stream" MG_BENCH_DUMMY_STREAMS=$K MG_DP_GRAD_SIDE=0
  run "K=$K native" MG_BENCH_DUMMY_STREAMS=$K MG_COMM=native
done
