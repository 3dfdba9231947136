#!/bin/bash
# One rank, every collective forced (MG_DP_FORCE=1): the step through torch.distributed's collectives vs the C ABI's (MG_COMM=native), with the two traffic
# classes switched off in turn.  GPU box: bash tools/native_comm_ab.sh
run() { name=$1; shift; env "$@" MG_DP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29540 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extra --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-34s %s ms/step' % ('$name', j['ms_per_step']))"; }
run torch A=1
run native MG_COMM=native
run torch_nothing_to_reduce MG_DP_NO_SYNCBN=1 MG_DP_NO_GRAD=1
run native_nothing_to_reduce MG_COMM=native MG_DP_NO_SYNCBN=1 MG_DP_NO_GRAD=1
run torch_only_syncbn MG_DP_NO_GRAD=1
run native_only_syncbn MG_COMM=native MG_DP_NO_GRAD=1
run torch_only_grads MG_DP_NO_SYNCBN=1
run native_only_grads MG_COMM=native MG_DP_NO_SYNCBN=1
run native_grads_in_stream MG_COMM=native MG_DP_GRAD_SIDE=0
run torch_grads_in_stream MG_DP_GRAD_SIDE=0
