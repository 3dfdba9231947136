#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
run() { name=$1; shift; env MG_DP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 "$@" timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-extra --no-roofline 2> $O/$name.err | tail -1 | cut -c1-170 > $O/$name.json; echo "$name: $(cut -c58-170 $O/$name.json)"; }
run rccl1_default
run rccl1_syncbn_async MG_SYNCBN_ASYNC=1
run rccl1_prio_normal MG_WGRAD_STREAM_PRIO=normal
run rccl1_two_groups MG_DP_TWO_GROUPS=1
run rccl1_single_stream MG_WGRAD_STREAM=0
