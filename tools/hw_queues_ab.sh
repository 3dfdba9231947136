#!/bin/bash
# Which hardware queue the second stream lands on decides whether it overlaps the compute stream at all (round 6): the step under the default
# GPU_MAX_HW_QUEUES and under 8, with torch's RCCL communicator, with the C ABI's, and single-process; MG_STREAM_PROBE=0 = the first stream created (rounds 5),
# 1 = candidates are probed for real overlap (default).  GPU box: bash tools/hw_queues_ab.sh
run() { name=$1; shift; env "$@" MG_DP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29540 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extra --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-40s %s ms/step  probe %s' % ('$name', j['ms_per_step'], j.get('second_stream_probe')))"; }
plain() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extra --no-roofline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-40s %s ms/step  probe %s' % ('$name', j['ms_per_step'], j.get('second_stream_probe')))"; }
for P in 0 1; do
plain single_process_probe$P MG_STREAM_PROBE=$P
plain single_process_hwq8_probe$P MG_STREAM_PROBE=$P GPU_MAX_HW_QUEUES=8
run torch_probe$P MG_STREAM_PROBE=$P
run torch_hwq8_probe$P MG_STREAM_PROBE=$P GPU_MAX_HW_QUEUES=8
run native_probe$P MG_STREAM_PROBE=$P MG_COMM=native
run native_hwq8_probe$P MG_STREAM_PROBE=$P MG_COMM=native GPU_MAX_HW_QUEUES=8
done
