#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python tools/ab_pyflag.py WGRAD_HALF_CU > $O/ab_half_cu_bs8.txt 2>&1
timeout 600 python tools/ab_wgrad_stream.py 8 > $O/ab_wgrad_bs8.txt 2>&1
timeout 600 python tools/ab_wgrad_stream.py 4 > $O/ab_wgrad_bs4.txt 2>&1
MG_DP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-extra --no-roofline 2>/dev/null | tail -1 | cut -c58-170 > $O/rccl1.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_trainer.py -m gpu -q -x -k "wgrad or side_stream or trainer" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -3 $O/pytest.log; grep -hv "amdgpu.ids\|^Network" $O/ab_half_cu_bs8.txt $O/ab_wgrad_bs8.txt $O/ab_wgrad_bs4.txt; cat $O/rccl1.txt
