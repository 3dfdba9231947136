#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_trainer.py -m gpu -q -x -k "side_stream or benchmark_config or fullwidth or trainer" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python tools/ab_wgrad_stream.py 8 > $O/ab_wgrad_bs8.txt 2>&1
tail -5 $O/pytest.log; grep -v "amdgpu.ids\|^Network" $O/ab_wgrad_bs8.txt
