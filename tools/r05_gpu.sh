#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for i in 1 2; do timeout 600 python bench.py --mode gfwd --dtype fp32 --batch-per-gpu 4 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | cut -c1-140 >> $O/gfwd_fp32.txt; done
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_trainer.py tests/test_gpu_parity.py -m gpu -q -x -k "f32 or fp32 or two_level or configs1 or fullwidth or trainer or generator or race" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -3 $O/pytest.log; cat $O/gfwd_fp32.txt
