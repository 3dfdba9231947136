#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for i in 1 2; do timeout 600 python bench.py --mode gfwd --dtype fp32 --batch-per-gpu 4 --no-cpu-baseline --no-traffic 2>/dev/null | tail -1 | cut -c1-140 >> $O/gfwd_fp32.txt; done
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "two_level or side_stream" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
tail -3 $O/pytest.log; cat $O/gfwd_fp32.txt
