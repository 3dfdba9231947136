#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
MG_BRANCH_STREAMS=2 timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_fullsize.py -m gpu -q -x -k "trainer or benchmark_config" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python tools/ab_intflag.py BRANCH_STREAMS 0 1 2 > $O/ab_branch.txt 2>&1
tail -3 $O/pytest.log; grep -v "amdgpu.ids\|^Network" $O/ab_branch.txt
