#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 600 python tools/ab_pyflag.py BG_SIDE_STREAM > $O/ab_bg_bs8.txt 2>&1
tail -3 $O/pytest.log; grep -v "amdgpu.ids\|^Network" $O/ab_bg_bs8.txt
