#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python tools/ab_intflag.py WGRAD_HALF_CU 0 1 2 > $O/ab_half_cu.txt 2>&1
grep -hv "amdgpu.ids\|^Network" $O/ab_half_cu.txt
