#!/bin/bash
# round 5, first GPU call: the two parity holes of VERDICT r4 (reference trainer over dropin.install() on the real kernels; configs[1] as stated;
# tightened bf16 bands) + the three-resident halo kernel: bit identity, per-launch A/B, step A/B.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r05a && export TMPDIR=/tmp
O=gpurun_out/r05a
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests/test_dropin.py tests/test_gpu_kernels.py -m gpu -q -x -k "hip or halo3 or halo or wide_epilogue or race" -rs > $O/pytest_dropin_halo3.log 2>&1; echo "rc $?" >> $O/pytest_dropin_halo3.log
timeout 900 python tools/ab_halo3.py > $O/ab_halo3.txt 2>&1
timeout 600 python tools/ab_option.py 20 0 1 > $O/ab_step_halo3.txt 2>&1
timeout 600 python tools/ab_option.py 20 0 2 > $O/ab_step_halo3_all.txt 2>&1
timeout 1800 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -s -k "configs1 or bf16" > $O/pytest_fullsize.log 2>&1; echo "rc $?" >> $O/pytest_fullsize.log
tail -5 $O/pytest_dropin_halo3.log; cat $O/ab_halo3.txt; cat $O/ab_step_halo3.txt $O/ab_step_halo3_all.txt; tail -8 $O/pytest_fullsize.log
