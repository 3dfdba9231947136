#!/bin/bash
# round 5, second GPU call: side-stream weight gradients (trainer parity + A/B at two priorities and two batch sizes), the re-ordered
# determinism test, the CPU baseline from the staged reference.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r05b && export TMPDIR=/tmp
O=gpurun_out/r05b
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_fullsize.py -m gpu -q -x -k "side_stream or bitwise_deterministic" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
MG_WGRAD_STREAM_PRIO=low timeout 600 python tools/ab_wgrad_stream.py 8 > $O/ab_wgrad_low_bs8.txt 2>&1
MG_WGRAD_STREAM_PRIO=normal timeout 600 python tools/ab_wgrad_stream.py 8 > $O/ab_wgrad_normal_bs8.txt 2>&1
MG_WGRAD_STREAM_PRIO=low timeout 600 python tools/ab_wgrad_stream.py 4 > $O/ab_wgrad_low_bs4.txt 2>&1
tail -4 $O/pytest.log; cat $O/ab_wgrad_low_bs8.txt $O/ab_wgrad_normal_bs8.txt $O/ab_wgrad_low_bs4.txt | grep -v amdgpu.ids | grep -v "^Network"
