#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r05d && export TMPDIR=/tmp
O=gpurun_out/r05d
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_trainer.py -m gpu -q -x -k "side_stream" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
MG_WGRAD_STREAM_PRIO=low timeout 900 python tools/ab_wgrad_stream.py 8 > $O/ab_cases_low_bs8.txt 2>&1
MG_WGRAD_STREAM_PRIO=normal timeout 900 python tools/ab_wgrad_stream.py 8 > $O/ab_cases_normal_bs8.txt 2>&1
( cd /tmp && MG_WGRAD_STREAM=1 timeout 900 rocprofv3 --kernel-trace -d /tmp/kt -o trace --output-format csv -- python $OLDPWD/tools/step_times.py 8 > $OLDPWD/$O/trace_run.log 2>&1 )
python tools/stream_overlap.py /tmp/kt > $O/stream_overlap_wgrad_only.txt 2>&1
tail -3 $O/pytest.log; cat $O/ab_cases_low_bs8.txt $O/ab_cases_normal_bs8.txt | grep -v amdgpu.ids | grep -v "^Network"; cat $O/stream_overlap_wgrad_only.txt
