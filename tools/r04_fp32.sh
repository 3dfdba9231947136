#!/bin/bash
# One GPU-box call:  tools/r04_fp32.sh <tag>  -- what the fp32 (parity-configuration) kernels changed: parity suites, the gradient / noise probes,
# the default bench line (its `extra` leg is configs[1], fp32).
set -u
TAG=${1:-r04p}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/rc.log
timeout 500 python tools/grad_probe.py > $OUT/grad_probe.txt 2>&1; echo "grad_probe rc=$?" | tee -a $OUT/rc.log
timeout 400 python tools/noise_probe.py > $OUT/noise_probe.txt 2>&1; echo "noise_probe rc=$?" | tee -a $OUT/rc.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/rc.log
timeout 300 python bench.py --mode gfwd --dtype fp32 --batch-per-gpu 4 --no-cpu-baseline > $OUT/gfwd_fp32.json 2> $OUT/gfwd_fp32.err; echo "gfwd rc=$?" | tee -a $OUT/rc.log
tail -6 $OUT/pytest_gpu.log | cut -c1-400; grep -n "^==\|max-abs: HIP\|rel-L2 : HIP" $OUT/grad_probe.txt | head -8; sed -n 4,22p $OUT/grad_probe.txt | cut -c1-150
grep -v "amdgpu\|^Network" $OUT/noise_probe.txt | tail -8 | cut -c1-220; python - <<PY
import json
j=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1]); print(j["value"], j["ms_per_step"], j["roofline"]["frac"], json.dumps(j.get("extra"))[:400])
j=json.loads([l for l in open("$OUT/gfwd_fp32.json") if l.startswith("{")][-1]); print("gfwd", j["value"], j["ms_per_step"], j["roofline"].get("frac"))
PY
cat $OUT/rc.log
