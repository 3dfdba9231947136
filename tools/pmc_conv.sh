#!/bin/bash
# PMC passes over the conv microbenchmark (run on the GPU box): tools/pmc_conv.sh <shape-substring> <dtype>
set -u
SHAPE=${1:-spade_gb_128x2*128@512}; DT=${2:-bf16}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --n 4 --dtype $DT --only "$SHAPE" > $OUT/$name.log 2>&1
  python - <<PY
import csv, collections, glob
f = glob.glob("$OUT/$name/*counter_collection.csv")
if not f: print("$name: no counter file", open("$OUT/$name.log").read()[-600:]); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:70] + " grid=" + r.get("Grid_Size", "?")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "conv_taps" in k or "wgrad" in k or "halo" in k:
        print("$name", k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcw GRBM_GUI_ACTIVE
