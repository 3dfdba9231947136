#!/bin/bash
# The first multi-GPU lease in ONE call: for N in {2, 4, 8} (as many GPUs as the node has) run bench.py at the
# 8-GPU target's per-GPU batch (4: BASELINE configs[3], bs 32 on 8 GPUs) in the default form and in each A/B form of the data-parallel
# design points that only hardware can settle, and print one table:
#   images/s, ms/step, sync-BN all-reduce ms/step (rank 0's compute stream), spread between the ranks, collectives per step.
#     bash tools/scale_first_run.sh [steps] [batch-per-gpu]        (an 8-GPU MI355X node; ~1 min per line)
cd "$(dirname "$0")/.." || exit 1
STEPS=${1:-10}; BPG=${2:-4}; OUT=${OUT:-gpurun_out/scale_first_run}; mkdir -p "$OUT"
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
export HSA_ENABLE_IPC_MODE_LEGACY=0
line() {   # name, N, env...
    local name=$1 n=$2; shift 2
    local log="$OUT/${name}_n${n}.json"
    if [ "$n" = 1 ]; then env "$@" python bench.py --gpus 1 --steps "$STEPS" --warmup 3 --batch-per-gpu "$BPG" --no-cpu-baseline --no-traffic --no-extra 2>"$log.err" | tail -1 > "$log"
    else env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) \
             bench.py --gpus "$n" --steps "$STEPS" --warmup 3 --batch-per-gpu "$BPG" --no-cpu-baseline --no-traffic --no-extra 2>"$log.err" | tail -1 > "$log"; fi
    python - "$name" "$n" "$log" <<'PY'
import json, sys
name, n, path = sys.argv[1:4]
try:
    j = json.loads(open(path).read())
    sb = j.get("syncbn_allreduce_ms_per_step") or {}
    print(f"{name:28s} N={n}  {j['value']:8.1f} img/s  {j['ms_per_step']:7.2f} ms/step  sync-BN {sb.get('ms', float('nan')) if isinstance(sb, dict) else sb!s:>6} ms"
          f"  rank spread {j.get('rank_spread_ms', float('nan')):6.2f} ms  collectives {j.get('collectives_per_step', {})}")
except Exception as e:
    print(f"{name:28s} N={n}  FAILED ({e}); see {path}.err")
PY
}
echo "# bench.py --batch-per-gpu $BPG --steps $STEPS on $NGPU GPU(s); default = one communicator, in-stream sync-BN, grad buckets on the process group's stream, weight gradients + the generator step's D branch / VGG(real) on the second stream"
line default 1
for N in 2 4 8; do
    [ "$N" -le "$NGPU" ] || continue
    line default           $N
    line syncbn_async      $N MG_SYNCBN_ASYNC=1
    line two_groups        $N MG_DP_TWO_GROUPS=1
    line grads_in_stream   $N MG_DP_GRAD_SIDE=0
    line single_stream     $N MG_WGRAD_STREAM=0
    line no_branch_streams $N MG_BRANCH_STREAMS=0
    line native_comm       $N MG_COMM=native
done
echo "# scaling efficiency = value(N) / (N x value(1)); the >= 6.5x target of BASELINE.json is value(8) / value(1) at this per-GPU batch"
