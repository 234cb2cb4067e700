# tiles in flight x hardware queues (same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab5; mkdir -p $O
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
for rep in 1 2; do
for cfg in "8 2" "8 3" "8 4" "8 5" "8 6" "4 3"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 python bench.py $Q --tiles-in-flight $2 > $O/b_$1_$2_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/b_$1_$2_$rep.json").read().strip().splitlines()[-1])
print("GPU_MAX_HW_QUEUES=$1 tiles in flight $2 rep $rep: %.3f ms per tile  one-tile %.3f  latency-median %.3f" % (d["ms_per_step"], d["one_tile_at_a_time"]["ms_per_step"], d["latency_ms_median"]))
PY
done; done
