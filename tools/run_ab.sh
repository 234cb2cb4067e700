# same-box A/B of the forward: TL_BLK=0 | TL_BLK=1 (unit builder on the main stream) | TL_BLK=1 + side stream; 3 tiles in flight and 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab}; mkdir -p $O
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
for rep in 1 2; do
for cfg in "0 1" "1 0" "1 1"; do
  set -- $cfg
  TL_BLK=$1 TL_BLK_SIDE=$2 python bench.py $Q > $O/b_$1$2_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/b_$1$2_$rep.json").read().strip().splitlines()[-1])
print("TL_BLK=$1 SIDE=$2 rep $rep: 3-in-flight %.3f ms  one-tile %.3f  latency-median %.3f  conv_ms %.3f frac %.4f" % (d["ms_per_step"], d["one_tile_at_a_time"]["ms_per_step"], d["latency_ms_median"], d["roofline"]["conv_ms_per_step"], d["roofline"]["frac"]))
PY
done; done
