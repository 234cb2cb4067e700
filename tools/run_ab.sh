# same-box A/B of the forward: TL_BLK=0 | TL_BLK=1 with the unit builder's side stream forced off / on / automatic; 3 tiles in flight and 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ab}; mkdir -p $O
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
for rep in 1 2; do
for cfg in "0 0" "1 0" "1 1" "1 auto"; do
  set -- $cfg
  if [ "$2" = "auto" ]; then unset TL_BLK_SIDE; else export TL_BLK_SIDE=$2; fi
  TL_BLK=$1 python bench.py $Q > $O/b_$1$2_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/b_$1$2_$rep.json").read().strip().splitlines()[-1])
print("TL_BLK=$1 SIDE=$2 rep $rep: 3-in-flight %.3f ms  one-tile %.3f  latency-median %.3f  conv_ms %.3f frac %.4f" % (d["ms_per_step"], d["one_tile_at_a_time"]["ms_per_step"], d["latency_ms_median"], d["roofline"]["conv_ms_per_step"], d["roofline"]["frac"]))
PY
done; done
