cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_blk.py -x -q > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log; tail -5 $O/t1.log
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
for rep in 1 2; do for P in 0 1; do
  TL_BLK_PRO=$P python bench.py $Q > $O/b_$P_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/b_$P_$rep.json").read().strip().splitlines()[-1])
print("TL_BLK_PRO=$P rep $rep: in-flight %.3f ms  one-tile %.3f  latency-median %.3f  conv_ms %.3f frac %.4f" % (d["ms_per_step"], d["one_tile_at_a_time"]["ms_per_step"], d["latency_ms_median"], d["roofline"]["conv_ms_per_step"], d["roofline"]["frac"]))
PY
done; done
