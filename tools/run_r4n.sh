cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4n; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -6 $O/gpu_tests.log
bash tools/run_r4m.sh 2>&1 | grep "TL_BLK_PRO\|k_conv_blk\|k_conv_direct\|k_conv_small"
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
for rep in 1 2; do for P in 0 1; do
  TL_BLK_PRO=$P python bench.py $Q > $O/b${P}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/b${P}_$rep.json").read().strip().splitlines()[-1])
print("TL_BLK_PRO=$P rep $rep: in-flight %.3f ms  one-tile %.3f  latency-median %.3f  conv_ms %.3f frac %.4f" % (d["ms_per_step"], d["one_tile_at_a_time"]["ms_per_step"], d["latency_ms_median"], d["roofline"]["conv_ms_per_step"], d["roofline"]["frac"]))
PY
done; done
