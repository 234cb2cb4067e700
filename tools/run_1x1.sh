cd $GRAFT_REPO_ROOT; O=gpurun_out/x11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "kernel_families or real_rulebooks or conv_shapes" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -3 $O/t.log
python tools/dev_conv_table.py 2>/dev/null | grep "K= 1\|launches"
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
for rep in 1 2; do
  python bench.py $Q > $O/b_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/b_$rep.json").read().strip().splitlines()[-1])
print("rep $rep: in-flight %.3f ms  one-tile %.3f  latency-median %.3f  conv_ms %.3f frac %.4f" % (d["ms_per_step"], d["one_tile_at_a_time"]["ms_per_step"], d["latency_ms_median"], d["roofline"]["conv_ms_per_step"], d["roofline"]["frac"]))
PY
done
