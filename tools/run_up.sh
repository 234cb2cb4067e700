cd $GRAFT_REPO_ROOT; O=gpurun_out/up; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_blk.py tests/test_gpu_parity.py -x -q -k "inverse_conv_scatter or kernel_families or hdbscan_auto" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -3 $O/t.log
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
for rep in 1 2 3; do for U in 0 1; do
  TL_TUNING="up=$U" python bench.py $Q > $O/b${U}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/b${U}_$rep.json").read().strip().splitlines()[-1])
print("up=$U rep $rep: in-flight %.3f ms  one-tile %.3f  latency-median %.3f  conv_ms %.3f frac %.4f" % (d["ms_per_step"], d["one_tile_at_a_time"]["ms_per_step"], d["latency_ms_median"], d["roofline"]["conv_ms_per_step"], d["roofline"]["frac"]))
PY
done; done
