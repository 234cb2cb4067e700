# A/B of the config-3 training step under TL_TUNING / environment settings, alternating to average out box drift:
#   bash tools/ab_train.sh "wgrad_dma=1" "wgrad_dma=2" ...      (on the GPU box, through gpurun)
for rep in 1 2; do
  for t in "$@"; do
    printf "%s  " "$t"
    TL_TUNING="$t" timeout 200 python bench.py --workload config3 --steps 10 --warmup 3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), d["config"]["last_losses"])'
  done
done
