# latency / throughput / training with an idle host and with every granted core kept busy by other processes
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe"
run() {
  python bench.py --steps 30 --warmup 6 $Q 2>/dev/null > /tmp/o.json
  python -c "
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); l=d['latency']
print('$1', 'ms/tile', round(d['ms_per_step'],3), '| one at a time', round(d['one_tile_at_a_time']['ms_per_step'],3), '| latency median', round(l['median_ms'],3), 'min', round(l['min_ms'],3), 'max', round(l['max_ms'],3), 'host enqueue', round(l['host_enqueue_ms_median'],3), '| training', round(d['training_step']['ms_per_step'],2), '| config4 ms/tile', round(d['config4']['ms_per_tile'],3))"
}
python -c "import os; print('cores granted', len(os.sched_getaffinity(0)))"
run idle
NB=${NB:-16}
pids=""
for i in $(seq 1 $NB); do python -c "
while True:
    s = 0
    for i in range(10000000): s += i
" & pids="$pids $!"; done
sleep 2
run busy$NB
kill $pids
wait 2>/dev/null
run idle
