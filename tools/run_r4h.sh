cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_blk.py -x -q -s -k "training" > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log; grep -v "^$" $O/t1.log | tail -6
for rep in 1 2; do for B in 0 1; do
  TL_BLK_TRAIN=$B python bench.py --workload config3 --steps 8 --warmup 3 > $O/c3_$B.json 2> $O/c3_$B.err
  python -c "
import json; d=json.loads(open('$O/c3_$B.json').read().strip().splitlines()[-1]); print('TL_BLK_TRAIN=$B config3 ms_per_step', round(d['ms_per_step'],2), d['config']['last_losses'])"
done; done
export TL_BLK_TRAIN=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3s1 -o p -- python3 bench.py --workload config3 --steps 4 --warmup 2 > $O/c3p.json 2> $O/c3p.err
f=$(find $O/c3s1 -name "*kernel_stats.csv" | head -1); cp $f $O/c3_stats_1.csv; find $O/c3s1 -name "*kernel_trace.csv" -delete
grep "k_conv_blk\|k_conv_direct" $O/c3_stats_1.csv | cut -c1-150
