cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_blk.py -x -q > $O/test_blk.log 2>&1; echo "pytest rc=$?" >> $O/test_blk.log; tail -15 $O/test_blk.log
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
for B in 0 1; do
  export TL_BLK=$B
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_blk$B -o p -- python3 bench.py --steps 6 --warmup 2 --tiles-in-flight 1 $Q > $O/bench_blk$B.json 2> $O/stats_blk$B.err
  f=$(find $O/stats_blk$B -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_blk$B.csv
  find $O/stats_blk$B -name "*kernel_trace.csv" -delete
done
unset TL_BLK
head -40 $O/kernel_stats_blk1.csv | cut -c1-220
