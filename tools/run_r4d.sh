cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_blk.py -x -q > $O/test_blk.log 2>&1; echo "pytest rc=$?" >> $O/test_blk.log; tail -5 $O/test_blk.log
timeout 400 python tools/dev_blk.py > $O/dev_blk.log 2>&1; tail -12 $O/dev_blk.log
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 bench.py --steps 6 --warmup 2 --tiles-in-flight 1 $Q > $O/bench_1.json 2> $O/stats.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; find $O/stats -name "*kernel_trace.csv" -delete
grep -i "blk_\|k_scan_u32" $O/kernel_stats.csv | cut -c1-160
