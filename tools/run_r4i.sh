cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_blk.py -x -q > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log; tail -4 $O/t1.log
timeout 2400 python -m pytest tests/ -q -m gpu -k "kernel_families_agree or side_stream or bf16_vs_float64" > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log; tail -6 $O/t2.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 tools/dev_blk_geom.py 10 > $O/geom.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; find $O/stats -name "*kernel_trace.csv" -delete
grep -i "blk_" $O/kernel_stats.csv | cut -c1-130
