cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j; mkdir -p $O
export TL_BLK_SIDE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 tools/dev_blk_geom.py 10 > $O/geom.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; find $O/stats -name "*kernel_trace.csv" -delete
grep -i "blk_" $O/kernel_stats.csv | cut -c1-130
