cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_blk.py -x -q -k "geometry" > $O/test_blk.log 2>&1; echo "pytest rc=$?" >> $O/test_blk.log; tail -5 $O/test_blk.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 tools/dev_blk_geom.py 10 > $O/geom.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; find $O/stats -name "*kernel_trace.csv" -delete
grep -i "blk_" $O/kernel_stats.csv | cut -c1-150
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq1 -o p -- python3 tools/dev_blk_geom.py 4 > $O/sq1.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r4f/sq1/**/*counter_collection.csv", recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if "k_blk_units" not in n: continue
    per["u"][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print({c: round(v / cnt[c]) for c, v in per["u"].items()})
PY
