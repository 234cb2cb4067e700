cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/sq1 -o p -- python3 tools/dev_blk_geom.py 4 > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -o p -- python3 tools/dev_blk_geom.py 4 > $O/sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r4e/sq1", "gpurun_out/r4e/sq2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print("no file", d); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if "k_blk" not in n: continue
        k = n[n.index("k_blk"):][:14]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k in per:
        print(k, {c: round(v / max(cnt[(k, c)], 1)) for c, v in per[k].items()})
PY
