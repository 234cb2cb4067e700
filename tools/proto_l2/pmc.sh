# effective shader clock and MFMA-busy share of the gather kernel and the two prototypes (rocprofv3 PMC pass over run_l2.py)
export GPU_MAX_HW_QUEUES=8   # rocprofv3 starts the HIP runtime before python runs: the package's own default would come too late
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/proto/pmc; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O -o p -- python3 tools/proto_l2/run_l2.py > $O/out.txt 2> $O/err.txt
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/proto/pmc/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_conv_l2" not in n and "k_conv_streamq<27, 2, 1" not in n: continue
    key = n.split("(")[0][-40:]
    per[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        per[key]["n"] += 1; per[key]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, c in per.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print("%-42s dispatches %4d  avg %.1f us  effective clock %.2f GHz  MFMA-busy %.1f %%" % (k, c["n"], c["ns"] / c["n"] / 1e3, cyc / c["ns"], 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)))
PY
