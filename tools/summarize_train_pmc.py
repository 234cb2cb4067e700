"""SQ counters of the training step (bench.py --workload config3 under rocprofv3 --pmc, kernels serialised by the profiler) per kernel family:
    python tools/summarize_train_pmc.py <dir with sq_c3/ and lds_c3/ counter_collection.csv> <out.csv>
MFMA-pipe utilisation, shader clock and the wave-cycle split as tools/summarize_profile.py computes them for the forward, plus -- from the
second pass -- how much of a wave's time is spent with an LDS instruction in flight and the bank-conflict share of the LDS's active cycles."""
import collections, csv, glob, os, re, sys

src, out = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def load(sub):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            did = r["Dispatch_Id"]
            if did not in seen:
                seen.add(did)
                per[k]["dispatches"] += 1
                per[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per


sq, lds = load("sq_c3"), load("lds_c3")
rows = []
for k, c in sq.items():
    if c["ns"] < 1e5:                                       # below 0.1 ms in total: noise
        continue
    cyc, wc = max(c["GRBM_GUI_ACTIVE"] / 8, 1.0), max(c["SQ_WAVE_CYCLES"], 1.0)
    l = lds.get(k, {})
    lwc = max(l.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    rows.append((c["ns"], k, int(c["dispatches"]), c["ns"] / c["dispatches"] / 1e3, cyc / max(c["ns"], 1.0), 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                 100 * c["SQ_ACTIVE_INST_ANY"] / wc, 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc,
                 100 * l.get("SQ_ACTIVE_INST_LDS", 0.0) / lwc, 100 * l.get("SQ_WAIT_INST_LDS", 0.0) / lwc, 100 * l.get("SQ_ACTIVE_INST_VMEM", 0.0) / lwc, 100 * l.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(l.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0),
                 l.get("SQ_INSTS_LDS", 0.0) / max(l.get("dispatches", 0.0), 1.0)))
with open(out, "w") as f:
    f.write("kernel,dispatches,avg_us,sclk_ghz,mfma_util_pct,wave_cycles_issuing_pct,wave_cycles_parked_pct,wave_cycles_issue_stall_pct,"
            "wave_cycles_lds_inst_pct,wave_cycles_lds_issue_stall_pct,wave_cycles_vmem_inst_pct,lds_bank_conflict_pct_of_lds_active,lds_insts_per_dispatch\n")
    for r in sorted(rows, reverse=True)[:40]:
        f.write('"%s",%d,%.1f,%.2f,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f,%.0f\n' % r[1:])
print(open(out).read())
