"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: counters summed over dispatches, MFMA-pipe utilisation, wave-cycle split.
    python tools/pmc_kernels.py DIR [name substring ...]"""
import collections, csv, glob, os, re, sys
src = sys.argv[1]; subs = sys.argv[2:]
files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(files[0])):
    n = r["Kernel_Name"]
    if subs and not any(s in n for s in subs):
        continue
    k = re.sub(r"\(anonymous namespace\)::|void |\(ConvP.*|\(unsigned short const.*|\(void const.*|\(float const.*", "", n)[:70]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        per[k]["n"] += 1; per[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, c in sorted(per.items(), key=lambda kv: -kv[1]["ns"]):
    wc = max(c["SQ_WAVE_CYCLES"], 1.0); cyc = max(c["GRBM_GUI_ACTIVE"] / 8, 1.0)
    print(f"{k:72s} n={int(c['n']):3d} avg {c['ns'] / max(c['n'], 1) / 1e3:8.1f} us  clk {cyc / max(c['ns'], 1):.2f} GHz  mfma {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):5.1f} %  "
          f"issuing {100 * c['SQ_ACTIVE_INST_ANY'] / wc:5.1f} %  parked {100 * c['SQ_WAIT_ANY'] / wc:5.1f} %  stalled {100 * c['SQ_WAIT_INST_ANY'] / wc:5.1f} %  "
          f"lds-conflict cycles / busy {100 * c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_BUSY_CYCLES'], 1):6.2f} %  " + " ".join(f"{a}={v:.3g}" for a, v in c.items() if a.startswith("SQ_INST") or a.startswith("SQ_ACTIVE_INST_") or a.startswith("SQ_LDS")))
