"""Turn the rocprofv3 outputs of a bench.py run into the files kept under profiles/<tag>/.

    python tools/summarize_profile.py <gpurun_out/dir> <profiles/tag>

Expects in <dir>: stats/*kernel_stats.csv (--kernel-trace --stats), fetch/*counter_collection.csv (--pmc FETCH_SIZE),
write/*counter_collection.csv (--pmc WRITE_SIZE), optionally sq/*counter_collection.csv (--pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES
SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE), bench_line.json.
PMC units are KB; gfx950 FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, shutil, sys, collections

src, dst = sys.argv[1], sys.argv[2]
forwards = {}                                    # per PMC pass: forwards profiled = k_head dispatches seen
os.makedirs(dst, exist_ok=True)
ks = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(dst, "bf16_kernel_stats.csv"))


def pmc(sub, counter, out_name):
    files = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return None
    per = collections.defaultdict(lambda: [0, 0.0])
    total = 0.0
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if "k_head" in name:
            forwards[sub] = forwards.get(sub, 0) + 1
        if "k_conv" not in name:
            continue
        name = '"' + name[name.index("k_conv"):].split("(ConvP")[0] + '"'
        per[name][0] += 1; per[name][1] += float(r["Counter_Value"]); total += float(r["Counter_Value"])
    with open(os.path.join(dst, out_name), "w") as f:
        f.write("kernel,dispatches,%s_kb_total\n" % counter.lower())
        for k, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k},{n},{v:.1f}\n")
    return total


def sq():
    """Per conv kernel: MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (elapsed cycles x 1024 SIMDs), elapsed cycles =
    GRBM_GUI_ACTIVE / 8 (the counter comes back summed over the 8 XCDs: 4.67 M for a 301 us dispatch = 1.94 GHz each;
    MFMA_BUSY counts cycles summed over SIMDs = 32 x the number of 32x32x16 MFMAs, checked on the 64->64 conv:
    1.105 M rows / 32 x 27 taps x 8 = 7.46 M MFMAs x 32 = 238.7 M, the exact counter value), the effective shader clock
    under the profiler, and how the wave-cycles split into issuing / parked (s_waitcnt, barrier) / issue-stalled."""
    files = glob.glob(os.path.join(src, "sq", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"]
        if "k_conv" not in name and "k_head" not in name:
            continue
        key = "k_head" if "k_head" in name else name[name.index("k_conv"):].split("(ConvP")[0]
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            per[key]["dispatches"] += 1
            per[key]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    with open(os.path.join(dst, "pmc_sq.csv"), "w") as f:
        # the clock estimate GRBM_GUI_ACTIVE / 8 / duration is unreliable for dispatches below ~40 us (it returned 3.5-4.7 GHz for the
        # small-level kernels in round 1): left blank there
        f.write("kernel,dispatches,avg_us,sclk_ghz,mfma_util_pct,wave_cycles_issuing_pct,wave_cycles_parked_pct,wave_cycles_issue_stall_pct\n")
        for k, c in sorted(per.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
            cyc, wc = max(c["GRBM_GUI_ACTIVE"] / 8, 1.0), max(c["SQ_WAVE_CYCLES"], 1.0)
            avg_us = c["ns"] / c["dispatches"] / 1e3
            f.write('"%s",%d,%.1f,%s,%.1f,%.1f,%.1f,%.1f\n' % (k, c["dispatches"], avg_us, ("%.2f" % (cyc / max(c["ns"], 1.0))) if avg_us >= 40 else "",
                    100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                    100 * c["SQ_ACTIVE_INST_ANY"] / wc, 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc))


sq()
fetch = pmc("fetch", "FETCH_SIZE", "pmc_fetch_size.csv")
write = pmc("write", "WRITE_SIZE", "pmc_write_size.csv")
for name in ("bench_line.json", "bench_line_1_in_flight.json", "bench_unprofiled.json", "bench_unprofiled_1_in_flight.json", "power_per_layer.txt",
             "train_per_layer.txt", "bench_config3.json", "bench_config5.json", "bench_config4.json", "bench_config3_profiled.json", "config3_timeline.txt",
             "config3_phases.txt", "wgrad_dense_vs_pair_list.txt", "conv_launch_table.txt"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
ks3 = glob.glob(os.path.join(src, "c3", "**", "*kernel_stats.csv"), recursive=True)
if ks3:
    shutil.copy(ks3[0], os.path.join(dst, "config3_kernel_stats.csv"))
ks1 = glob.glob(os.path.join(src, "stats1", "**", "*kernel_stats.csv"), recursive=True)
if ks1:
    shutil.copy(ks1[0], os.path.join(dst, "bf16_kernel_stats_1_in_flight.csv"))
if fetch is not None and write is not None:
    t = {"dtype": "bf16", "workload": "config2", "forwards_profiled": forwards,
         "fetch_size_kb_per_step": fetch / forwards["fetch"], "write_size_kb_per_step": write / forwards["write"],
         "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> doubled (MI355X_MICROARCH.md HBM section); units KB",
         "hbm_gb_per_step": (2 * fetch / forwards["fetch"] + write / forwards["write"]) * 1024 / 1e9,
         "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --output-format csv -- python3 bench.py --steps 3 --warmup 1 --tiles-in-flight 1 --no-cpu-baseline --no-fp32-mode --no-power-probe",
         "sequence": int(__import__("time").time())}     # bench.py reads the set with the largest sequence
    json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps(t, indent=1))
