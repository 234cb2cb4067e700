"""Turn the rocprofv3 outputs of a bench.py run into the files kept under profiles/<tag>/.

    python tools/summarize_profile.py <gpurun_out/dir> <profiles/tag> <steps+warmup forwards profiled>

Expects in <dir>: stats/*kernel_stats.csv (--kernel-trace --stats), fetch/*counter_collection.csv (--pmc FETCH_SIZE),
write/*counter_collection.csv (--pmc WRITE_SIZE), bench_line.json.  PMC units are KB; gfx950 FETCH_SIZE is doubled
(MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, shutil, sys, collections

src, dst, forwards = sys.argv[1], sys.argv[2], int(sys.argv[3])
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
        if "k_conv" not in name:
            continue
        name = '"' + name[name.index("k_conv"):].split("(ConvP")[0] + '"'
        per[name][0] += 1; per[name][1] += float(r["Counter_Value"]); total += float(r["Counter_Value"])
    with open(os.path.join(dst, out_name), "w") as f:
        f.write("kernel,dispatches,%s_kb_total\n" % counter.lower())
        for k, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k},{n},{v:.1f}\n")
    return total


fetch = pmc("fetch", "FETCH_SIZE", "pmc_fetch_size.csv")
write = pmc("write", "WRITE_SIZE", "pmc_write_size.csv")
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl):
    shutil.copy(bl, os.path.join(dst, "bench_line.json"))
if fetch is not None and write is not None:
    t = {"dtype": "bf16", "workload": "config2", "forwards_profiled": forwards,
         "fetch_size_kb_per_step": fetch / forwards, "write_size_kb_per_step": write / forwards,
         "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> doubled (MI355X_MICROARCH.md HBM section); units KB",
         "hbm_gb_per_step": (2 * fetch + write) / forwards * 1024 / 1e9,
         "command": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"}
    json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps(t, indent=1))
