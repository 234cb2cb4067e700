"""Timeline of ONE lone forward from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`, one tile at a time): every dispatch
between two occurrences of the marker kernel (default: the first big kernel of a forward, k_minmax*) with its start offset, duration and
the idle gap in front of it; then the totals per phase (geometry = up to the input conv, network, head).

    python tools/trace_forward.py DIR [which forward, default the last complete one] [marker]
"""
import csv, glob, os, re, sys


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+)(<[^>]*>)?", name)
    if m:
        return (m.group(1) + (m.group(2) or ""))[:64]
    return name[:64]


src = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
marker = sys.argv[3] if len(sys.argv) > 3 else "k_minmax"          # (k_minmax_one in tl_forward's single-tile form, k_minmax otherwise)
files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(files[0])))
cuts = [i for i, r in enumerate(rows) if marker in r[2]]
seg = rows[cuts[which]:cuts[which + 1]] if which + 1 != 0 else rows[cuts[which]:]
t0 = seg[0][0]
end = t0
busy = 0
phase = "geometry"
tot = {"geometry": [0, 0, 0], "network": [0, 0, 0], "head": [0, 0, 0]}      # kernel time, gap time, dispatches
print(f"# forward {which} of {len(cuts)}: {len(seg)} dispatches, {1e-6 * (max(r[1] for r in seg) - t0):.3f} ms from first start to last end")
print("#   start us     dur us   gap us  kernel")
for s, e, n in seg:
    nm = short(n)
    if phase == "geometry" and ("k_conv" in nm):
        phase = "network"
    if "k_head" in nm:
        phase = "head"
    gap = max(0, s - end)
    tot[phase][0] += e - s; tot[phase][1] += gap; tot[phase][2] += 1
    print(f"{1e-3 * (s - t0):12.1f} {1e-3 * (e - s):10.1f} {1e-3 * gap:8.1f}  {nm}")
    end = max(end, e)
for k, (kt, gt, nd) in tot.items():
    print(f"# {k:9s}: {nd:3d} dispatches, kernel time {1e-6 * kt:.3f} ms, idle gaps in front of them {1e-6 * gt:.3f} ms")
