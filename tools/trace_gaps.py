"""GPU timeline of a profiled run: how much of a step the device is busy, and where it idles.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3 bench.py --workload config3 ...
    python tools/trace_gaps.py DIR [marker_kernel_substring] > summary.txt

Steps are cut at every dispatch of the marker kernel (default `k_conv_in4`: the input conv, once per forward).  Per step: wall time,
busy time (union of the kernel intervals over all streams), idle time, the sum of kernel durations (> busy when streams overlap), the
kernel families by time, and the idle gaps attributed to the kernel that ENDED before the gap (launch-bound chains show up as many small
gaps behind small kernels; host read-backs as few large ones).
"""
import collections
import csv
import glob
import os
import re
import sys


def family(name):
    m = re.search(r"(k_[A-Za-z0-9_]+)", name)
    if m:
        return m.group(1)
    if "at::native" in name or "at::cuda" in name:
        m = re.search(r"at::native::(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+)", name)
        return "aten:" + (m.group(1) if m else "?")
    if "rocprim" in name:
        return "rocprim"
    return name[:40]


def main():
    src = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "k_conv_in4"
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + src)
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if marker in r[2]]
    print(f"{len(rows)} dispatches, {len(cuts)} '{marker}' markers")
    if len(cuts) < 2:
        cuts = [0, len(rows) - 1]
    for si in range(len(cuts) - 1):
        seg = rows[cuts[si]:cuts[si + 1]]
        t0, t1 = seg[0][0], max(r[1] for r in seg)
        t1 = min(t1, rows[cuts[si + 1]][0])
        busy = 0
        cur_end = t0
        gaps = collections.defaultdict(lambda: [0, 0])
        last_name = seg[0][2]
        last_end_name = last_name
        for s, e, n in seg:
            if s > cur_end:
                g = gaps[family(last_end_name)]
                g[0] += s - cur_end; g[1] += 1
                cur_end = s
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
                last_end_name = n
        ksum = sum(e - s for s, e, _ in seg)
        fam = collections.defaultdict(lambda: [0, 0])
        for s, e, n in seg:
            f = fam[family(n)]
            f[0] += e - s; f[1] += 1
        wall = t1 - t0
        print(f"\n== step {si}: wall {wall / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(wall - busy) / 1e6:.2f} ms, "
              f"sum of kernel durations {ksum / 1e6:.2f} ms, {len(seg)} dispatches")
        if si not in (len(cuts) - 2, len(cuts) - 3):
            continue
        print("  kernel families (ms, dispatches):")
        for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:45]:
            print(f"    {k:44s} {v[0] / 1e6:8.3f} {v[1]:5d}")
        ind = collections.defaultdict(lambda: [0, 0, 0])
        for s_, e_, n_ in seg:
            if "k_" in n_:
                k_ = re.sub(r"\(anonymous namespace\)::|void |\(ConvP.*|\(unsigned short const.*|\(void const.*|\(float const.*|\(double const.*", "", n_)[:90]
                d_ = ind[k_]; d_[0] += e_ - s_; d_[1] += 1; d_[2] = max(d_[2], e_ - s_)
        print("  individual kernels (total ms, dispatches, mean us, max us):")
        for k, v in sorted(ind.items(), key=lambda kv: -kv[1][0])[:40]:
            print(f"    {v[0] / 1e6:8.3f} {v[1]:5d} {v[0] / v[1] / 1e3:8.1f} {v[2] / 1e3:8.1f}  {k}")
        at = collections.defaultdict(lambda: [0, 0])
        for s_, e_, n_ in seg:
            if "at::" in n_ or "rocclr" in n_:
                a_ = at[re.sub(r"\s+", " ", n_)[:230]]
                a_[0] += e_ - s_; a_[1] += 1
        print("  ATen / runtime kernels by full name (ms, dispatches):")
        for k, v in sorted(at.items(), key=lambda kv: -kv[1][0])[:24]:
            print(f"    {v[0] / 1e6:8.3f} {v[1]:5d}  {k}")
        print("  idle attributed to the kernel before the gap (ms, gaps):")
        for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
            print(f"    {k:44s} {v[0] / 1e6:8.3f} {v[1]:5d}")


if __name__ == "__main__":
    main()
