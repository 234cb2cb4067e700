"""Two-stream view of a profiled training step (rocprofv3 --kernel-trace CSV): per kernel family on the main queue, its duration next to a
side-queue kernel and alone; the short kernels that a side-queue kernel slows down most; the largest gaps of the main queue.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3 bench.py --workload config3 --steps 3 --warmup 2
    python tools/trace_overlap.py DIR/p_kernel_trace.csv
"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_conv_in4" in r["Kernel_Name"]]
step = rows[idx[-2]:idx[-1]]
qs = collections.Counter(r["Queue_Id"] for r in step)
mainq = qs.most_common(1)[0][0]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)


def fam(n, full=False):
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>]*>)?)" if full else r"(k_[A-Za-z0-9_]+)", n)
    return m.group(1) if m else re.sub(r"void |at::native::|\(anonymous namespace\)::", "", n)[:40]


print(f"step wall {(t1 - t0) / 1e6:.2f} ms; dispatches per queue {dict(qs)}")
for q in qs:
    print(f"  queue {q}: sum of kernel durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step if r['Queue_Id'] == q) / 1e6:.2f} ms")
side = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam(r["Kernel_Name"], True)) for r in step if r["Queue_Id"] != mainq]
ms = [r for r in step if r["Queue_Id"] == mainq]
st = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
slow = collections.defaultdict(list)
for r in ms:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); f = fam(r["Kernel_Name"])
    o = [n for ss, se, n in side if ss < e and se > s]
    if o:
        st[f][0] += 1; st[f][1] += (e - s) / 1e3
        if "finish" in f or "reduce" in f or "state" in f: slow[(f, o[0])].append((e - s) / 1e3)
    else:
        st[f][2] += 1; st[f][3] += (e - s) / 1e3
print(f"{'main-queue family':34s} {'n next to side':>14s} {'mean us':>9s} {'n alone':>8s} {'mean us':>9s} {'total ms':>9s}")
for f, v in sorted(st.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:24]:
    print(f"{f:34s} {v[0]:14d} {v[1] / max(v[0], 1):9.1f} {v[2]:8d} {v[3] / max(v[2], 1):9.1f} {(v[1] + v[3]) / 1e3:9.2f}")
print("short main-queue kernels next to a side-queue kernel:")
for k, v in sorted(slow.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"  {k[0]:22s} next to {k[1]:44s} n {len(v):3d} mean {sum(v) / len(v):7.1f} us")
gaps = []
prev_e, prev_n = t0, "-"
for r in ms:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > prev_e: gaps.append(((s - prev_e) / 1e3, prev_n, fam(r["Kernel_Name"])))
    if e > prev_e: prev_e, prev_n = e, fam(r["Kernel_Name"])
print(f"main queue idle {sum(g[0] for g in gaps) / 1e3:.2f} ms in {len(gaps)} gaps; largest:")
for g in sorted(gaps, reverse=True)[:14]:
    print(f"  {g[0]:8.1f} us after {g[1]:30s} before {g[2]}")
