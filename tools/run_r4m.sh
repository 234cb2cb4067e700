cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m; mkdir -p $O
Q="--no-cpu-baseline --no-fp32-mode --no-power-probe --no-extra-workloads"
for P in 0 1; do
  export TL_BLK_PRO=$P
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$P -o p -- python3 bench.py --steps 8 --warmup 2 --tiles-in-flight 1 $Q > $O/line$P.json 2> $O/err$P.txt
  f=$(find $O/s$P -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, os
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("TL_BLK_PRO=%s total kernel time %.3f ms" % (os.environ["TL_BLK_PRO"], tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print("  %-70s calls %5s avg %8.1f us total %8.3f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  find $O/s$P -name "*kernel_trace.csv" -delete
done
