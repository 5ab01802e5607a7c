"""Per-kernel table of a rocprofv3 --kernel-trace --stats directory: calls, average / total time; optional name filter (regex)."""
import csv, glob, os, re, sys
root = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    print(f"-- {os.path.relpath(f, root)}")
    tot = 0.0
    for r in rows:
        name = r.get("Name", "")
        if pat and not pat.search(name):
            continue
        t = float(r.get("TotalDurationNs", 0)) / 1e6
        tot += t
        print(f"{name[:110]:110s} {r.get('Calls',''):>6s} avg_us {float(r.get('AverageNs',0))/1e3:10.2f} total_ms {t:9.3f}")
    print(f"{'sum of the rows shown':110s} {'':6s} {'':17s} total_ms {tot:9.3f}")
