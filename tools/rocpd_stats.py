"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; this prints / writes the per-kernel summary that
`--stats` used to emit as CSV: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev.
usage: python tools/rocpd_stats.py <run_results.db> [out.csv]"""
import sqlite3, csv, math, sys, collections
c = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(list)
for name, s, e in c.execute("select name,start,end from kernels"):
    acc[name].append(e - s)
tot = sum(sum(v) for v in acc.values())
rows = []
for k, v in acc.items():
    n = len(v); t = sum(v); m = t / n
    sd = math.sqrt(sum((x - m) ** 2 for x in v) / n)
    rows.append((k, n, t, "%.6f" % m, "%.2f" % (100 * t / tot), min(v), max(v), "%.6f" % sd))
rows.sort(key=lambda r: -r[2])
out = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
for r in rows:
    out.writerow(r)
