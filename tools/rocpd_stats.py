"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; this prints / writes the per-kernel summary that
`--stats` used to emit as CSV: Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev.
usage: python tools/rocpd_stats.py <run_results.db> [out.csv]
The database is read COMPLETELY before the output file is touched, the CSV goes to a temporary name and replaces the target
only when it holds at least one kernel row: a trace that died (no `kernels` table) can no longer leave a 0-byte file behind."""
import sqlite3, csv, math, os, sys, collections


def kernel_rows(db):
    if not os.path.exists(db) or os.path.getsize(db) == 0:
        raise SystemExit("rocpd_stats: %s is missing or empty" % db)
    c = sqlite3.connect(db)
    try:
        it = list(c.execute("select name,start,end from kernels"))
    except sqlite3.OperationalError as e:
        raise SystemExit("rocpd_stats: %s holds no kernel trace (%s)" % (db, e))
    acc = collections.defaultdict(list)
    for name, s, e in it:
        acc[name].append(e - s)
    if not acc:
        raise SystemExit("rocpd_stats: %s holds no kernel dispatches" % db)
    tot = sum(sum(v) for v in acc.values())
    rows = []
    for k, v in acc.items():
        n = len(v); t = sum(v); m = t / n
        sd = math.sqrt(sum((x - m) ** 2 for x in v) / n)
        rows.append((k, n, t, "%.6f" % m, "%.2f" % (100 * t / tot), min(v), max(v), "%.6f" % sd))
    rows.sort(key=lambda r: -r[2])
    return rows


def write_csv(rows, f):
    out = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        out.writerow(r)


if __name__ == "__main__":
    rows = kernel_rows(sys.argv[1])
    if len(sys.argv) > 2:
        tmp = sys.argv[2] + ".tmp"
        with open(tmp, "w", newline="") as f:
            write_csv(rows, f)
        os.replace(tmp, sys.argv[2])
    else:
        write_csv(rows, sys.stdout)
