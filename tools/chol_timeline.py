"""Timeline of the Cholesky launches of ONE LM iteration from a rocprofv3 rocpd database (--kernel-trace): start offset,
duration, queue and gap to the previous kernel's end, plus the chain summary (sum of durations / sum of gaps / overlap with
the side stream).  usage: python tools/chol_timeline.py <results.db> [iteration index, default: last full one]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(c.execute("select name,start,end%s from kernels order by start" % (("," + qcol) if qcol else "")))
begins = [i for i, r in enumerate(rows) if "k_ba_iter_begin" in r[0]]
it = int(sys.argv[2]) if len(sys.argv) > 2 else len(begins) - 3
lo, hi = begins[it], begins[it + 1]
seg = rows[lo:hi]
t0 = seg[0][1]
prev_end = {}
tot = {}
print("iteration %d: %d kernels, %.1f us" % (it, len(seg), (seg[-1][2] - t0) / 1e3))
last_end_any = t0
for r in seg:
    name = r[0].split("(")[0].replace("orbhip::", "").replace("void ", "")
    q = r[3] if qcol else 0
    gap = (r[1] - prev_end.get(q, r[1])) / 1e3
    d = (r[2] - r[1]) / 1e3
    tot.setdefault(name, [0, 0.0]); tot[name][0] += 1; tot[name][1] += d
    if "-v" in sys.argv: print("%9.1f  %7.1f us  gap %6.1f  q%-3s %s" % ((r[1] - t0) / 1e3, d, gap, q, name))
    prev_end[q] = r[2]
print("per kernel:")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]): print("  %-28s %4d  %9.1f us  (%.1f us each)" % (k, v[0], v[1], v[1] / v[0]))
chol = [r for r in seg if "k_chol" in r[0]]
if chol:
    print("cholesky span %.1f us, sum of kernel durations %.1f us, queues %s" % ((max(r[2] for r in chol) - min(r[1] for r in chol)) / 1e3,
          sum(r[2] - r[1] for r in chol) / 1e3, sorted(set(r[3] for r in chol)) if qcol else "?"))
