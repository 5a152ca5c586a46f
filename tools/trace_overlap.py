"""Concurrency seen in a rocprofv3 kernel trace (rocpd database): per kernel name the mean duration, and over the busy window the
union of all kernel intervals, the sum of durations (mean number of kernels in flight = sum / union) and the mean number in flight per
kernel name.  usage: python tools/trace_overlap.py <run_results.db> [t0_frac t1_frac]   (the fractions cut the window: 0.5 1.0 = second half)"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = list(c.execute("select name,start,end from kernels order by start"))
t_lo, t_hi = min(r[1] for r in rows), max(r[2] for r in rows)
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
w0, w1 = t_lo + f0 * (t_hi - t_lo), t_lo + f1 * (t_hi - t_lo)
rows = [(n, max(s, w0), min(e, w1)) for n, s, e in rows if e > w0 and s < w1]
ev = []
for n, s, e in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; depth = 0; last = None; hist = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last; hist[depth] += t - last
    depth += d; last = t
tot = sum(e - s for _, s, e in rows)
print("window %.1f ms, busy (>= 1 kernel) %.1f ms = %.3f, sum of durations %.1f ms, mean kernels in flight while busy %.2f" % ((w1 - w0) / 1e6, busy / 1e6, busy / (w1 - w0), tot / 1e6, tot / max(busy, 1)))
print("time share by number of kernels in flight:", {k: round(v / (w1 - w0), 3) for k, v in sorted(hist.items())})
acc = collections.defaultdict(list)
for n, s, e in rows: acc[n.split("(")[0][:40]].append(e - s)
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-42s calls %6d  mean %9.1f us  total %8.1f ms  share of window %.3f" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, sum(v) / (w1 - w0)))
