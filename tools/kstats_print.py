"""Print the top rows of a tools/rocpd_stats.py CSV: kernel, calls, mean duration, share.  usage: kstats_print.py <stats.csv> [rows]"""
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))[1:]
tot = sum(float(r[2]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print("%-34s calls %6s avg %9.1f us  %5.1f%%" % (r[0].split("(")[0].replace("orbhip::", "").replace("void ", "")[:34], r[1], float(r[3]) / 1e3, 100 * float(r[2]) / tot))
print("total kernel ms %.1f" % (tot / 1e6))
