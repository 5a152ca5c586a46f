"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean counter value per launch."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in cs.items():
        print("   %-28s n=%-4d mean %.4g" % (c, len(v), sum(v) / len(v)))
