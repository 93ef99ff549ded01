"""summarise a rocprofv3 --pmc counter_collection.csv per kernel (development aid)"""
import collections
import csv
import sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    calls[(k, row["Counter_Name"])] += 1
for k, v in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k, {c: "%.4g (x%d)" % (x, calls[(k, c)]) for c, x in v.items()})
