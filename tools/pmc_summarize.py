"""Aggregate a rocprofv3 --pmc counter_collection.csv by kernel name (mean per dispatch)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"].split("(")[0][:70]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in acc.items():
    if not any(k in name for k in sys.argv[2:] or [""]):
        continue
    print(name)
    for c, v in sorted(ctrs.items()):
        print(f"   {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
