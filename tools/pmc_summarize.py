"""Aggregate a rocprofv3 --pmc counter_collection.csv by kernel name (mean per dispatch).
   python tools/pmc_summarize.py file.csv [name-filter ...]      human-readable
   python tools/pmc_summarize.py file.csv --json                 {kernel: {counter: mean, "n": dispatches}}"""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
if "--json" in sys.argv:
    out = {}
    for name, ctrs in acc.items():
        key = name.split("(")[0].replace("void ", "")
        out[key] = {c: sum(v) / len(v) for c, v in ctrs.items()}
        out[key]["n"] = max(len(v) for v in ctrs.values())
    json.dump(out, sys.stdout, indent=1)
    sys.exit(0)
for name, ctrs in acc.items():
    if not any(k in name for k in sys.argv[2:] or [""]):
        continue
    print(name.split("(")[0][:90])
    for c, v in sorted(ctrs.items()):
        print(f"   {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
