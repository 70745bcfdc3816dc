#!/usr/bin/env python3
"""Summarise the counter_collection csv files of tools/pmc_profile.sh: per
kernel name, the sum of every counter over the dispatches and the mean per
dispatch; optionally only the largest dispatches of a kernel (--top N)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "shoot_kernel"
rows = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            rows[r["Counter_Name"]][r["Dispatch_Id"]].append(
                float(r["Counter_Value"]))
for name in sorted(rows):
    per_dispatch = [sum(v) for v in rows[name].values()]
    per_dispatch.sort()
    top = per_dispatch[-3:]
    print("%-24s dispatches %3d  max3 mean %.6g  total %.6g" %
          (name, len(per_dispatch), sum(top) / len(top), sum(per_dispatch)))
