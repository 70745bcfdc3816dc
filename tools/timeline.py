#!/usr/bin/env python3
"""Timeline of one iteration from a `rocprofv3 --kernel-trace` CSV: every
dispatch in start order with its duration and the idle gap before it, for the
last iteration in the trace (from the last direction_key_kernel / first
shoot_kernel on).

    python tools/timeline.py DIR [max_rows]
"""
import csv
import glob
import re
import sys

root = sys.argv[1]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:52]


starts = [i for i, r in enumerate(rows)
          if r["Kernel_Name"].startswith(("direction_key_kernel",
                                          "emission_key_kernel"))]
first = starts[-1] if starts else 0
# the iteration ends before the next reset (fillBuffer after the update)
end = len(rows)
sel = rows[first:end]
t0 = int(sel[0]["Start_Timestamp"])
prev_end = t0
busy = idle = 0
print("%10s %9s %9s  %s" % ("t (us)", "dur (us)", "gap (us)", "kernel"))
agg = {}
for k, r in enumerate(sel):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev_end)
    name = short(r["Kernel_Name"])
    a = agg.setdefault(name, [0, 0, 0])
    a[0] += 1
    a[1] += e - s
    a[2] += gap
    busy += e - s
    idle += gap
    if k < limit:
        print("%10.1f %9.1f %9.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3,
                                          gap / 1e3, name))
    prev_end = max(prev_end, e)
print("\nspan %.2f ms, kernels %.2f ms, idle gaps %.2f ms" %
      ((prev_end - t0) / 1e6, busy / 1e6, idle / 1e6))
print("%-52s %6s %10s %12s" % ("kernel", "calls", "busy ms", "gap-before ms"))
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-52s %6d %10.3f %12.3f" % (name, a[0], a[1] / 1e6, a[2] / 1e6))
