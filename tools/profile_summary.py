#!/usr/bin/env python3
"""Compact summary of a `rocprofv3 --kernel-trace --stats` directory made by
tools/round_measure.sh: the kernel_stats table with the template noise cut from
the names, plus the durations of the bench's TIMED launches of the dominant
kernel (the last `steps` of the process; the all-calls average also holds the
untimed iterations that bring the grid to its converged state, whose packets
fly further) next to the HIP-event figure bench.py printed in the same run.

    python tools/profile_summary.py OUT/stats_stromgren OUT/stats_stromgren.log \
        > profiles/r02/bench_stromgren_kernel_stats.txt
"""
import csv
import glob
import json
import re
import sys

root, log = sys.argv[1], sys.argv[2]


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", name)
    m = re.search(r"(radix_sort_onesweep_\w+)", name)
    if "rocprim" in name and m:
        return "rocprim " + m.group(1)
    return name[:90]


stats = glob.glob(root + "/**/*kernel_stats.csv", recursive=True)[0]
print("# %s" % stats.split("/")[-1])
print("%-60s %6s %14s %12s %7s %12s %12s" %
      ("kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"))
agg = {}
for r in csv.DictReader(open(stats)):
    k = short(r["Name"])
    a = agg.setdefault(k, [0, 0, 0., 1e30, 0])
    a[0] += int(r["Calls"])
    a[1] += int(r["TotalDurationNs"])
    a[2] += float(r["Percentage"])
    a[3] = min(a[3], int(r["MinNs"]))
    a[4] = max(a[4], int(r["MaxNs"]))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %6d %14d %12d %7.2f %12d %12d" %
          (k[:60], a[0], a[1], a[1] // a[0], a[2], a[3], a[4]))

bench = json.loads([l for l in open(log) if l.startswith("{")][-1])
# the dominant kernel of the bench line (the first generation of every
# iteration); the timed launches are the last `steps` of the process
dominant = bench["roofline"]["kernel"]
n = bench["steps"]
trace = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(trace))
        if r["Kernel_Name"].replace("void ", "").startswith(dominant + "(")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
timed = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
         for r in rows[-n:]]
print()
print("# %s: %d launches in the process; the %d timed ones (ns): total %d, "
      "average %d" % (dominant, len(rows), n, sum(timed), sum(timed) // n))
print("# bench.py (HIP events, same run): roofline.kernel_avg_ms %.4f, "
      "transport_only %.4g packets/s, value %.4g packets/s" %
      (bench["roofline"]["kernel_avg_ms"],
       bench["transport_only_packets_per_s"], bench["value"]))
print("# timed launches (ns): " + " ".join(str(t) for t in timed))
