#!/usr/bin/env python3
"""Per-kernel device time per iteration from a `rocprofv3 --kernel-trace
--stats` run of tools/run_config.py:

    python tools/kernel_breakdown.py DIR ITERATIONS
"""
import csv
import glob
import re
import sys

root, iters = sys.argv[1], float(sys.argv[2])
f = glob.glob(root + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
total = sum(int(r["TotalDurationNs"]) for r in rows)
print("%-72s %7s %12s %11s %6s" % ("kernel", "calls", "ms/iteration",
                                    "avg us", "%"))
for r in rows[:18]:
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", r["Name"])
    n = re.sub(r"void ", "", n)[:72]
    print("%-72s %7s %12.2f %11.1f %6.1f" % (
        n, r["Calls"], int(r["TotalDurationNs"]) / 1e6 / iters,
        int(r["TotalDurationNs"]) / 1e3 / int(r["Calls"]),
        float(r["Percentage"])))
print("all kernels: %.1f ms per iteration" % (total / 1e6 / iters))
