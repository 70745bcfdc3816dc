#!/usr/bin/env python3
"""Table of the kernels' register / LDS / occupancy figures from
cmacionize_amd/csrc/resource_usage.txt (`make asm`)."""
import re
import sys
rows, cur = [], None
for line in open(sys.argv[1] if len(sys.argv) > 1 else "resource_usage.txt"):
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.rsplit(":", 1)
        cur[k.strip()] = v.strip()
print("%-52s %5s %5s %4s %7s %7s %7s" % ("kernel", "VGPR", "SGPR", "occ",
                                        "sspill", "vspill", "LDS"))
for r in rows:
    print("%-52s %5s %5s %4s %7s %7s %7s" % (
        r["name"][:52], r.get("VGPRs"), r.get("TotalSGPRs"),
        r.get("Occupancy [waves/SIMD]"), r.get("SGPRs Spill"),
        r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
