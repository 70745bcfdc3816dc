#!/usr/bin/env python3
"""Per-kernel table from the counter passes of tools/pmc_profile.sh (any
script): for every kernel the mean over its LAST `n` launches of the duration
and of every counter, plus the derived utilisations of MI355X_MICROARCH.md.

    python tools/pmc_kernel_table.py OUTDIR [n=3] [name filter]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 3
flt = sys.argv[3] if len(sys.argv) > 3 else ""
N_SIMD, N_CU = 1024, 256


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


table = defaultdict(dict)  # kernel -> counter -> mean
for d in sorted(glob.glob(os.path.join(root, "pass*"))):
    if not os.path.isdir(d):
        continue
    rows = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            x = rows.setdefault(int(r["Dispatch_Id"]), {
                "name": short(r["Kernel_Name"]),
                "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            x[r["Counter_Name"]] = x.get(r["Counter_Name"], 0.) + \
                float(r["Counter_Value"])
    by = defaultdict(list)
    for i in sorted(rows):
        by[rows[i]["name"]].append(rows[i])
    for name, lst in by.items():
        lst = lst[-last:]
        for key in lst[0]:
            if key == "name":
                continue
            vals = [x.get(key, 0.) for x in lst]
            k = key if key != "ns" else "ns_" + os.path.basename(d)
            table[name][k] = sum(vals) / len(vals)

for name, c in sorted(table.items()):
    if flt not in name:
        continue
    print(name)
    ns = [v for k, v in c.items() if k.startswith("ns_")]
    ms = sum(ns) / len(ns) * 1e-6
    print("   ms %.3f" % ms)
    cyc = c.get("GRBM_GUI_ACTIVE", 0.) / 8.
    if cyc:
        print("   clock %.2f GHz" % (cyc / (ms * 1e-3) / 1e9))
        for key, unit in (("SQ_ACTIVE_INST_VALU", N_SIMD),
                          ("SQ_ACTIVE_INST_LDS", N_CU)):
            if key in c:
                print("   %-22s busy %.3f" % (key, 4. * c[key] / (cyc * unit)))
        if "SQ_WAVE_CYCLES" in c:
            for key in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
                        "SQ_ACTIVE_INST_ANY"):
                if key in c:
                    print("   %-22s / wave cycles %.3f" %
                          (key, c[key] / c["SQ_WAVE_CYCLES"]))
    for key in sorted(c):
        if not key.startswith("ns_"):
            print("   %-24s %.4g" % (key, c[key]))
