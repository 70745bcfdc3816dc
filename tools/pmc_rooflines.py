#!/usr/bin/env python3
"""From the counter passes of tools/pmc_profile.sh over `bench.py --config C
--steps K --warmup 0 --no-cpu-baseline` (one directory per config), the
per-launch figures bench.py's roofline uses and a per-kernel table of the
timed phase:

    CMI_PROFILE_COMMIT=$(git rev-parse --short HEAD) \
        python tools/pmc_rooflines.py OUTDIR K > profiles/r03/counters.json
    (OUTDIR/pmc_<config>/pass*/**/*counter_collection.csv)

Units (MI355X_MICROARCH.md): SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count
quad-cycles (x4 = cycles summed over SIMDs resp. waves); SQ_BUSY_CYCLES is
summed over the 32 shader engines; GRBM_GUI_ACTIVE over the 8 XCDs (/8 = the
kernel's cycles); FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE counts 128-B
requests as 64 B on gfx950 (x2); TCC_EA0_ATOMIC = 64-B atomic requests that
reach the memory side.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root, steps = sys.argv[1], int(sys.argv[2])
# (the grid the profile was taken on: beyond 2^25 cells the hydrogen-only
# first generation is the BIG build - last template argument true)
NCELL = int(sys.argv[3]) if len(sys.argv) > 3 else 256
# (template arguments: FULL, HEAT, REEMIT, EXACT, TABLE, PRE, PAD, TRACK -
# the first generation of an iteration runs the TABLE variant)
DOMINANT = {"stromgren":
                "shoot_kernel<false, false, false, false, true, false, true, false, false>",
            "stromgren_diffuse":
                "shoot_kernel<false, false, false, false, true, false, true, false, false>",
            "lexington":
                "shoot_kernel<true, true, false, false, true, true, false, false, false>"}
if NCELL ** 3 > 1 << 25:
    for _c in ("stromgren", "stromgren_diffuse"):
        DOMINANT[_c] = DOMINANT[_c].replace("true, false, false>",
                                            "true, false, true>")
N_SIMD, N_CU, MAXCLK = 1024, 256, 2.4e9


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    if "rocprim" in name:
        m = re.search(r"(radix_sort_onesweep|radix_sort_\w+?)[_<]", name)
        return "rocprim::" + (m.group(1) if m else "sort")
    return name[:80]


def load(config):
    """{pass: {dispatch: {"name", "t0", "t1", counters...}}}"""
    passes = {}
    for d in sorted(glob.glob(os.path.join(root, "pmc_" + config, "pass*"))):
        if not os.path.isdir(d):
            continue
        rows = {}
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                x = rows.setdefault(int(r["Dispatch_Id"]), {
                    "name": short(r["Kernel_Name"]),
                    "t0": int(r["Start_Timestamp"]),
                    "t1": int(r["End_Timestamp"])})
                x[r["Counter_Name"]] = x.get(r["Counter_Name"], 0.) + \
                    float(r["Counter_Value"])
        if rows:
            passes[os.path.basename(d)] = rows
    return passes


def timed_phase(rows, dominant):
    """dispatch ids of the timed phase: from the first of the `steps` longest
    launches of the dominant kernel among the last launches on."""
    ids = sorted(d for d, r in rows.items() if r["name"] == dominant)
    if not ids:
        return [], []
    tail = ids[-max(steps * 40, 1):]
    # the first generation is the longest launch of an iteration; the timed
    # iterations are the last `steps` of them
    per_iter = sorted(tail, key=lambda d: rows[d]["t1"] - rows[d]["t0"])
    # converged iterations all take about the same time: the longest launches
    # among the last ones, restricted to the run's end
    cut = 0.5 * (rows[per_iter[-1]]["t1"] - rows[per_iter[-1]]["t0"])
    long_ones = [d for d in tail if rows[d]["t1"] - rows[d]["t0"] >= cut]
    first_gen = long_ones[-steps:]
    phase = [d for d in sorted(rows) if d >= first_gen[0]]
    return first_gen, phase


out = {}
for config, dominant in DOMINANT.items():
    passes = load(config)
    if not passes:
        continue
    entry = {"ncell": NCELL, "steps_profiled": steps, "dominant": {},
             "other_kernels": [],
             # the source the profiled library was built from
             "commit": os.environ.get("CMI_PROFILE_COMMIT")}
    merged = {}   # counter -> per-launch mean over the first-gen launches
    table = defaultdict(dict)
    for pname, rows in sorted(passes.items()):
        first_gen, phase = timed_phase(rows, dominant)
        if not first_gen:
            continue
        for c in rows[first_gen[0]]:
            if c in ("name", "t0", "t1"):
                continue
            merged[pname + ":" + c] = sum(rows[d].get(c, 0.)
                                          for d in first_gen) / len(first_gen)
        merged[pname + ":ns"] = sum(rows[d]["t1"] - rows[d]["t0"]
                                    for d in first_gen) / len(first_gen)
        # per kernel of the timed phase, per iteration
        agg = defaultdict(lambda: defaultdict(float))
        for d in phase:
            r = rows[d]
            a = agg[r["name"]]
            a["ns"] += r["t1"] - r["t0"]
            a["calls"] += 1
            for c, v in r.items():
                if c not in ("name", "t0", "t1"):
                    a[c] += v
        for name, a in agg.items():
            for c, v in a.items():
                table[name][pname + ":" + c] = v / steps

    def m(counter, src=merged):
        for k, v in src.items():
            if k.split(":", 1)[1] == counter:
                return k.split(":")[0], v
        return None, None

    def frac_of(src):
        """utilisation fractions from one kernel's merged counters"""
        res = {}
        p, valu = m("SQ_ACTIVE_INST_VALU", src)
        if p:
            cycles = src[p + ":GRBM_GUI_ACTIVE"] / 8.
            res["kernel_cycles"] = cycles
            res["clock_GHz"] = cycles / src[p + ":ns"]
            res["valu_busy"] = 4. * valu / (N_SIMD * cycles)
            res["wave_wait_frac"] = src[p + ":SQ_WAIT_ANY"] / \
                src[p + ":SQ_WAVE_CYCLES"]
            res["valu_insts"] = src[p + ":SQ_INSTS_VALU"]
        p, lds = m("SQ_LDS_IDX_ACTIVE", src)
        if p:
            cycles = src[p + ":GRBM_GUI_ACTIVE"] / 8.
            res["lds_busy"] = lds / (N_CU * cycles)
            res["lds_bank_conflict_frac"] = \
                src[p + ":SQ_LDS_BANK_CONFLICT"] / max(lds, 1.)
            res["salu_insts"] = src[p + ":SQ_INSTS_SALU"]
        return res

    dom = frac_of(merged)
    p, _ = m("SQ_ACTIVE_INST_VALU")
    ns = merged[p + ":ns"]
    dom["kernel_ms"] = ns * 1e-6
    dom["valu_busy_cycles"] = 4. * merged[p + ":SQ_ACTIVE_INST_VALU"]
    pl, lds = m("SQ_LDS_IDX_ACTIVE")
    dom["lds_busy_cycles"] = lds
    pa, atom = m("TCC_EA0_ATOMIC_sum")
    dom["atomic_requests"] = atom
    pf, fetch = m("FETCH_SIZE")
    pw, write = m("WRITE_SIZE")
    dom["hbm_bytes"] = (2. * fetch + write) * 1024.
    dom["kernel"] = dominant
    dom["launches_averaged"] = steps
    entry["dominant"] = dom
    for name, src in sorted(table.items(),
                            key=lambda kv: -max([v for k, v in kv[1].items()
                                                 if k.endswith(":ns")] + [0])):
        p, ns = m("ns", src)
        if not p or ns * 1e-6 < 0.05:
            continue
        # fractions need per-kernel sums of the same pass: reuse frac_of on
        # the per-iteration sums (ratios are unaffected by the division)
        f = frac_of(src)
        row = {"kernel": name, "ms_per_iteration": ns * 1e-6,
               "calls_per_iteration": src.get(p + ":calls")}
        for k in ("valu_busy", "lds_busy", "wave_wait_frac", "clock_GHz"):
            if k in f:
                row[k] = f[k]
        if "valu_insts" in f:
            # vector instructions per iteration (one per wave = 64 lane
            # operations) and the rate of lane operations
            row["valu_insts_per_iteration"] = f["valu_insts"]
            pv, _ = m("SQ_INSTS_VALU", src)
            row["valu_lane_ops_per_s"] = 64. * f["valu_insts"] / \
                (src[pv + ":ns"] * 1e-9)
        pa, atom = m("TCC_EA0_ATOMIC_sum", src)
        if pa:
            row["atomic_requests_per_s"] = atom / (src[pa + ":ns"] * 1e-9)
        pf, fetch = m("FETCH_SIZE", src)
        pw, write = m("WRITE_SIZE", src)
        if pf and pw:
            row["hbm_GBps"] = (2. * fetch * 1024. / (src[pf + ":ns"]) +
                               write * 1024. / (src[pw + ":ns"]))
        entry["other_kernels"].append(row)
    out[config] = entry
json.dump(out, sys.stdout, indent=1)
print()
