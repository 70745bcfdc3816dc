#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/round_measure.sh into
profiles/traffic.json (fabric bytes per DDA step of the transport kernel).

The timed launches of a bench.py run are the LAST `kernel_launches`
shoot_kernel dispatches of the process; the bench line in the pass's log gives
their number and the DDA steps they executed. Corrections as the MI355X guide
prescribes: rocprofv3 reports both counters in KiB... (x1024), FETCH_SIZE
tallies 128-B read requests at 64 B on gfx950 (x2).

    python tools/pmc_traffic.py gpurun_out/r01_final profiles/traffic.json
"""
import csv
import glob
import json
import sys

root, out = sys.argv[1], sys.argv[2]
result = {}
for cfg in ("stromgren", "stromgren_diffuse", "lexington"):
    entry = {"ncell": 256}
    for counter, factor in (("FETCH_SIZE", 2. * 1024.), ("WRITE_SIZE", 1024.)):
        line = [l for l in open("%s/pmc_%s_%s.log" % (root, cfg, counter))
                if l.startswith("{")][-1]
        bench = json.loads(line)
        launches = bench["roofline"]["kernel_launches"]
        steps = bench["roofline"]["dda_steps_per_launch"] * launches
        per_dispatch = {}
        for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" %
                           (root, cfg, counter), recursive=True):
            for r in csv.DictReader(open(f)):
                if "shoot_kernel" in r["Kernel_Name"] and \
                        r["Counter_Name"] == counter:
                    d = int(r["Dispatch_Id"])
                    per_dispatch[d] = per_dispatch.get(d, 0.) + \
                        float(r["Counter_Value"])
        last = sorted(per_dispatch)[-launches:]
        total = sum(per_dispatch[d] for d in last) * factor
        entry[counter.lower() + "_bytes_per_dda_step"] = total / steps
        entry["dispatches"] = launches
    entry["hbm_bytes_per_dda_step"] = (entry["fetch_size_bytes_per_dda_step"] +
                                       entry["write_size_bytes_per_dda_step"])
    result[cfg] = entry
json.dump(result, open(out, "w"), indent=1)
print(json.dumps(result, indent=1))
