#!/usr/bin/env python3
"""Whole-run rate of the CPU baseline (oracle/cmio_transport_fast.c) at 64^3 on
this machine's cores, set beside the reference's own numbers for the same
runs (the reference built and run by the survey in the
same container, 8 threads, 10^6 packets x 20 iterations; rate = packets over
its "Total photon shooting time" line, src/IonizationSimulation.cpp:667-674).

    python tests/calibrate_cpu_baseline.py [profiles/rNN/cpu_calibration.json]

With a path the record bench.py's `cpu_baseline.calibration` reads is written
there."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402

# BASELINE.md section 2 (packets/s over the reference's shooting time)
REFERENCE = {
    "stromgren": dict(reference_classic=1.24e6, reference_task_based=2.79e6),
    "stromgren_diffuse": dict(reference_classic=0.98e6,
                              reference_task_based=1.84e6),
    "lexington": dict(reference_classic=1.04e6, reference_task_based=1.52e6),
}

O.build()
threads = O.num_threads()
print("threads", threads)
record = {
    "what": "whole 20-iteration run at 64^3, 1e6 packets per iteration, "
            "packets/s over the shooting time: the CPU baseline port "
            "(oracle/cmio_transport_fast.c, this script) and the reference "
            "itself (BASELINE.md section 2) on the same container's cores",
    "threads": threads,
    "host": "build container, %d CPUs" % threads,
    "reference_source": "BASELINE.md section 2",
    "configs": {},
}
for name, mk, npk, its in (
        ("stromgren", lambda: O.stromgren_simulation(64, compact=True),
         1000000, 20),
        ("stromgren_diffuse",
         lambda: O.stromgren_simulation(64, diffuse=True, compact=True),
         1000000, 20),
        ("lexington", lambda: O.lexington_simulation(64), 1000000, 20)):
    sim = mk()
    # the reference's run: 20 iterations from the ionized start; time shooting
    tot_fast = 0.
    for loop in range(its):
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        t0 = time.perf_counter()
        sim.shoot_fast(42, loop, 0, npk)
        tot_fast += time.perf_counter() - t0
        sim.update(loop, sim.totweight)
    rate = its * npk / tot_fast
    print("%-18s whole run %d it: %.2f s -> %.3g packets/s (reference "
          "classic %.3g, task-based %.3g)" %
          (name, its, tot_fast, rate, REFERENCE[name]["reference_classic"],
           REFERENCE[name]["reference_task_based"]))
    record["configs"][name] = dict(port=rate, iterations=its,
                                   packets_per_iteration=npk,
                                   shooting_time_s=tot_fast,
                                   **REFERENCE[name])
if len(sys.argv) > 1:
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    with open(sys.argv[1], "w") as f:
        json.dump(record, f, indent=1)
        f.write("\n")
