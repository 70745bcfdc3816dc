#!/usr/bin/env python3
"""Whole-run rate of the CPU baseline (oracle/cmio_transport_fast.c) at 64^3 on
this machine's cores, to set beside the reference's own numbers for the same
runs: the figures in bench.py's CALIBRATION table."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle_lib as O
O.build()
print("threads", O.num_threads())
for name, mk, npk in (("stromgren", lambda: O.stromgren_simulation(64, compact=True), 1000000),
                      ("diffuse", lambda: O.stromgren_simulation(64, diffuse=True, compact=True), 1000000),
                      ("lexington", lambda: O.lexington_simulation(64), 300000)):
    sim = mk()
    # the reference's run: 20 iterations from the ionized start; time shooting
    tot_fast = 0.
    its = 20 if name != "lexington" else 8
    for loop in range(its):
        sim.reset(); sim.totweight = 0.; sim.typecount[:] = 0.
        t0 = time.perf_counter(); sim.shoot_fast(42, loop, 0, npk); tot_fast += time.perf_counter() - t0
        sim.update(loop, sim.totweight)
    # the SoA/16-atomics oracle on the final state
    sim.reset(); t0 = time.perf_counter(); sim.shoot(42, 99, 0, npk // 4); t_old = (time.perf_counter() - t0) * 4
    sim.reset(); t0 = time.perf_counter(); sim.shoot_fast(42, 99, 0, npk); t_new = time.perf_counter() - t0
    print("%-10s whole run %d it: %.2f s -> %.3g packets/s | converged state: fast %.3g packets/s, oracle %.3g packets/s" %
          (name, its, tot_fast, its * npk / tot_fast, npk / t_new, npk / t_old))
