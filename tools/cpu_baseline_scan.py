#!/usr/bin/env python3
"""CPU baseline (oracle/cmio_transport_fast.c) on this host: packets/s of the
stromgren transport loop against the thread count and the half-width of the
cube of per-thread private accumulators around the source
(CMIO_FAST_HOT_RADIUS). Test infrastructure, like everything under oracle/.

    python tools/cpu_baseline_scan.py [ncell] [packets per thread] [x_H.npy]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 128
per_thread = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
state = sys.argv[3] if len(sys.argv) > 3 else None
O.build()
sim = O.stromgren_simulation(ncell, compact=True)
if state:
    # x_H of a converged run (tools/converged_state.py)
    sim.x[0][:] = np.load(state)
else:
    # an ionized sphere of the converged run's volume (0.36 of the box)
    ax = (np.arange(ncell) + 0.5) / ncell - 0.5
    r = np.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 +
                ax[None, None, :] ** 2).ravel()
    sim.x[0][:] = np.where(r < 0.442, 1e-3, 1.)
all_threads = O.num_threads()
print("host threads: %d, grid %d^3" % (all_threads, ncell))
for threads in sorted({all_threads, max(all_threads // 2, 1),
                       max(all_threads // 4, 1), min(8, all_threads)}):
    O.set_num_threads(threads)
    for radius in (0, 4, 8, 16, 32):
        os.environ["CMIO_FAST_HOT_RADIUS"] = str(radius)
        n = per_thread * threads
        sim.reset()
        t0 = time.perf_counter()
        sim.shoot_fast(42, 7, 0, n)
        dt = time.perf_counter() - t0
        print("threads %4d  radius %3d  %10.4g packets/s" %
              (threads, radius, n / dt), flush=True)
O.set_num_threads(all_threads)
