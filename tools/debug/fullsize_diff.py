#!/usr/bin/env python3
"""GPU box: where does the engine differ from the oracle on 256^3 diffuse?
usage: fullsize_diff.py [model] [ncell] [npacket]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from cmacionize_amd import engine as E
import test_gpu_fullsize_oracle as T
from test_gpu_fullsize_physics import converge

model = sys.argv[1] if len(sys.argv) > 1 else "diffuse"
ncell = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1500000
oracle.build()
eng = T.engine_for(model, ncell)
converge(eng, 6, 4000000)
sim = T.oracle_for(oracle, model, ncell, eng)
sim.reset(); sim.totweight = 0.; sim.typecount[:] = 0.
sim.shoot(11, 60, 0, n)
want = T.oracle_accumulators(sim, model)
for tuning in (dict(), dict(tile_rounds=0), dict(tile_rounds=0, pad_march=0),
               dict(exact_dda=1), dict(sort_packets=0, aggregate=0, tile_rounds=0)):
    eng.set_tuning(**tuning)
    eng.reset_grid()
    eng.shoot(11, 60, 0, n)
    tw, tc, ns = eng.get_counters()
    got = T.accumulators(eng, model)
    print(tuning, "tc", tc, "oracle", sim.typecount, flush=True)
    for k, (a, b) in enumerate(zip(got, want)):
        scale = np.abs(b).max()
        if scale == 0.:
            continue
        d = np.abs(a - b)
        bad = d > 1e-9 * np.abs(b) + 1e-13 * scale
        worst = np.argsort(d)[-5:]
        print("  acc %d scale %.3e max|d|/scale %.2e nbad %d  sum rel %.2e" % (
            k, scale, d.max() / scale, bad.sum(),
            abs(a.sum() - b.sum()) / abs(b.sum())))
        for c in worst[::-1]:
            ix, r = divmod(int(c), ncell * ncell)
            iy, iz = divmod(r, ncell)
            print("    cell (%d,%d,%d) gpu %.17g oracle %.17g d/scale %.2e d/b %.2e"
                  % (ix, iy, iz, a[c], b[c], d[c] / scale,
                     d[c] / max(abs(b[c]), 1e-300)))
    # reset the defaults the tuning changed
    eng.set_tuning(tile_rounds=1, pad_march=1, exact_dda=0, sort_packets=1,
                   aggregate=2)
