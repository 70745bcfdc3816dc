#!/usr/bin/env python3
"""GPU box, probe: what would the multi-ion tile kernel gain if its waves
were all "soft" (12 of the 16 adds per step skipped wave-uniformly)?
lexingtonHII40 with (almost) no helium has only hydrogen re-emission - every
re-emitted flight is below 21.6 eV - otherwise the same kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import run_config as R
he = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
R.LEX[0] = he
eng = R.make("lexington", 256)
for loop in range(7):
    eng.reset_grid(); eng.get_timing(reset=True)
    eng.shoot(42, loop, 0, 100000000)
    tw, tc, ns = eng.get_counters()
    launches = eng.get_launch_times()
    tm = eng.get_timing(reset=True)
    eng.update_cells(loop, tw); eng.synchronize()
tiles = [(ms, pk) for ms, pk in launches[1:] if pk > 2e6]
print("He %.3g: shoot %.1f ms, first gen %.1f ms, tile rounds %d: %.1f ms for %.3g visits = %.1f ps/visit; first round %.2f ms / %.3g" % (
    he, tm["shoot_ms"], launches[0][0], len(tiles), sum(m for m, _ in tiles),
    sum(p for _, p in tiles), 1e9 * sum(m for m, _ in tiles) / sum(p for _, p in tiles), tiles[0][0], tiles[0][1]))
