#!/usr/bin/env python3
"""GPU box: the set-up of tests/test_continuous_sources.py::
test_continuous_source_on_a_decomposed_grid with switches, typecounts against
the oracle.  usage: cont_decomposed.py [key=value ...] [heat=0|1] [cont=0|1]
[blocks=2,1,2]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
from cmacionize_amd import STROMGREN as S, GpuEngine
from cmacionize_amd import engine as E
from cmacionize_amd.engine import EngineGroup
from cmacionize_amd.simulation import DomainDecomposition, DomainGpuBackend

FREQ_C = 4.2e15
opts = dict(a.split("=") for a in sys.argv[1:])
heat = int(opts.pop("heat", 1))
cont = int(opts.pop("cont", 1))
blocks = tuple(int(b) for b in opts.pop("blocks", "2,1,2").split(","))
tuning = dict(reemit_inline_below=64, tile_min_flights=0, tile_min_per_item=0)
tuning.update({k: int(v) for k, v in opts.items()})
ncell, npacket = 24, 30000
oracle.build()
sim = oracle.stromgren_simulation(ncell, diffuse=True)
rng = np.random.default_rng(3)
n = ncell ** 3
sim.x[0][:] = 10. ** rng.uniform(-5, -2, n)
Lc = 2.5 * S["luminosity"]
pos = [[0.2 * S["sides"][0], -0.1 * S["sides"][0], 0.05 * S["sides"][0]]]
sim.set_sources(pos, [1.], S["luminosity"])
if cont:
    sim.set_continuous_source(Lc, frequency=FREQ_C)
sim.model.reemit_type = oracle.REEMIT_PHYSICAL
dec = DomainDecomposition((ncell,) * 3, blocks)
shape = (ncell,) * 3
backends = []
for rank in range(dec.world):
    b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                         track_heating=bool(heat), export_capacity=4 * npacket)
    e = b.engine
    sigma = np.zeros(14); sigma[0] = S["sigma_H"]
    alpha = np.zeros(14); alpha[0] = S["alpha_H"]
    e.set_cross_sections_fixed(sigma)
    e.set_recombination_rates_fixed(alpha)
    e.set_sources(pos, [1.], S["luminosity"])
    e.set_spectrum_monochromatic(S["frequency"])
    if cont:
        e.set_continuous_spectrum_monochromatic(FREQ_C)
        e.set_continuous_source(E.CONTINUOUS_ISOTROPIC, Lc)
    e.set_reemission(1)
    e.set_tuning(**tuning)
    off, size = dec.block(rank)
    sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
    e.upload_cells(np.asarray(sim.number_density).reshape(shape)[sl].ravel(),
                   np.asarray(sim.temperature).reshape(shape)[sl].ravel(),
                   np.array([np.asarray(x).reshape(shape)[sl].ravel()
                             for x in sim.x]))
    backends.append(b)
group = EngineGroup([b.engine for b in backends])
for b in backends:
    b.reset_grid()
    b.shoot(21, 0, 0, npacket)
rounds = 0
while group.exchange_flights(21, 0):
    rounds += 1
tw, tc = 0., np.zeros(4)
for b in backends:
    b.synchronize()
    t, c, _ = b.get_counters()
    tw += t
    tc += np.asarray(c)
sim.reset(); sim.totweight = 0.; sim.typecount[:] = 0.
sim.shoot(21, 0, 0, npacket)
ok = abs(tw - sim.totweight) <= 1e-12 * sim.totweight and \
    np.allclose(tc, sim.typecount, rtol=1e-12, atol=0.)
print("%-70s %s rounds %d gpu %s oracle %s" % (
    " ".join(sys.argv[1:]) or "(test)", "OK  " if ok else "FAIL", rounds,
    tc, np.asarray(sim.typecount)), flush=True)
