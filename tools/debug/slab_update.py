"""GPU box: the cell update of an H-only grid, whole against in slabs."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from cmacionize_amd import GpuEngine, STROMGREN as S
from cmacionize_amd import engine as E
from test_gpu_domain import configure
n = 64
eng = GpuEngine((n,) * 3, S["anchor"], S["sides"], (0, 0, 0), device=0,
                track_heating=False)
configure(eng, "diffuse", n ** 3)
for loop in range(4):
    eng.reset_grid()
    eng.shoot(42, loop, 0, 1000000)
    tw, _, _ = eng.get_counters()
    eng.update_cells(loop, tw)
eng.reset_grid()
eng.shoot(13, 80, 0, 500000)
tw, _, _ = eng.get_counters()
state = [E.FIELD_TEMPERATURE] + [E.FIELD_IONIC_FRACTION + i for i in range(14)]
inputs = state + [E.FIELD_MEAN_INTENSITY]
before = {f: eng.download_field(f) for f in inputs}
eng.update_cells(9, tw)
ref = {f: eng.download_field(f) for f in state}
for f in inputs:
    eng.upload_field(f, before[f])
nc = n ** 3
cuts = [0, nc // 3 + 5, 2 * (nc // 3) - 7, nc]
for a, b in zip(cuts, cuts[1:]):
    eng.update_cells_range(9, tw, a, b - a)
for f in state:
    got = eng.download_field(f)
    bad = np.flatnonzero(~((got == ref[f]) | (np.isnan(got) & np.isnan(ref[f]))))
    print(f, "nan", np.isnan(got).sum(), np.isnan(ref[f]).sum(), "differ",
          bad.size, bad[:5], got[bad[:3]], ref[f][bad[:3]],
          before[f][bad[:3]] if f in before else None)
