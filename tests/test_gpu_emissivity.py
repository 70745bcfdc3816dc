"""GPU parity of the emissivity post-processing (cmi_gpu_compute_emissivities,
EmissivityCalculator of the reference) - through the C ABI, against the
oracle on random cell states, against the reference's Balmer jump fixture, and
on a converged benchmark grid."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_physics import lexington_engine
from test_oracle_pinning import load, rel_ok

pytestmark = pytest.mark.gpu

# device and host libm differ in the last bits of exp / log / pow, and the
# level populations come out of a 5x5 elimination: relative tolerance
RTOL = 1.e-10


def random_state(ncell, seed):
    rng = np.random.default_rng(seed)
    n = ncell ** 3
    density = 10. ** rng.uniform(6., 10., n)
    temperature = 10. ** rng.uniform(3.3, 4.5, n)
    x = np.empty((14, n))
    x[0] = 10. ** rng.uniform(-6., -0.5, n)   # some cells above the 0.2 cut
    x[1] = rng.uniform(0., 1., n)
    # the ions of an element share one unit of abundance
    for group in ((2, 3), (4, 5, 6), (7, 8), (9, 10), (11, 12, 13)):
        parts = rng.dirichlet(np.ones(len(group) + 1), n).T
        for k, ion in enumerate(group):
            x[ion] = parts[k]
    temperature[:5] = (2000., 2999., 3000., 3001., 25000.)
    x[0, 5:8] = (0.19, 0.2, 0.21)
    return density, temperature, x


def test_emissivities_match_oracle_cell_by_cell(oracle):
    from cmacionize_amd import engine as E
    ncell = 12
    sim = oracle.lexington_simulation(ncell)
    density, temperature, x = random_state(ncell, 7)
    eng = lexington_engine(ncell)
    eng.upload_cells(density, temperature, x)
    got = eng.compute_emissivities()
    assert list(got) == E.EMISSION_LINES == oracle.EMISSION_LINES
    n = ncell ** 3
    ref = np.array([oracle.emissivities(sim.model, density[c], temperature[c],
                                        x[:, c]) for c in range(n)]).T
    dark = ~((x[0] < 0.2) & (temperature > 3000.))
    assert dark.sum() > 10 and (~dark).sum() > 1000
    for k, name in enumerate(E.EMISSION_LINES):
        g = got[name]
        assert not g[dark].any(), name
        err = np.abs(g - ref[k]) / np.maximum(np.abs(ref[k]), 1e-300)
        assert err[~dark].max() < RTOL, (name, err.max())
        assert (g[~dark] > 0.).all(), name
    eng.close()


def test_selected_lines_and_cell_ranges(oracle):
    ncell = 8
    density, temperature, x = random_state(ncell, 11)
    eng = lexington_engine(ncell)
    eng.upload_cells(density, temperature, x)
    everything = eng.compute_emissivities()
    some = eng.compute_emissivities(["OIII_5007", "HAlpha", "WFC2_F675W"],
                                    first_cell=100, ncell=77)
    assert list(some) == ["OIII_5007", "HAlpha", "WFC2_F675W"]
    for name, values in some.items():
        assert values.shape == (77,)
        assert np.array_equal(values, everything[name][100:177])
    assert eng.compute_emissivities(["HBeta"], 5, 0)["HBeta"].shape == (0,)
    eng.close()


def test_balmer_jump_fixture_through_the_engine(oracle):
    """bjump_testdata.txt (test/testEmissivityCalculator.cpp:50-86): with
    helium fully neutral the Balmer jump emissivities are n_e n_H+ times the
    hydrogen coefficients; with hydrogen's share subtracted, helium's."""
    data = load("bjump_testdata.txt")
    ncell = 8
    n = ncell ** 3
    assert len(data) <= n
    T = np.full(n, 8000.)
    T[:len(data)] = data[:, 0]
    density = np.full(n, 1.e8)
    x = np.zeros((14, n))
    x[1] = 1.
    eng = lexington_engine(ncell)
    eng.upload_cells(density, T, x)
    h = eng.compute_emissivities(["BALMER_JUMP_HIGH", "BALMER_JUMP_LOW"])
    x[1] = 0.
    eng.upload_cells(density, T, x)
    hhe = eng.compute_emissivities(["BALMER_JUMP_HIGH", "BALMER_JUMP_LOW"])
    AHe = 0.1
    checked = 0
    for c, row in enumerate(data):
        if not row[0] > 3000.:
            assert h["BALMER_JUMP_HIGH"][c] == 0.
            continue
        unit = 1.e-20 * 1.e-13
        for name, ih, ihe in (("BALMER_JUMP_HIGH", 1, 3),
                              ("BALMER_JUMP_LOW", 2, 4)):
            assert rel_ok(h[name][c] / 1.e16, row[ih] * unit, 1.e-3), (c, name)
            ne = 1.e8 * (1. + AHe)
            he = (hhe[name][c] / ne - 1.e8 * h[name][c] / 1.e16) / (1.e8 * AHe)
            assert rel_ok(he, row[ihe] * unit, 1.e-3), (c, name)
        checked += 1
    assert checked > 50
    eng.close()


def test_emissivities_of_a_converged_benchmark_grid(oracle):
    """lexingtonHII40 run to a warm state on the device: the engine's own
    grid, post-processed on the device and by the oracle from the downloaded
    fields."""
    from cmacionize_amd import engine as E
    ncell, npacket = 16, 40000
    sim = oracle.lexington_simulation(ncell)
    eng = lexington_engine(ncell, sim)
    for loop in range(6):
        eng.reset_grid()
        eng.shoot(42, loop, 0, npacket)
        tw, _, _ = eng.get_counters()
        eng.update_cells(loop, tw)
    T = eng.download_field(E.FIELD_TEMPERATURE)
    dens = eng.download_field(E.FIELD_NUMBER_DENSITY)
    x = np.array([eng.download_field(E.FIELD_IONIC_FRACTION + i)
                  for i in range(14)])
    got = eng.compute_emissivities()
    lit = (x[0] < 0.2) & (T > 3000.)
    assert lit.sum() > 100
    for c in np.flatnonzero(lit)[::7]:
        ref = oracle.emissivities(sim.model, dens[c], T[c], x[:, c])
        for k, name in enumerate(E.EMISSION_LINES):
            assert rel_ok(got[name][c], ref[k], RTOL), (c, name)
    # the classic diagnostic of the benchmark: [OIII] 5007 outshines H beta
    # inside the nebula
    assert got["OIII_5007"][lit].sum() > got["HBeta"][lit].sum()
    eng.close()


def test_bad_arguments_are_refused():
    from cmacionize_amd import engine as E
    eng = lexington_engine(4)
    lib = E.load_library()
    out = (C.c_double * 64)()
    line = (C.c_int32 * 1)(0)
    assert lib.cmi_gpu_compute_emissivities(eng._h, 1, line, 0, 64, out) != 0
    assert b"cell data" in lib.cmi_gpu_last_error()
    eng.upload_cells(np.full(64, 1e8), np.full(64, 8000.), np.zeros((14, 64)))
    line[0] = 42
    assert lib.cmi_gpu_compute_emissivities(eng._h, 1, line, 0, 64, out) != 0
    assert b"no emission line 42" in lib.cmi_gpu_last_error()
    line[0] = 0
    assert lib.cmi_gpu_compute_emissivities(eng._h, 1, line, 60, 5, out) != 0
    assert lib.cmi_gpu_compute_emissivities(eng._h, 0, line, 0, 64, out) != 0
    assert lib.cmi_gpu_compute_emissivities(eng._h, 1, line, 0, 64, out) == 0
    eng.close()


def test_hiilines_fixture_on_device():
    """hiilines_testdata.txt (test/testEmissivityCalculator.cpp:88-260): the
    100 HII-region cells of Kenny Wood's code as the cells of a 5 x 5 x 4
    grid, 28 line sums at the reference's tolerances (1e-6; Balmer jump
    1e-3) - `emissivity_kernel` end to end against reference numbers."""
    from cmacionize_amd import GpuEngine
    from test_oracle_emissivity import check_hiilines
    data = load("hiilines_testdata.txt")
    assert data.shape == (100, 46)
    eng = GpuEngine((5, 5, 4), (0., 0., 0.), (5., 5., 4.), device=0)
    eng.set_cross_sections_verner()
    eng.set_recombination_rates_verner()
    eng.set_abundances([0.1, 2.2e-4, 4.e-5, 3.3e-4, 5.e-5, 9.e-6])
    eng.upload_cells(data[:, 0] * 1.e6, data[:, 1],
                     np.ascontiguousarray(data[:, 2:16].T))
    got = eng.compute_emissivities()
    for c in range(100):
        check_hiilines(data[c], {name: got[name][c] for name in got})
    assert (got["HAlpha"] > 0.).sum() > 50
    eng.close()
