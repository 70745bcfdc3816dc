"""The reference's known-answer fixtures for the atomic data, checked directly
on the DEVICE functions through cmi_gpu_physics_probe, and h0 through the
cell update itself (the other fixtures - ioneng, tbal - run on the device in
test_gpu_physics.py, hiilines and bjump in test_gpu_emissivity.py). Same data files, same tolerances as the reference's
tests; the device's pow/exp/log differ from libm by ulps only."""
import ctypes as C

import numpy as np
import pytest

from test_oracle_pinning import load, rel_ok
from test_oracle_physics import LEX

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from cmacionize_amd import GpuEngine
    eng = GpuEngine((4, 4, 4), (0., 0., 0.), (1., 1., 1.), device=0)
    eng.set_cross_sections_verner()
    eng.set_recombination_rates_verner()
    eng.set_abundances(LEX[1:])
    yield eng
    eng.close()


def test_verner_cross_sections_on_device(engine, oracle):
    """testVernerCrossSections.cpp:46-164, tolerance 1e-9"""
    data = load("verner_testdata.txt")
    nu = np.array([oracle.eV_to_Hz(row[0] * 13.6) for row in data])
    sigma = engine.physics_probe(0, nu)
    nonzero = 0
    for row, s in zip(data, sigma):
        for ion in range(14):
            got = s[ion] / (0.01 * 0.01) * 1.e18
            assert rel_ok(row[1 + ion], got, 1.e-9), (row[0], ion, got)
            nonzero += got > 0.
    assert nonzero > 500
    # and the oracle's numbers, everywhere (including just above thresholds)
    L = oracle.lib()
    ref = np.array([[L.cmio_verner_cross_section(ion, e) for ion in range(14)]
                    for e in nu])
    assert np.allclose(sigma, ref, rtol=1e-12, atol=0.)


def test_verner_recombination_rates_on_device(engine, oracle):
    """testVernerRecombinationRates.cpp:44-147; the reference asserts 1e-15
    against numbers printed by the same libm - the device's pow/exp are
    allowed a few ulps: 1e-13."""
    data = load("verner_rec_testdata.txt")
    alpha = engine.physics_probe(1, data[:, 0])
    for row, a in zip(data, alpha):
        for ion in range(14):
            got = a[ion] / (0.01 ** 3)
            assert rel_ok(got, row[1 + ion], 1.e-13), (row[0], ion, got)


def test_charge_transfer_rates_on_device(engine):
    """testChargeTransferRates.cpp:78-140, tolerance 1e-6"""
    ion = {(6, 4): 3, (7, 1): 4, (7, 2): 4, (7, 3): 5, (7, 4): 6, (8, 1): 7,
           (8, 2): 7, (8, 3): 8, (10, 3): 10, (16, 3): 11, (16, 4): 12,
           (16, 5): 13}
    data = load("KingdonFerland_testdata.txt")
    rates = engine.physics_probe(4, data[:, 2] * 1.e-4).reshape(-1, 14, 3)
    ntested = 0
    for (stage, atom, T, rec, ionr), r in zip(data, rates):
        key = (int(atom), int(stage))
        if key not in ion:
            continue
        if stage > 1:
            assert rel_ok(rec, r[ion[key], 0] * 1.e6, 1.e-6), (key, T)
            ntested += 1
        if key in ((7, 1), (8, 1)):
            assert rel_ok(ionr, r[ion[key], 1] * 1.e6, 1.e-6), (key, T)
            ntested += 1
    assert ntested > 1000


def test_line_cooling_on_device(engine):
    """testLineCoolingData.cpp:123-148: T, n_e, 13 abundances -> cooling at
    1e-6; and the n_e = 0 convention (:1772-1775)"""
    data = load("linecool_testdata.txt")
    rows = np.column_stack([data[:, 0], data[:, 1] * 1.e6, data[:, 2:15]])
    cool = engine.physics_probe(2, rows)[:, 0]
    for row, c in zip(data, cool):
        assert rel_ok(c * 1.e7, row[15], 1.e-6), (row[0], c)
    zero = engine.physics_probe(2, [[8000., 0.] + [1.] * 13])
    assert zero[0, 0] == 1.e-99


def test_line_cooling_over_the_temperature_range(engine, oracle):
    """the device's line cooling (Boltzmann factors as quotients of level
    factors, T^a6 from its series) against the oracle's plain restatement of
    LineCoolingData::get_cooling from 10 K - where the level factors
    underflow and the transitions' own exponentials take over - to the
    solve's upper limit of 1.1e10 K, at densities from 1 to 1e6 cm^-3"""
    rng = np.random.default_rng(5)
    T = np.concatenate([[10., 30., 60., 100., 130., 200., 450., 500.],
                        np.logspace(2.7, 10.04, 40)])
    rows, expect = [], []
    for t in T:
        for ne in (1.e6, 1.e9, 1.e12):
            ab = rng.uniform(1.e-6, 1.e-3, 13)
            rows.append([t, ne] + list(ab))
            expect.append(oracle.lib().cmio_line_cooling(
                t, ne, ab.ctypes.data_as(C.POINTER(C.c_double))))
    cool = engine.physics_probe(2, rows)[:, 0]
    expect = np.array(expect)
    # (the fits are extrapolated far beyond their range: no sign is asserted)
    assert np.isfinite(cool).all() and np.isfinite(expect).all()
    assert np.allclose(cool, expect, rtol=1e-11, atol=0.), \
        np.abs(cool / expect - 1.).max()


def test_reemission_probabilities_on_device(engine):
    """testPhysicalDiffuseReemissionHandler.cpp:44-71; reference tolerance
    1e-15 against its own libm, a few ulps of the device's pow: 1e-14"""
    data = load("probset_testdata.txt")
    p = engine.physics_probe(3, data[:, 0])
    for row, got in zip(data, p):
        for k in range(5):
            assert rel_ok(got[k], row[1 + k], 1.e-14), (row[0], k)


def test_ionization_state_calculator_on_device(oracle):
    """h0_testdata.txt (testIonizationStateCalculator.cpp:68-204) through the
    engine's own cell update (`ionization_kernel<FULL>`): the file's 100
    states are the 100 cells of a 5 x 5 x 4 grid of unit cells, their 14 mean
    intensities uploaded as the accumulators; one source photon per second and
    a total weight of 1 make the normalisation factor
    luminosity / (weight x cell volume) the 1 of the reference test.
    Tolerance 1e-9, the reference's."""
    from cmacionize_amd import GpuEngine
    from cmacionize_amd import engine as E
    data = load("h0_testdata.txt")
    assert data.shape == (100, 30)
    eng = GpuEngine((5, 5, 4), (0., 0., 0.), (5., 5., 4.), device=0,
                    track_heating=True)
    eng.set_sources([[2.5, 2.5, 2.]], [1.], 1.)
    eng.set_spectrum_planck(40000.)
    eng.set_cross_sections_verner()
    eng.set_recombination_rates_verner()
    eng.set_abundances([0.1, 0., 0., 0., 0., 0.])
    eng.set_temperature_params(do_temperature_calculation=0)
    x0 = np.full((14, 100), 0.5)
    eng.upload_cells(data[:, 15] * 1.e6, data[:, 14], x0)
    eng.reset_grid()
    for ion in range(14):
        eng.upload_field(E.FIELD_MEAN_INTENSITY + ion, data[:, ion])
    eng.update_cells(0, 1.)
    worst = 0.
    for ion in range(14):
        got = eng.download_field(E.FIELD_IONIC_FRACTION + ion)
        for c in range(100):
            assert rel_ok(got[c], data[c, 16 + ion], 1.e-9), \
                (c, ion, got[c], data[c, 16 + ion])
        nz = data[:, 16 + ion] != 0.
        worst = max(worst, np.abs(got[nz] / data[nz, 16 + ion] - 1.).max())
    assert worst < 1.e-9
    # the oracle from the same inputs, much closer than the fixture's digits
    L = oracle.lib()
    from test_oracle_pinning import verner_model
    import ctypes as C
    m = verner_model(oracle, 0.1)
    for c in (0, 17, 99):
        J = np.ascontiguousarray(data[c, :14])
        heating, x = np.zeros(2), np.zeros(14)
        L.cmio_ionization_state_cell(C.byref(m), 1., 1., data[c, 15] * 1.e6,
                                     data[c, 14], J.ctypes.data_as(oracle.dp),
                                     heating.ctypes.data_as(oracle.dp),
                                     x.ctypes.data_as(oracle.dp))
        got = np.array([eng.download_field(E.FIELD_IONIC_FRACTION + ion)[c]
                        for ion in range(14)])
        assert np.allclose(got, x, rtol=1e-11, atol=0.)
    eng.close()
