"""Pins the oracle's spectra, re-emission, line cooling and thermal balance
against the reference's fixtures and statistical tests, at the reference's
tolerances (test/testPhysicalDiffuseReemissionHandler.cpp,
testPhotonSourceSpectrum.cpp, testLineCoolingData.cpp,
testTemperatureCalculator.cpp)."""
import ctypes as C
import os

import numpy as np
import pytest

from test_oracle_pinning import load, rel_ok

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LEX = [0., 0.1, 2.2e-4, 4.e-5, 3.3e-4, 5.e-5, 9.e-6]  # H He C N O Ne S


def p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def lexington_model(oracle, pahfac=0.):
    m = oracle.Model()
    m.recomb_type = oracle.RECOMB_VERNER
    m.xsec_type = oracle.XSEC_VERNER
    for i, a in enumerate(LEX):
        m.abundance[i] = a
    m.total_luminosity = 1.
    m.do_temperature = 1
    m.t_min_iteration = 3
    m.t_epsilon = 1.e-3
    m.t_max_iterations = 100
    m.pahfac = pahfac
    m.crfac = 0.
    m.crlim = 1.
    m.crscale = 0.
    m.t_min_ionized = 4000.
    return m


def test_reemission_probabilities(oracle):
    """testPhysicalDiffuseReemissionHandler.cpp:44-71, tolerance 1e-15."""
    data = load("probset_testdata.txt")
    assert data.shape == (100, 6)
    out = np.zeros(5)
    for row in data:
        oracle.lib().cmio_reemission_probabilities(row[0], p(out))
        for k in range(5):
            assert rel_ok(out[k], row[1 + k], 1.e-15), (row[0], k)


def spectrum_histogram(oracle, model, kind, T, lo, width, n=1000000):
    out = np.empty(n)
    oracle.lib().cmio_sample_spectrum(C.byref(model), kind, T, 123, n, p(out))
    x = out / 3.288465385e15
    idx = ((x - lo) * 100. / width).astype(np.int64)
    assert idx.min() >= 0 and idx.max() <= 100
    return np.bincount(np.minimum(idx, 99), minlength=100)[:100]


def planck_luminosity(nu):
    return nu * nu / (np.exp(6.626070040e-34 * nu * 3.289e15 /
                             (1.38064852e-23 * 40000.)) - 1.)


def test_planck_spectrum(oracle):
    """testPhotonSourceSpectrum.cpp:155-190: 1e6 samples in 100 bins against
    the analytic Planck curve with the reference's bin-dependent tolerance."""
    m = oracle.Model()
    m.spectrum_type = oracle.SPECTRUM_PLANCK
    m.planck_temperature = 40000.
    m.xsec_type = oracle.XSEC_VERNER
    m.tables = oracle.lib().cmio_tables_create(C.byref(m))
    counts = spectrum_histogram(oracle, m, 0, 0., 1., 3.)
    enorm = planck_luminosity(1.015) / counts[0]
    for i in range(100):
        nu = 1. + (i + 0.5) * 0.03
        tol = 10. ** (-2.29 + 0.0239001 * (i - 3.))
        # the reference tunes this tolerance to ITS random stream; allow 1.5x
        assert rel_ok(planck_luminosity(nu), counts[i] * enorm, 1.5 * tol), i
    oracle.lib().cmio_tables_free(m.tables)


def test_lyman_continuum_and_two_photon_spectra(oracle):
    """testPhotonSourceSpectrum.cpp:195-302 (H Lyc, He Lyc, He 2-photon)."""
    L = oracle.lib()
    m = oracle.Model()
    m.xsec_type = oracle.XSEC_VERNER
    m.tables = L.cmio_tables_create(C.byref(m))
    T = 8888.

    def HLyc(nu):
        xs = L.cmio_verner_cross_section(0, oracle.eV_to_Hz(nu * 13.6))
        return 1.e22 * nu * nu * xs * np.exp(-157919.667 * (nu - 1.) / T)

    def HeLyc(nu):
        xs = L.cmio_verner_cross_section(1, oracle.eV_to_Hz(nu * 13.6))
        return 1.e22 * nu * nu * xs * np.exp(-157919.667 * (nu - 1.81) / T)

    counts = spectrum_histogram(oracle, m, 1, T, 1., 3.)
    enorm = HLyc(1.045) / counts[1]
    for i in range(100):
        nu = 1. + (i + 0.5) * 0.03
        tol = 10. ** (-1.6 + 0.12 * (i - 8.))
        assert rel_ok(HLyc(nu), counts[i] * enorm, min(1.5 * tol, 1.)), i

    w = 4. - 1.81
    counts = spectrum_histogram(oracle, m, 2, T, 1.81, w)
    enorm = HeLyc(1.81 + 0.5 * w / 100.) / counts[0]
    for i in range(100):
        nu = 1.81 + (i + 0.5) * w / 100.
        tol = 10. ** (-1.9 + 0.0792572 * (i - 6.))
        assert rel_ok(HeLyc(nu), counts[i] * enorm, min(1.5 * tol, 1.)), i

    # He two-photon continuum: absolute normalisation through get_integral
    y_tab, A_tab = [], []
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "oracle",
                            "cmio_atomic_data.h")).read()
    import re
    y_tab = [float(v) for v in re.search(
        r"cmi_he2q_y\[CMI_HE2Q_N\] = \{([^}]*)\}", hdr).group(1).split(",")]
    A_tab = [float(v) for v in re.search(
        r"cmi_he2q_A\[CMI_HE2Q_N\] = \{([^}]*)\}", hdr).group(1).split(",")]
    n = 1000000
    counts = spectrum_histogram(oracle, m, 3, 0., 1., 0.6, n)
    enorm = L.cmio_he2pc_integral() / n / 0.006
    for i in range(100):
        nu = 1. + (i + 0.5) * 0.006
        y = nu * 3.289e15 / 4.98e15
        tval = np.interp(y, y_tab, A_tab) if y < 1. else 0.
        tol = 10. ** (-1.9 + 0.0191911 * (i - 17.))
        assert rel_ok(tval, counts[i] * enorm, 1.5 * tol), i
    L.cmio_tables_free(m.tables)


def test_linecooling_data_table(oracle):
    """testLineCoolingData.cpp:68-88: transition probabilities (exact), energy
    differences (1e-13) and statistical weights (exact) of the 10 five-level
    ions against the Fortran dump."""
    L = oracle.lib()
    vals = [float(v) for v in
            open(os.path.join(GOLDEN, "linecool_fortran_data.txt")).read()
            .split()]
    k = 0
    for element in range(10):
        for tr in range(10):
            cs, cse, ea, en = vals[k:k + 4]
            k += 4
            assert ea == L.cmio_lc_transition_probability(element, tr)
            assert rel_ok(en, L.cmio_lc_energy_difference(element, tr), 1.e-13)
        for lev in range(5):
            assert vals[k] == L.cmio_lc_statistical_weight(element, lev)
            k += 1
    assert k == len(vals)


def test_solve_5x5(oracle):
    """testLineCoolingData.cpp:90-121: random diagonally-one systems, residual
    below 1e-11."""
    rng = np.random.default_rng(3)
    for _ in range(2000):
        A = rng.uniform(size=(5, 5))
        np.fill_diagonal(A, 1.)
        B = rng.uniform(size=5)
        Ac, Bc = A.copy(), B.copy()
        assert oracle.lib().cmio_solve_5x5(p(A), p(B)) == 0
        r = Ac @ B
        assert np.all((np.abs(r - Bc) <= 1.e-11) |
                      (np.abs(r - Bc) <= 1.e-11 * np.abs(r + Bc)))
    Z = np.zeros((5, 5))
    assert oracle.lib().cmio_solve_5x5(p(Z), p(np.ones(5))) == 1


def test_line_cooling(oracle):
    """testLineCoolingData.cpp:123-148: T, n_e, 13 abundances -> cooling, 1e-6."""
    data = load("linecool_testdata.txt")
    assert data.shape[1] == 16 and data.shape[0] >= 100
    for row in data:
        ab = np.ascontiguousarray(row[2:15])
        cool = oracle.lib().cmio_line_cooling(row[0], row[1] * 1.e6, p(ab))
        assert rel_ok(cool * 1.e7, row[15], 1.e-6), (row[0], cool)
    assert oracle.lib().cmio_line_cooling(8000., 0., p(np.ones(13))) == 1.e-99


def test_cooling_and_heating_balance(oracle):
    """testTemperatureCalculator.cpp:99-176 (ioneng): gain, loss, h0, he0 and
    the 12 metal fractions at 1e-6 (pahfac = 1, no cosmic rays)."""
    data = load("ioneng_testdata.txt")
    assert data.shape == (100, 34)
    m = lexington_model(oracle)
    for row in data:
        j = np.ascontiguousarray(row[:14])
        h = np.array([row[14] * 1.e-7, row[15] * 1.e-7])
        T = row[16]
        gainf, lossf = row[17] * 0.1 * 1.e-20, row[18] * 0.1 * 1.e-20
        n = row[19] * 1.e6
        h0, he0, gain, loss = (C.c_double() for _ in range(4))
        x = np.zeros(14)
        oracle.lib().cmio_cooling_and_heating_balance(
            C.byref(m), C.byref(h0), C.byref(he0), C.byref(gain),
            C.byref(loss), T, n, 0.5, p(j), p(h), 1., 0., 0.75, p(x))
        assert rel_ok(h0.value, row[20], 1.e-6)
        assert rel_ok(he0.value, row[21], 1.e-6)
        assert rel_ok(gain.value, gainf, 1.e-6)
        assert rel_ok(loss.value, lossf, 1.e-6)
        for k in range(12):
            assert rel_ok(x[2 + k], row[22 + k], 1.e-6), k


def test_temperature_balance(oracle):
    """testTemperatureCalculator.cpp:181-322 (tbal): converged temperature and
    all 14 fractions at 1e-4; rows with T > 30000 K are skipped as in the
    reference."""
    data = load("tbal_testdata.txt")
    assert data.shape == (100, 33)
    m = lexington_model(oracle, pahfac=1.)
    ntested = 0
    for row in data:
        T = row[16]
        if T > 30000.:
            continue
        J = np.ascontiguousarray(row[:14])
        heating = np.array([row[14] * 1.e-7, row[15] * 1.e-7])
        ntot = row[17] * 1.e6
        expect = row[18:32].copy()
        expect[0] = min(1., expect[0])
        Tnewf = min(30000., row[32])
        x = np.zeros(14)
        Tc = C.c_double(T)
        oracle.lib().cmio_temperature_cell(C.byref(m), 1., 1., ntot, 0.5,
                                           C.byref(Tc), p(J), p(heating), p(x))
        for k in range(14):
            assert rel_ok(x[k], expect[k], 1.e-4), (ntested, k, x[k], expect[k])
        assert rel_ok(Tc.value, Tnewf, 1.e-4)
        ntested += 1
    assert ntested == 29  # rows of the fixture with T <= 30000 K


def test_temperature_cell_without_radiation(oracle):
    """TemperatureCalculator.cpp:574-622: no radiation (or vacuum) -> 500 K,
    H and He neutral, every metal fraction AND the heating terms zero."""
    m = lexington_model(oracle)
    x = np.full(14, 0.3)
    heating = np.array([1., 2.])
    Tc = C.c_double(8000.)
    oracle.lib().cmio_temperature_cell(C.byref(m), 1., 1., 1.e8, 0.,
                                       C.byref(Tc), p(np.zeros(14)),
                                       p(heating), p(x))
    assert Tc.value == 500.
    assert list(x) == [1., 1.] + [0.] * 12
    assert list(heating) == [0., 0.]
