"""Every branch of PhysicalDiffuseReemissionHandler::reemit
(src/PhysicalDiffuseReemissionHandler.cpp:219-370), walked on the oracle with
a scripted sequence of uniform random numbers.

The reference's own test (test/testPhysicalDiffuseReemissionHandler.cpp:44-71)
only pins the five probabilities (probset_testdata.txt); the branch logic that
turns them and the random numbers into a channel, a photon type and a new
frequency had no vector behind it. Here the expected outcome of each branch is
derived in this file, from the reference's text: which uniform is compared
with which probability, which spectrum is sampled with the next uniform, which
type results, and how many uniforms the call consumes.
"""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

NFREQ, NTEMP = 1000, 100


class Tables(C.Structure):
    """struct cmio_tables (oracle/cmio.h)"""
    _fields_ = [("planck_logfreq", C.c_double * NFREQ),
                ("planck_cdf", C.c_double * NFREQ),
                ("planck_logcdf", C.c_double * NFREQ),
                ("lyc_T", C.c_double * NTEMP),
                ("lyc_freq", (C.c_double * NFREQ) * 2),
                ("lyc_cdf", ((C.c_double * NFREQ) * NTEMP) * 2),
                ("he2pc_freq", C.c_double * NFREQ),
                ("he2pc_cdf", C.c_double * NFREQ)]


def locate(x, arr):
    """Utilities::locate, src/Utilities.hpp:726-742"""
    lo = max(int(np.searchsorted(arr, x, side="left")) - 1, 0)
    return min(lo, len(arr) - 2)


@pytest.fixture(scope="module")
def handler():
    sim = O.lexington_simulation(4)
    sim.build_tables()
    L = O.lib()
    L.cmio_reemit_scripted.restype = C.c_double
    L.cmio_reemit_scripted.argtypes = [C.POINTER(O.Model)] + [C.c_double] * 6 \
        + [O.dp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
    t = C.cast(sim.model.tables, C.POINTER(Tables)).contents
    tables = dict(lyc_T=np.array(t.lyc_T),
                  lyc_freq=np.array(t.lyc_freq),
                  lyc_cdf=np.array(t.lyc_cdf),
                  he2pc_freq=np.array(t.he2pc_freq),
                  he2pc_cdf=np.array(t.he2pc_cdf))
    return sim, tables


# state of the absorbing cell and of the packet
T = 8300.
XH, XHE, AHE = 2.e-3, 5.e-3, 0.1
SIGMA_H, SIGMA_HE = 2.e-22, 6.e-22


def call(sim, uniforms):
    u = np.array(list(uniforms) + [np.nan] * 4)  # reading past = NaN result
    typ = C.c_int32(-1)
    draws = C.c_uint32(0)
    nu = O.lib().cmio_reemit_scripted(
        C.byref(sim.model), SIGMA_H, SIGMA_HE, AHE, T, XH, XHE, O._ptr(u),
        C.byref(typ), C.byref(draws))
    return nu, typ.value, draws.value


def lyc(tables, which, x):
    """Hydrogen/HeliumLymanContinuumSpectrum::get_random_frequency,
    src/HydrogenLymanContinuumSpectrum.cpp:136-153"""
    iT = locate(T, tables["lyc_T"])
    i1 = locate(x, tables["lyc_cdf"][which][iT])
    i2 = locate(x, tables["lyc_cdf"][which][iT + 1])
    nu = tables["lyc_freq"][which]
    return nu[i1] + (T - tables["lyc_T"][iT]) * (nu[i2] - nu[i1]) / \
        (tables["lyc_T"][iT + 1] - tables["lyc_T"][iT])


def he2pc(tables, x):
    """HeliumTwoPhotonContinuumSpectrum::get_random_frequency,
    src/HeliumTwoPhotonContinuumSpectrum.cpp:167-180"""
    i = locate(x, tables["he2pc_cdf"])
    f, c = tables["he2pc_freq"], tables["he2pc_cdf"]
    return f[i] + (f[i + 1] - f[i]) * (x - c[i]) / (c[i + 1] - c[i])


def probabilities():
    p = np.zeros(5)
    O.lib().cmio_reemission_probabilities(T, O._ptr(p))
    return p


def test_probabilities_are_the_reference_formulae():
    """PhysicalDiffuseReemissionHandler.hpp:66-105, restated in numpy"""
    p = probabilities()
    T4 = T * 1.e-4
    a1H = 1.58e-13 * T4 ** -0.53
    aA = 4.18e-13 * T4 ** -0.7
    a = np.array([1.54e-13 * T4 ** -0.486, 2.1e-13 * T4 ** -0.381,
                  2.06e-14 * T4 ** -0.451, 4.17e-14 * T4 ** -0.695])
    assert p[0] == pytest.approx(a1H / aA, rel=1e-14)
    assert np.allclose(p[1:], np.cumsum(a) / a.sum(), rtol=1e-14)
    assert p[4] == pytest.approx(1., rel=1e-14)


def test_every_branch_of_the_physical_handler(handler):
    sim, tables = handler
    p = probabilities()
    pHabs = XH * SIGMA_H / (XH * SIGMA_H + XHE * AHE * SIGMA_HE)
    pHots = np.sqrt(T) * XH / (np.sqrt(T) * XH + 77. * XHE)
    assert 0.05 < pHabs < 0.95 and 0.05 < pHots < 0.95
    below = lambda v: np.nextafter(v, 0.)  # noqa: E731
    above = lambda v: np.nextafter(v, 2.)  # noqa: E731
    H, HE, ABS = O_TYPE = (1, 2, 3)  # PhotonType.hpp:36-50
    del O_TYPE
    s = 0.3137  # the uniform handed to a spectrum
    cases = [
        # (uniforms, expected frequency, type, draws)
        # -- absorbed by hydrogen (x <= pHabs, inclusive: `<=` at :242)
        ("H -> Lyc", [pHabs, p[0], s], lyc(tables, 0, s), H, 3),
        ("H -> Lyc, low", [0.01, 0.01, 0.9], lyc(tables, 0, 0.9), H, 3),
        ("H -> lost", [pHabs, above(p[0])], 0., ABS, 2),
        # -- absorbed by helium
        ("He -> He Lyc", [above(pHabs), p[1], s], lyc(tables, 1, s), HE, 3),
        ("He -> 19.8 eV", [above(pHabs), above(p[1])], 4.788e15, HE, 2),
        ("He -> 19.8 eV, top", [0.99, p[2]], 4.788e15, HE, 2),
        ("He -> 2-photon, ionizing", [0.99, above(p[2]), below(0.56), s],
         he2pc(tables, s), HE, 4),
        ("He -> 2-photon, lost (x == 0.56: `<` at :289)",
         [0.99, p[3], 0.56], 0., ABS, 3),
        ("He Lya -> on the spot -> H Lyc",
         [0.99, above(p[3]), below(pHots), p[0], s], lyc(tables, 0, s), H, 5),
        ("He Lya -> on the spot -> lost",
         [0.99, above(p[3]), below(pHots), above(p[0])], 0., ABS, 4),
        ("He Lya -> 2-photon, ionizing (x == pHots: `<` at :325)",
         [0.99, p[4], pHots, 0.1, s], he2pc(tables, s), HE, 5),
        ("He Lya -> 2-photon, lost", [0.99, p[4], 0.999, 0.7], 0., ABS, 4),
    ]
    if above(p[4]) < 1.:
        # the reference's "should never be called" branch, :363-367
        cases.append(("beyond the last channel", [0.99, above(p[4])], 0.,
                      ABS, 2))
    for name, uniforms, nu_expected, type_expected, ndraws in cases:
        nu, typ, draws = call(sim, uniforms)
        assert typ == type_expected, name
        assert draws == ndraws, name
        assert nu == pytest.approx(nu_expected, rel=1e-15, abs=0.), name
        if nu_expected != 0.:
            assert nu > 3.28e15, name  # re-emitted photons ionize hydrogen


def test_sampled_frequencies_are_in_range(handler):
    sim, tables = handler
    p = probabilities()
    nuH = 3.289e15
    for s in np.linspace(1e-6, 1. - 1e-6, 41):
        nu, typ, _ = call(sim, [0.001, 0.001, s])
        assert typ == 1 and nuH <= nu <= 4. * nuH
        nu, typ, _ = call(sim, [0.999, 0.5 * p[1], s])
        assert typ == 2 and 1.81 * 3.288465385e15 <= nu <= 4. * 3.288465385e15
        nu, typ, _ = call(sim, [0.999, 0.5 * (p[2] + p[3]), 0.1, s])
        assert typ == 2 and 3.288465385e15 <= nu <= 1.6 * 3.288465385e15


def test_fixed_value_handler():
    """FixedValueDiffuseReemissionHandler::reemit,
    src/FixedValueDiffuseReemissionHandler.hpp:73-86"""
    sim = O.stromgren_simulation(4)
    sim.model.reemit_type = O.REEMIT_FIXED
    sim.model.reemit_fixed_probability = 0.364
    sim.model.reemit_fixed_frequency = 3.5e15
    L = O.lib()
    L.cmio_reemit_scripted.restype = C.c_double
    L.cmio_reemit_scripted.argtypes = [C.POINTER(O.Model)] + [C.c_double] * 6 \
        + [O.dp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
    assert call(sim, [np.nextafter(0.364, 0.)]) == (3.5e15, 1, 1)
    assert call(sim, [0.364]) == (0., 3, 1)
