"""WeightedSpectrumTracker's pieces in the oracle (oracle/cmio_transport.c) and
in the engine's host/device function (cmi_gpu_projected_areas), pinned by the
known answers of test/testWeightedSpectrumTracker.cpp:40-98 and by the
definitions of src/LinearFrequencyBins.hpp / src/LevelFrequencyBins.hpp."""
import numpy as np
import pytest

import oracle_lib as O


def directions_of_the_reference_test():
    axes = [[1., 0., 0.], [0., 1., 0.], [0., 0., 1.],
            [-1., 0., 0.], [0., -1., 0.], [0., 0., -1.]]
    face = np.array([1., 1., 0.])
    face /= np.sqrt((face * face).sum())
    body = np.array([1., 1., 1.])
    body /= np.sqrt((body * body).sum())
    return axes, face, body


def check_known_answers(area):
    axes, face, body = directions_of_the_reference_test()
    # testWeightedSpectrumTracker.cpp:40-69: exactly 1 along the axes
    for d in axes:
        assert area(d) == 1.
    # :72-77: exactly sqrt(2) along a face diagonal
    assert area(face) == np.sqrt(2.)
    # :80-86: sqrt(3) to 1e-16 (relative) along the body diagonal
    assert abs(area(body) - np.sqrt(3.)) <= 1e-16 * (area(body) + np.sqrt(3.))
    # :89-111: positive and not NaN for random directions
    rng = np.random.default_rng(42)
    cost = 2. * rng.random(20000) - 1.
    phi = 2. * np.pi * rng.random(20000)
    sint = np.sqrt(np.maximum(1. - cost * cost, 0.))
    d = np.stack([sint * np.cos(phi), sint * np.sin(phi), cost], axis=1)
    d /= np.sqrt((d * d).sum(axis=1))[:, None]
    a = np.array([area(x) for x in d])
    assert np.all(a > 0.) and not np.isnan(a).any()
    # the shadow of a unit cube: |dx| + |dy| + |dz|
    assert np.allclose(a, np.abs(d).sum(axis=1), rtol=1e-12)
    return d, a


def test_projected_area_known_answers_oracle():
    check_known_answers(O.projected_area)


def test_projected_area_known_answers_engine_function():
    from cmacionize_amd import engine as E
    d, a = check_known_answers(lambda x: E.projected_areas(x)[0])
    # the engine's function and the oracle's restatement agree to the bit
    # where the reference guards against rounding (the y and z pairs) and
    # elsewhere
    want = np.array([O.projected_area(x) for x in d])
    assert np.array_equal(E.projected_areas(d), want)


def test_frequency_bins():
    ev = 1.6021766208e-19 / 6.626070040e-34
    lo, hi, n = 13.6 * ev, 54.4 * ev, 100
    width = (hi - lo) / n
    # src/LinearFrequencyBins.hpp:115-125
    assert O.frequency_bin("Linear", n, lo, hi, 0.5 * lo) == 0
    assert O.frequency_bin("Linear", n, lo, hi, lo) == 0
    assert O.frequency_bin("Linear", n, lo, hi, hi) == n - 1
    assert O.frequency_bin("Linear", n, lo, hi, 10. * hi) == n - 1
    for i in (0, 1, 17, 98, 99):
        assert O.frequency_bin("Linear", n, lo, hi, lo + (i + 0.5) * width) == i
    # src/LevelFrequencyBins.hpp:52-86 over src/ElementData.hpp:39-105: H0
    # 3.288e15 < O0 3.293e15 < N0 3.514e15 < Ne0 < S+ < C+ < He0 < N+ < S++ <
    # O+ < Ne+ < S3+ < N++ < C++ < 4 x H0
    energies = sorted([3.28810279e+15, 5.94523574e+15, 5.89588678e+15,
                       1.15792700e+16, 3.51435505e+15, 7.15759434e+15,
                       1.14732262e+16, 3.29284691e+15, 8.49136314e+15,
                       5.21432028e+15, 9.90492110e+15, 5.64310422e+15,
                       8.41222200e+15, 1.14182796e+16])
    for i, e in enumerate(energies):
        assert O.frequency_bin("Level", 14, 0., 0., e * 1.0001) == i
    # Utilities::locate: below the first edge the first bin, above the upper
    # edge the last; an edge itself belongs to the bin below it
    assert O.frequency_bin("Level", 14, 0., 0., 1.e15) == 0
    assert O.frequency_bin("Level", 14, 0., 0., 1.e17) == 13
    assert O.frequency_bin("Level", 14, 0., 0., energies[5]) == 4
    assert O.frequency_bin("Level", 14, 0., 0., 4 * 3.28810279e+15) == 13
