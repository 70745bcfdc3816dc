"""GPU tests at BASELINE.json's full grid size (256^3), where the oracle is too
slow to be the checker: size-independent properties of the transport step -
packet conservation, additivity over packet ranges, invariance under every
reordering the engine applies (sorting, aggregation, launch splitting, block
decomposition) and the symmetry of the benchmark itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NCELL = 256


@pytest.fixture(scope="module")
def converged():
    """stromgren.param on 256^3, brought near its converged state."""
    from test_gpu_transport import make_engine
    eng = make_engine(NCELL, track_heating=False)
    for loop in range(6):
        eng.reset_grid()
        eng.shoot(42, loop, 0, 4000000)
        tw, _, _ = eng.get_counters()
        eng.update_cells(loop, tw)
    yield eng
    eng.close()


def test_fullsize_conservation_additivity_symmetry(converged):
    from cmacionize_amd import engine as E
    eng = converged
    n = 10000000
    eng.reset_grid()
    eng.shoot(7, 50, 0, n)
    tw, tc, ns = eng.get_counters()
    J = eng.download_field(E.FIELD_MEAN_INTENSITY)
    # every packet ends exactly once, as an escaped primary or absorbed
    assert tw == n and tc.sum() == n and tc[1] == tc[2] == 0
    assert 100. < ns / n < 200.  # ~1.5 x the Stromgren radius in cells
    # additivity over packet ranges (disjoint Philox counters): three uneven
    # parts, accumulated in place, give the same integrals
    eng.reset_grid()
    for first, count in ((0, 1234567), (1234567, 5000000),
                         (6234567, n - 6234567)):
        eng.shoot(7, 50, first, count)
    tw2, tc2, ns2 = eng.get_counters()
    J2 = eng.download_field(E.FIELD_MEAN_INTENSITY)
    assert tw2 == tw and np.array_equal(tc2, tc) and ns2 == ns
    assert np.allclose(J2, J, rtol=1e-11, atol=1e-13 * J.max())
    # the source sits at the centre of a uniform box: the 8 octants receive
    # the same integral within Monte Carlo noise (<< 1 % at 1e7 packets)
    Jc = J.reshape((NCELL,) * 3)
    h = NCELL // 2
    octants = np.array([Jc[i:i + h, j:j + h, k:k + h].sum()
                        for i in (0, h) for j in (0, h) for k in (0, h)])
    assert np.abs(octants / octants.mean() - 1.).max() < 5.e-3


def test_fullsize_reordering_invariance(converged):
    """Sorting, tau classes, aggregation, refill policy, launch splitting and
    the exact marcher only reorder the work or change roundings."""
    from cmacionize_amd import engine as E
    eng = converged
    n = 3000000
    results = []
    default = dict(sort_packets=1, aggregate=2, refill_threshold=64, chunk=64,
                   sort_tau_bits=-1, max_packets_per_launch=1 << 27,
                   exact_dda=0, pad_march=1)
    for kw in (dict(),
               # the march on the plain records instead of the padded ones
               dict(pad_march=0),
               dict(pad_march=0, sort_tau_bits=3, chunk=128),
               dict(sort_packets=0, aggregate=0),
               dict(sort_tau_bits=3, max_packets_per_launch=700001),
               dict(aggregate=1, refill_threshold=16, chunk=256),
               dict(exact_dda=1)):
        tuning = dict(default)
        tuning.update(kw)
        eng.set_tuning(**tuning)
        eng.reset_grid()
        eng.shoot(11, 60, 5, n)
        results.append((eng.get_counters(),
                        eng.download_field(E.FIELD_MEAN_INTENSITY)))
    eng.set_tuning(**default)
    (tw0, tc0, ns0), J0 = results[0]
    for (tw, tc, ns), J in results[1:-1]:
        assert tw == tw0 == n and np.array_equal(tc, tc0) and ns == ns0
        assert np.allclose(J, J0, rtol=1e-10, atol=1e-13 * J0.max())
    # the exact marcher rounds differently (accumulated 1e-15 per step): a
    # packet in ~1e6 ends one cell earlier or later
    (tw, tc, ns), J = results[-1]
    assert tw == n and np.abs(tc - tc0).max() <= 3
    assert abs(ns - ns0) <= 1e-6 * ns0
    assert abs(J.sum() - J0.sum()) <= 1e-9 * J0.sum()


@pytest.mark.parametrize("heating", [False, True])
def test_large_grid_reordering_invariance(heating):
    """beyond 2^25 cells the padded march runs in blocks of 1024 threads with
    a table of 4096 slots (shoot_kernel<..., PAD, ..., BIG>): 336^3, the same
    packets through that kernel, through the march on the plain records and
    one packet per lane with single atomics - the same tallies."""
    from cmacionize_amd import engine as E
    from test_gpu_transport import make_engine
    ncell, n = 336, 3000000
    assert ncell ** 3 > 1 << 25
    eng = make_engine(ncell, track_heating=heating)
    for loop in range(4):
        eng.reset_grid()
        eng.shoot(42, loop, 0, 3000000)
        tw, _, _ = eng.get_counters()
        eng.update_cells(loop, tw)
    fields = [E.FIELD_MEAN_INTENSITY] + ([E.FIELD_HEATING] if heating else [])
    results = []
    default = dict(sort_packets=1, aggregate=2, pad_march=1)
    for kw in (dict(), dict(pad_march=0), dict(sort_packets=0, aggregate=0)):
        tuning = dict(default)
        tuning.update(kw)
        eng.set_tuning(**tuning)
        eng.reset_grid()
        eng.shoot(11, 60, 5, n)
        results.append((eng.get_counters(),
                        [eng.download_field(f) for f in fields]))
    eng.close()
    (tw0, tc0, ns0), J0 = results[0]
    assert tw0 == n and 100. < ns0 / n < 300.
    for (tw, tc, ns), J in results[1:]:
        assert tw == tw0 and np.array_equal(tc, tc0) and ns == ns0
        for a, b in zip(J, J0):
            assert np.allclose(a, b, rtol=1e-10, atol=1e-13 * np.abs(b).max())


def test_fullsize_decomposition_invariance(converged):
    """256^3 as 2 x 2 x 2 blocks of 128^3 (config 5's shape at half the
    linear size) against the undivided grid, same state, same packets."""
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.simulation import (DomainDecomposition,
                                           DomainGpuBackend,
                                           LocalDomainDriver)
    from test_gpu_domain import assemble, configure
    eng = converged
    n = 4000000
    xH = eng.download_field(E.FIELD_IONIC_FRACTION).reshape((NCELL,) * 3)
    dec = DomainDecomposition((NCELL,) * 3, (2, 2, 2))
    backends = []
    for rank in range(dec.world):
        b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                             export_capacity=n)
        off, size = dec.block(rank)
        nloc = int(np.prod(size))
        configure(b.engine, "stromgren", nloc)
        sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
        x = np.zeros((14, nloc))
        x[0] = xH[sl].ravel()
        x[1] = 1.e-6
        b.engine.upload_cells(np.full(nloc, S["density"]),
                              np.full(nloc, S["temperature"]), x)
        backends.append(b)
    driver = LocalDomainDriver(backends, dec)
    driver.iteration(70, n, 3, update=False)
    eng.reset_grid()
    eng.shoot(3, 70, 0, n)
    tw, tc, ns = eng.get_counters()
    J = eng.download_field(E.FIELD_MEAN_INTENSITY)
    Jd = assemble(dec, backends, E.FIELD_MEAN_INTENSITY)
    assert driver.totweight == tw == n
    assert np.array_equal(driver.typecount, tc) and driver.nsteps == ns
    assert np.allclose(Jd, J, rtol=1e-10, atol=1e-13 * J.max())
    # every octant emits the packets that fly into it (the star is on their
    # common corner): nothing has to be handed over
    assert driver.flights_exchanged == 0
    for b in backends:
        b.engine.close()
