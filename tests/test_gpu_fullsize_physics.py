"""GPU tests at BASELINE.json's full sizes for the configs with re-emission:
stromgren_diffuse.param and lexingtonHII40.param on 256^3 (configs 3 and 4)
and lexingtonHII40.param on 512^3 in 2 x 2 x 2 blocks (config 5, all blocks on
one device). The oracle is too slow to be the checker at these sizes, so the
tests check size-independent properties: every packet ends exactly once,
additivity over packet ranges (disjoint Philox counters), invariance under
everything that only reorders work (aggregation mode, re-emission in passes
vs in place, launch splitting), and decomposed == undivided.

At these packet counts the block combining tables of the multi-ion kernel
(128 slots, periodic write-backs) and the per-generation queues actually fill,
which the 12^3 - 32^3 oracle tests never reach."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NCELL = 256
NFIELD = {"diffuse": 1, "lexington": 16}


def make(model, ncell=NCELL):
    from cmacionize_amd import GpuEngine, STROMGREN as S
    from test_gpu_domain import configure, lexington_fields
    heat = model == "lexington"
    eng = GpuEngine((ncell,) * 3, S["anchor"], S["sides"], (0, 0, 0),
                    device=0, track_heating=heat)
    if heat:
        dens, temp = lexington_fields(ncell)
        configure(eng, model, ncell ** 3, dens.ravel(), temp.ravel())
    else:
        configure(eng, model, ncell ** 3)
    return eng


def converge(eng, iterations, npacket):
    for loop in range(iterations):
        eng.reset_grid()
        eng.shoot(42, loop, 0, npacket)
        tw, _, _ = eng.get_counters()
        eng.update_cells(loop, tw)


def integrals(eng, model):
    from cmacionize_amd import engine as E
    return [eng.download_field(E.FIELD_MEAN_INTENSITY + k)
            for k in range(NFIELD[model])]


@pytest.fixture(scope="module", params=["diffuse", "lexington"])
def converged(request):
    model = request.param
    eng = make(model)
    # lexington: the temperature solve starts with loop 4
    converge(eng, 7, 4000000)
    yield model, eng
    eng.close()


def test_fullsize_reemission_conservation_and_additivity(converged):
    model, eng = converged
    n = 6000000
    eng.reset_grid()
    eng.get_timing(reset=True)
    eng.shoot(7, 50, 0, n)
    tw, tc, ns = eng.get_counters()
    launches = eng.get_launch_times()
    J = integrals(eng, model)
    # every packet ends exactly once whatever its number of re-emissions
    assert tw == n and tc.sum() == n and tc[3] > 0
    if model == "diffuse":
        assert tc[1] > 0  # some leave the box as diffuse H photons
    # several re-emission generations flew, each smaller than the one before
    flights = [f for _, f in launches]
    assert len(flights) >= 5 and flights[0] == n
    # (tile rounds: launch k flies whatever is in flight after k - 1 tile
    # crossings - first the re-emissions of the whole first generation)
    assert all(b < a for a, b in zip(flights, flights[1:]))
    assert 0.25 * n < flights[1] < 0.6 * n
    assert 100. < ns / n < 400.
    # additivity over packet ranges, accumulated in place
    eng.reset_grid()
    for first, count in ((0, 1234567), (1234567, 3000000),
                         (4234567, n - 4234567)):
        eng.shoot(7, 50, first, count)
    tw2, tc2, ns2 = eng.get_counters()
    assert tw2 == tw and np.array_equal(tc2, tc) and ns2 == ns
    for a, b in zip(integrals(eng, model), J):
        assert np.allclose(a, b, rtol=1e-10, atol=1e-13 * np.abs(b).max())
    assert all(np.isfinite(a).all() for a in J)


def test_fullsize_reemission_reordering_invariance(converged):
    """aggregation off / on, re-emission in passes / in place, eager refills,
    split launches: the same packets, the same tallies."""
    model, eng = converged
    n = 2000000
    default = dict(aggregate=2, aggregate_reemit=0, reemit_passes=1,
                   reemit_inline_below=4096, reemit_max_passes=12,
                   refill_threshold_reemit=32, sort_packets=1,
                   max_packets_per_launch=1 << 27, tile_rounds=1,
                   tile_min_flights=100000, tile_refill_threshold=16,
                   tile_min_per_item=-1, tile_counting_sort=1,
                   pre_emission=1, defer_weights=1)
    results = []
    for kw in (dict(),
               dict(tile_rounds=0),
               dict(tile_min_flights=0, tile_min_per_item=0, tile_refill_threshold=40),
               dict(tile_counting_sort=0),
               dict(pre_emission=0),
               dict(defer_weights=0),
               dict(aggregate=0, sort_packets=0),
               dict(reemit_passes=0),
               dict(aggregate_reemit=2, refill_threshold_reemit=8,
                    max_packets_per_launch=700001),
               dict(reemit_inline_below=0, reemit_max_passes=4)):
        tuning = dict(default)
        tuning.update(kw)
        eng.set_tuning(**tuning)
        eng.reset_grid()
        eng.shoot(11, 60, 3, n)
        results.append((eng.get_counters(), integrals(eng, model)))
    eng.set_tuning(**default)
    (tw0, tc0, ns0), J0 = results[0]
    assert tw0 == n
    for (tw, tc, ns), J in results[1:]:
        assert tw == tw0 and np.array_equal(tc, tc0) and ns == ns0
        for a, b in zip(J, J0):
            assert np.allclose(a, b, rtol=1e-10,
                               atol=1e-13 * np.abs(b).max())


def test_fullsize_reemission_decomposition_invariance(converged):
    """256^3 in 2 x 2 x 2 blocks of 128^3 against the undivided grid, same
    state, same packets: re-emitted flights cross the block faces."""
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.simulation import (DomainDecomposition,
                                           DomainGpuBackend,
                                           LocalDomainDriver)
    from test_gpu_domain import assemble, configure
    model, eng = converged
    n = 3000000
    shape = (NCELL,) * 3
    dens = eng.download_field(E.FIELD_NUMBER_DENSITY).reshape(shape)
    temp = eng.download_field(E.FIELD_TEMPERATURE).reshape(shape)
    xs = [eng.download_field(E.FIELD_IONIC_FRACTION + k).reshape(shape)
          for k in range(14)]
    dec = DomainDecomposition(shape, (2, 2, 2))
    backends = []
    for rank in range(dec.world):
        b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                             track_heating=model == "lexington",
                             export_capacity=n)
        off, size = dec.block(rank)
        sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
        configure(b.engine, model, int(np.prod(size)), dens[sl].ravel(),
                  temp[sl].ravel())
        b.engine.upload_cells(dens[sl].ravel(), temp[sl].ravel(),
                              np.array([x[sl].ravel() for x in xs]))
        backends.append(b)
    driver = LocalDomainDriver(backends, dec)
    driver.iteration(70, n, 3, update=False)
    eng.reset_grid()
    eng.shoot(3, 70, 0, n)
    tw, tc, ns = eng.get_counters()
    assert driver.totweight == tw == n
    assert np.array_equal(driver.typecount, tc) and driver.nsteps == ns
    assert driver.flights_exchanged > 0
    for k, ref in enumerate(integrals(eng, model)):
        got = assemble(dec, backends, E.FIELD_MEAN_INTENSITY + k)
        assert np.allclose(got, ref, rtol=1e-10,
                           atol=1e-13 * np.abs(ref).max()), k
    for b in backends:
        b.engine.close()


def test_fullsize_cell_update_invariance(converged):
    """the 256^3 cell update - for lexingtonHII40 the temperature solve of
    ~7e6 cells with its ~1e3 stragglers - gives the same state, bit for bit,
    as one kernel, as the pipeline of kernels (the default), with all slots
    finished by the one-wave-per-slot kernel from 4096 on, and solved in
    three slabs of cells (cmi_gpu_update_cells_range, what the copies of a
    block do); the inputs are restored afterwards."""
    from cmacionize_amd import engine as E
    model, eng = converged
    n = 3000000
    eng.reset_grid()
    eng.shoot(13, 80, 0, n)
    tw, _, _ = eng.get_counters()
    nacc = 16 if model == "lexington" else 1
    state = [E.FIELD_TEMPERATURE] + \
        [E.FIELD_IONIC_FRACTION + i for i in range(14)]
    inputs = state + [E.FIELD_MEAN_INTENSITY + i for i in range(nacc)]
    before = {f: eng.download_field(f) for f in inputs}

    def restore():
        for f in inputs:
            eng.upload_field(f, before[f])

    ncells = NCELL ** 3
    ways = [dict(temperature_pipeline=1, temperature_finish_slots=1024)]
    if model == "lexington":
        ways += [dict(temperature_pipeline=0),
                 dict(temperature_pipeline=1, temperature_finish_slots=0),
                 dict(temperature_pipeline=1, temperature_finish_slots=4096)]
    ways.append("slabs")
    reference = None
    for way in ways:
        restore()
        if way == "slabs":
            eng.set_tuning(**ways[0])
            cuts = [0, ncells // 3 + 5, 2 * (ncells // 3) - 7, ncells]
            for a, b in zip(cuts, cuts[1:]):
                eng.update_cells_range(9, tw, a, b - a)
        else:
            eng.set_tuning(**way)
            eng.update_cells(9, tw)
        if reference is None:
            reference = {f: eng.download_field(f) for f in state}
            T = reference[E.FIELD_TEMPERATURE]
            xH = reference[E.FIELD_IONIC_FRACTION]
            assert np.isfinite(T).all() and (xH != before[state[1]]).any()
            if model == "lexington":
                solved = T != before[E.FIELD_TEMPERATURE]
                assert solved.sum() > 1000000
                assert 6000. < T[xH < 0.1].mean() < 12000.
            continue
        for f in state:
            # (H-only models: the metals' balance is 0 / 0 in ionized cells,
            # as in the reference - NaN in the same cells either way)
            assert np.array_equal(eng.download_field(f), reference[f],
                                  equal_nan=True), (way, f)
    eng.set_tuning(**ways[0])
    restore()


def test_config5_512_cubed_in_blocks_equals_whole_grid():
    """BASELINE config 5 at its full size: lexingtonHII40 on 512^3 as
    2 x 2 x 2 blocks of 256^3 (the 8-GPU decomposition, all eight engines on
    this one device, handing flights over through device buffers) against ONE
    engine holding the whole 512^3 grid (36 GB): identical packet counters and
    step counts, every accumulator field summed over the grid at 1e-9, and
    a sub-volume around the source cell by cell."""
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.simulation import (DomainDecomposition,
                                           DomainGpuBackend,
                                           LocalDomainDriver)
    from test_gpu_domain import configure
    ncell, n = 512, 4000000
    whole = make("lexington", ncell)
    converge(whole, 5, 2000000)
    shape = (ncell,) * 3
    dec = DomainDecomposition(shape, (2, 2, 2))
    backends = []
    dens = whole.download_field(E.FIELD_NUMBER_DENSITY).reshape(shape)
    temp = whole.download_field(E.FIELD_TEMPERATURE).reshape(shape)
    xs = [whole.download_field(E.FIELD_IONIC_FRACTION + k).reshape(shape)
          for k in range(14)]
    for rank in range(dec.world):
        b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                             track_heating=True, export_capacity=n)
        off, size = dec.block(rank)
        sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
        d = np.ascontiguousarray(dens[sl]).ravel()
        t = np.ascontiguousarray(temp[sl]).ravel()
        configure(b.engine, "lexington", d.size, d, t)
        b.engine.upload_cells(d, t, np.array([x[sl].ravel() for x in xs]))
        backends.append(b)
    del xs
    driver = LocalDomainDriver(backends, dec)
    driver.iteration(9, n, 5, update=False)
    whole.reset_grid()
    whole.shoot(5, 9, 0, n)
    tw, tc, ns = whole.get_counters()
    assert driver.totweight == tw == n
    assert np.array_equal(driver.typecount, tc) and driver.nsteps == ns
    assert driver.flights_exchanged > 0 and driver.rounds > 0
    assert 200. < ns / n < 800.
    h = ncell // 2
    for k in range(16):
        ref = whole.download_field(E.FIELD_MEAN_INTENSITY + k).reshape(shape)
        total = 0.
        for rank, b in enumerate(backends):
            off, size = dec.block(rank)
            got = b.engine.download_field(
                E.FIELD_MEAN_INTENSITY + k).reshape(size)
            total += got.sum()
            # the 32^3 cells of this octant nearest to the star, cell by cell
            gs = tuple(slice(h - 32 - off[a], h - off[a]) if off[a] == 0
                       else slice(0, 32) for a in range(3))
            rs = tuple(slice(h - 32, h) if off[a] == 0 else slice(h, h + 32)
                       for a in range(3))
            assert np.allclose(got[gs], ref[rs], rtol=1e-10,
                               atol=1e-13 * np.abs(ref).max()), (k, rank)
        assert abs(total - ref.sum()) <= 1e-9 * abs(ref.sum()), k
        assert ref.sum() != 0. or k >= 14
    whole.close()
    for b in backends:
        b.engine.close()
