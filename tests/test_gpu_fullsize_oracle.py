"""The engine against the CPU oracle AT BASELINE.json's sizes (VERDICT r05,
"What's weak" 1): configs 2, 3 and 4 on 256^3 and the large-grid build of the
hydrogen-only kernel on 336^3.

The engine converges the model; the state is downloaded into the oracle (the
way bench.py's cpu_baseline leg does it); both shoot the SAME packets of one
iteration - first generation and every re-emission generation - and every
accumulator of every cell is compared, then the cell update from identical
integrals. The launch is large enough for everything that only exists at
scale to engage: the 512-thread blocks with the 2048-slot combining table
(1024 / 4096 beyond 2^25 cells), the padded march, tile rounds with units of
up to 16384 flights and row compaction, parking at positions, the multi-ion
first generation from pre-computed emission rows, the temperature pipeline with
its straggler kernel. The range classes of the sort key, which the engine
turns on by itself from 2^22 packets per source, are forced on in a second
engine run against the same oracle tallies.

Reference: src/CartesianDensityGrid.cpp:375-452 (interact),
src/DensityGrid.hpp:150-197 (update_integrals),
src/PhotonSource.cpp:272-308 (reemit),
src/IonizationStateCalculator.cpp / TemperatureCalculator.cpp (cell update).
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NPACKET = 1500000


@pytest.fixture(scope="module")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


def engine_for(model, ncell):
    from test_gpu_fullsize_physics import make
    from test_gpu_transport import make_engine
    if model == "stromgren":
        return make_engine(ncell, track_heating=False)
    return make(model, ncell)


def oracle_for(oracle, model, ncell, eng):
    """The oracle's simulation of the model, holding the engine's cell state."""
    from cmacionize_amd import engine as E
    if model == "lexington":
        sim = oracle.lexington_simulation(ncell)
        for ion in range(14):
            sim.x[ion][:] = eng.download_field(E.FIELD_IONIC_FRACTION + ion)
        sim.temperature[:] = eng.download_field(E.FIELD_TEMPERATURE)
    else:
        sim = oracle.stromgren_simulation(ncell, diffuse=(model == "diffuse"),
                                          compact=True)
        sim.x[0][:] = eng.download_field(E.FIELD_IONIC_FRACTION)
        sim.temperature[:] = eng.download_field(E.FIELD_TEMPERATURE)
    return sim


def accumulators(eng, model):
    from cmacionize_amd import engine as E
    if model == "lexington":
        return ([eng.download_field(E.FIELD_MEAN_INTENSITY + k)
                 for k in range(14)] +
                [eng.download_field(E.FIELD_HEATING + k) for k in range(2)])
    return [eng.download_field(E.FIELD_MEAN_INTENSITY)]


def oracle_accumulators(sim, model):
    if model == "lexington":
        return [np.asarray(sim.J[k]) for k in range(14)] + \
            [np.asarray(sim.heating[k]) for k in range(2)]
    return [np.asarray(sim.J[0])]


# Path lengths of the incremental marcher against the oracle's (the reference's
# arithmetic): a flight that starts somewhere inside a cell (every re-emitted
# one) and runs nearly parallel to a wall - direction cosine c along the wall's
# normal - crosses that wall at a path parameter known to ulp(position) / c;
# the marcher that parametrises the ray from its origin and the one that
# re-starts from the last wall place that crossing ~1e-14 / c cells apart, so a
# few 1e-7 cells of path length move between the two neighbouring cells (seen
# at 256^3: pairs of neighbours with differences of +d and -d, d up to 1.1e-12
# of the largest J; the sum over the grid agrees to 2e-15). With the exact
# marcher the engine's steps are the oracle's bit for bit. Flights from the
# star start on a cell corner: no such term in config 2.
ATOL = {"stromgren": 1e-13, "diffuse": 1e-11, "lexington": 1e-6}


def compare_shoot(eng, sim, model, seed, loop, n, rtol, atol=None):
    """One iteration's transport on both sides; returns the engine's total
    weight."""
    atol = ATOL[model] if atol is None else atol
    eng.reset_grid()
    eng.get_timing(reset=True)
    eng.shoot(seed, loop, 0, n)
    tw, tc, ns = eng.get_counters()
    got = accumulators(eng, model)
    assert tw == sim.totweight == n
    if model == "lexington":
        # device pow / log10 differ from libm by ulps: a frequency lands on
        # the other side of a threshold once in a long while (the 24^3 test
        # allows 3 in 4e4 packets x 6 iterations)
        assert np.abs(tc - sim.typecount).max() <= 8, (tc, sim.typecount)
    else:
        assert np.array_equal(tc, sim.typecount), (tc, sim.typecount)
    want = oracle_accumulators(sim, model)
    for k, (a, b) in enumerate(zip(got, want)):
        scale = np.abs(b).max()
        assert scale > 0. or k >= 2
        if model == "lexington":
            # a packet whose frequency crossed a threshold (above) flies
            # another path: its cells differ by one packet's path length
            close = np.isclose(a, b, rtol=rtol, atol=rtol * scale)
            assert (~close).sum() <= 4000, (k, (~close).sum())
            assert abs(a.sum() - b.sum()) <= 1e-5 * abs(b.sum()) + 1e-300
        else:
            assert np.allclose(a, b, rtol=rtol, atol=atol * scale), \
                (k, np.abs(a - b).max() / scale)
            assert abs(a.sum() - b.sum()) <= 1e-12 * b.sum()
    return tw, ns


@pytest.mark.parametrize("model,ncell", [("stromgren", 256), ("diffuse", 256),
                                         ("lexington", 256),
                                         ("stromgren", 336)])
def test_fullsize_matches_oracle(oracle, model, ncell):
    from cmacionize_amd import engine as E
    from test_gpu_fullsize_physics import converge
    eng = engine_for(model, ncell)
    # lexington: the temperature solve starts with loop 4
    converge(eng, 6, 4000000)
    sim = oracle_for(oracle, model, ncell, eng)
    seed, loop, n = 11, 60, NPACKET
    t0 = time.perf_counter()
    sim.reset()
    sim.totweight = 0.
    sim.typecount[:] = 0.
    sim.shoot(seed, loop, 0, n)
    print("oracle: %d packets on %d^3 %s in %.1f s" %
          (n, ncell, model, time.perf_counter() - t0))
    rtol = 1e-6 if model == "lexington" else 1e-9

    # 1. the default tuning: whatever bench.py's step runs at this size
    tw, ns = compare_shoot(eng, sim, model, seed, loop, n, rtol)
    launches = eng.get_launch_times()
    if model != "stromgren":
        # re-emission generations flew as tile rounds
        assert len(launches) >= 5
    assert 100. < ns / n < 400.

    # 2. what the engine turns on by itself at 1e8 packets: range classes in
    # the sort key (2^22 / 2^24 packets per source), and launches split at
    # max_packets_per_launch with parking per launch
    for tuning in (dict(sort_tau_bits=3 if model != "lexington" else 2),
                   dict(sort_tau_bits=-1, max_packets_per_launch=600000)):
        eng.set_tuning(**tuning)
        compare_shoot(eng, sim, model, seed, loop, n, rtol)
    eng.set_tuning(sort_tau_bits=-1, max_packets_per_launch=1 << 27)
    if model == "diffuse":
        # the reference's arithmetic in the march: no conditioning term
        eng.set_tuning(exact_dda=1)
        compare_shoot(eng, sim, model, seed, loop, n, rtol, atol=1e-13)
        eng.set_tuning(exact_dda=0)

    # 3. the cell update from identical integrals. The oracle solves a slab
    # of cells through the centre of the grid (the whole grid takes minutes
    # on the host for the multi-ion model): source cells, ionized region,
    # ionization front and neutral gas are all in it
    want = oracle_accumulators(sim, model)
    if model == "lexington":
        for k in range(14):
            eng.upload_field(E.FIELD_MEAN_INTENSITY + k, want[k])
        for k in range(2):
            eng.upload_field(E.FIELD_HEATING + k, want[14 + k])
    else:
        eng.upload_field(E.FIELD_MEAN_INTENSITY, want[0])
    eng.update_cells(loop, tw)
    eng.synchronize()
    plane = ncell * ncell
    first = (ncell // 2 - 1) * plane
    count = (2 if model == "lexington" else 8) * plane
    t0 = time.perf_counter()
    sim.update_range(loop, sim.totweight, first, count)
    print("oracle: update of %d cells in %.1f s" %
          (count, time.perf_counter() - t0))
    sl = slice(first, first + count)
    if model == "lexington":
        T = eng.download_field(E.FIELD_TEMPERATURE)[sl]
        assert np.allclose(T, sim.temperature[sl], rtol=1e-6, atol=0.)
        assert T.max() > 6000.
        for ion in range(14):
            x = eng.download_field(E.FIELD_IONIC_FRACTION + ion)[sl]
            ref = np.asarray(sim.x[ion])[sl]
            ok = np.isclose(x, ref, rtol=1e-5, atol=1e-300) | \
                (np.isnan(x) & np.isnan(ref))
            assert ok.all(), ion
    else:
        x = eng.download_field(E.FIELD_IONIC_FRACTION)[sl]
        assert np.array_equal(x, np.asarray(sim.x[0])[sl])
        assert x.min() < 1e-3 and x.max() > 0.5  # front inside the slab
    eng.close()
