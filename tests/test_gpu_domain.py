"""GPU tests of the domain-decomposed mode (a block of the grid per engine,
flights handed over between blocks): the decomposed run must reproduce the run
on the undivided grid - the marcher's state travels with a packet, so every
path length is bit-identical and only the order of the atomic sums differs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def configure(eng, model, n_local, density=None, temperature=None):
    """benchmarks/stromgren*.param or lexingtonHII40.param on an engine."""
    from cmacionize_amd import STROMGREN as S
    x = np.zeros((14, n_local))
    x[0] = 1.e-6
    x[1] = 1.e-6
    if model in ("stromgren", "diffuse"):
        eng.set_sources(S["source_position"], S["source_weight"],
                        S["luminosity"])
        eng.set_spectrum_monochromatic(S["frequency"])
        sigma = np.zeros(14)
        sigma[0] = S["sigma_H"]
        alpha = np.zeros(14)
        alpha[0] = S["alpha_H"]
        eng.set_cross_sections_fixed(sigma)
        eng.set_recombination_rates_fixed(alpha)
        if model == "diffuse":
            eng.set_reemission(1)
        eng.upload_cells(np.full(n_local, S["density"]),
                         np.full(n_local, S["temperature"]), x)
    else:
        from test_oracle_physics import LEX
        eng.set_sources([[0., 0., 0.]], [1.], 4.26e49)
        eng.set_spectrum_planck(40000.)
        eng.set_cross_sections_verner()
        eng.set_recombination_rates_verner()
        eng.set_abundances(LEX[1:])
        eng.set_reemission(1)
        eng.set_temperature_params(do_temperature_calculation=1,
                                   pah_heating_factor=0.)
        eng.upload_cells(density, temperature, x)


def lexington_fields(ncell):
    from cmacionize_amd.simulation import PC
    ax = -5. * PC + (np.arange(ncell) + 0.5) * (10. * PC / ncell)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    vacuum = np.sqrt(X * X + Y * Y + Z * Z) <= 3.e16
    return np.where(vacuum, 0., 1.e8), np.where(vacuum, 0., 8000.)


def assemble(decomposition, backends, field):
    out = np.zeros(decomposition.ncell)
    for rank, b in enumerate(backends):
        off, size = decomposition.block(rank)
        out[off[0]:off[0] + size[0], off[1]:off[1] + size[1],
            off[2]:off[2] + size[2]] = \
            b.engine.download_field(field).reshape(size)
    return out.ravel()


def run_pair(model, ncell, blocks, npacket, iterations, tuning=None):
    """Yield (whole-grid engine, decomposition, backends, driver) after each
    iteration of both runs."""
    from cmacionize_amd import GpuEngine, STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.simulation import (DomainDecomposition,
                                           DomainGpuBackend,
                                           LocalDomainDriver)
    heat = model == "lexington"
    whole = GpuEngine((ncell,) * 3, S["anchor"], S["sides"], (0, 0, 0),
                      device=0, track_heating=heat)
    dens = temp = None
    if model == "lexington":
        dens, temp = lexington_fields(ncell)
    configure(whole, model, ncell ** 3,
              None if dens is None else dens.ravel(),
              None if temp is None else temp.ravel())
    if tuning:
        whole.set_tuning(**tuning)
    dec = DomainDecomposition((ncell,) * 3, blocks)
    backends = []
    for rank in range(dec.world):
        b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                             track_heating=heat, export_capacity=npacket)
        off, size = dec.block(rank)
        sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
        configure(b.engine, model, int(np.prod(size)),
                  None if dens is None else dens[sl].ravel(),
                  None if temp is None else temp[sl].ravel())
        if tuning:
            b.engine.set_tuning(**tuning)
        backends.append(b)
    driver = LocalDomainDriver(backends, dec)
    nfield = 16 if model == "lexington" else 1
    for loop in range(iterations):
        whole.reset_grid()
        whole.shoot(42, loop, 0, npacket)
        tw, tc, ns = whole.get_counters()
        driver.iteration(loop, npacket, 42, update=False)
        # the integrals of this iteration, before the cell update
        J = [assemble(dec, backends, E.FIELD_MEAN_INTENSITY + k)
             for k in range(nfield)]
        Jref = [whole.download_field(E.FIELD_MEAN_INTENSITY + k)
                for k in range(nfield)]
        whole.update_cells(loop, tw)
        for b in backends:
            b.update_cells(loop, driver.totweight)
        yield loop, whole, (tw, tc, ns), dec, backends, driver, J, Jref
        # the closed forms of the balance amplify the 1e-15 differences of the
        # sums (1 - sqrt(1 + small)): start the next iteration from one state
        dens_w = whole.download_field(E.FIELD_NUMBER_DENSITY).reshape(
            (ncell,) * 3)
        temp_w = whole.download_field(E.FIELD_TEMPERATURE).reshape(
            (ncell,) * 3)
        x_w = [whole.download_field(E.FIELD_IONIC_FRACTION + k).reshape(
            (ncell,) * 3) for k in range(14)]
        for rank, b in enumerate(backends):
            off, size = dec.block(rank)
            sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
            b.engine.upload_cells(dens_w[sl].ravel(), temp_w[sl].ravel(),
                                  np.array([x[sl].ravel() for x in x_w]))
    whole.close()
    for b in backends:
        b.engine.close()


def check_integrals(J, Jref):
    for k, (a, b) in enumerate(zip(J, Jref)):
        # identical path lengths; only the order of the additions differs
        assert np.allclose(a, b, rtol=1e-11, atol=1e-13 * np.abs(b).max()), k
        assert np.abs(b).max() > 0. or k >= 14


@pytest.mark.parametrize("blocks", [(2, 2, 2), (3, 1, 1), (1, 2, 3)])
def test_decomposed_stromgren_equals_whole_grid(blocks):
    from cmacionize_amd import engine as E
    for loop, whole, (tw, tc, ns), dec, backends, driver, J, Jref in run_pair(
            "stromgren", 24, blocks, 50000, 3):
        assert driver.totweight == tw == 50000
        assert np.array_equal(driver.typecount, tc)
        assert driver.nsteps == ns  # same cells crossed, packet by packet
        if blocks == (2, 2, 2):
            # the star sits on the corner the 8 octants share: every packet is
            # emitted by the octant it flies into (the zero-length first steps
            # are taken before ownership is decided) and, flying straight
            # from that corner, never leaves it for another octant
            assert driver.rounds == 0 and driver.flights_exchanged == 0
        else:
            assert driver.rounds >= 1 and driver.flights_exchanged > 0
        check_integrals(J, Jref)
        x = assemble(dec, backends, E.FIELD_IONIC_FRACTION)
        ref = whole.download_field(E.FIELD_IONIC_FRACTION)
        assert np.allclose(x, ref, rtol=1e-6, atol=0.)
    assert (ref < 0.5).any() and (ref > 0.5).any()


def test_decomposed_diffuse_equals_whole_grid():
    """Re-emission happens in the block where a packet is absorbed; the
    re-emitted flight may cross blocks again."""
    from cmacionize_amd import engine as E
    for loop, whole, (tw, tc, ns), dec, backends, driver, J, Jref in run_pair(
            "diffuse", 24, (2, 2, 2), 40000, 3,
            tuning=dict(reemit_inline_below=64)):
        assert driver.totweight == tw == 40000
        assert np.array_equal(driver.typecount, tc)
        assert driver.nsteps == ns
        assert tc[1] > 0
        check_integrals(J, Jref)
        x = assemble(dec, backends, E.FIELD_IONIC_FRACTION)
        ref = whole.download_field(E.FIELD_IONIC_FRACTION)
        assert np.allclose(x, ref, rtol=1e-6, atol=0.)


def test_decomposed_lexington_equals_whole_grid():
    """All ions, heating terms and the temperature solve, block by block."""
    from cmacionize_amd import engine as E
    for loop, whole, (tw, tc, ns), dec, backends, driver, J, Jref in run_pair(
            "lexington", 24, (2, 2, 2), 30000, 5):
        assert driver.totweight == tw == 30000
        assert np.array_equal(driver.typecount, tc)
        assert driver.nsteps == ns
        check_integrals(J, Jref)
        T = assemble(dec, backends, E.FIELD_TEMPERATURE)
        Tref = whole.download_field(E.FIELD_TEMPERATURE)
        assert np.allclose(T, Tref, rtol=1e-6, atol=0.)
        for ion in (0, 1, 5, 8):
            x = assemble(dec, backends, E.FIELD_IONIC_FRACTION + ion)
            ref = whole.download_field(E.FIELD_IONIC_FRACTION + ion)
            ok = np.isclose(x, ref, rtol=1e-5, atol=1e-300) | \
                (np.isnan(x) & np.isnan(ref))
            assert ok.all(), (loop, ion)
    assert Tref.max() > 6000.


def test_export_buffer_overflow_is_an_error():
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd.engine import EngineError
    from cmacionize_amd.simulation import (DomainDecomposition,
                                           DomainGpuBackend)
    dec = DomainDecomposition((16, 16, 16), (2, 1, 1))
    b = DomainGpuBackend(dec, 1, S["anchor"], S["sides"], device=0,
                         export_capacity=16)
    configure(b.engine, "stromgren", 8 * 16 * 16)
    # a source well inside block 1: the packets that fly towards -x leave it
    b.engine.set_sources([[1.5e16, 0., 0.]], [1.], 4.26e49)
    b.reset_grid()
    b.shoot(1, 0, 0, 10000)
    with pytest.raises(EngineError):
        b.take_exports()
    b.engine.close()


def test_decomposed_grid_with_sources_in_different_blocks():
    """Every block runs through all packet ids and flies those whose source
    lies inside it: three weighted sources in three different blocks."""
    from cmacionize_amd import GpuEngine, STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.simulation import (DomainDecomposition,
                                           DomainGpuBackend,
                                           LocalDomainDriver)
    from test_gpu_transport import SOURCES
    ncell, npacket = 24, 60000
    whole = GpuEngine((ncell,) * 3, S["anchor"], S["sides"], (0, 0, 0),
                      device=0, track_heating=False)
    configure(whole, "stromgren", ncell ** 3)
    whole.set_sources(SOURCES[0], SOURCES[1], S["luminosity"])
    dec = DomainDecomposition((ncell,) * 3, (2, 2, 2))
    backends = []
    for rank in range(dec.world):
        b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                             export_capacity=npacket)
        configure(b.engine, "stromgren", int(np.prod(dec.block(rank)[1])))
        b.engine.set_sources(SOURCES[0], SOURCES[1], S["luminosity"])
        backends.append(b)
    driver = LocalDomainDriver(backends, dec)
    whole.reset_grid()
    whole.shoot(5, 0, 0, npacket)
    tw, tc, ns = whole.get_counters()
    driver.iteration(0, npacket, 5, update=False)
    assert driver.totweight == tw == npacket
    assert np.array_equal(driver.typecount, tc) and driver.nsteps == ns
    J = assemble(dec, backends, E.FIELD_MEAN_INTENSITY)
    Jref = whole.download_field(E.FIELD_MEAN_INTENSITY)
    assert np.allclose(J, Jref, rtol=1e-11, atol=1e-13 * Jref.max())
    # the three sources sit in three different blocks: each emitted its share
    emitted = sorted((b.get_counters()[0] > 0) for b in backends)
    assert sum(emitted) >= 3
    whole.close()
    for b in backends:
        b.engine.close()


# ---------------------------------------------------------------------------
# the decomposed path against the ORACLE (not against the engine's own
# undivided run): the semantics of the reference's block decomposition
# (DensitySubGridCreator + the photon-buffer traffic between DensitySubGrids,
# src/DensitySubGrid.hpp:1137-1274, src/PhotonTraversalTaskContext.hpp:100-278)
# are "the same packets deposit the same path lengths in the same cells as on
# the undivided grid", which is what the oracle computes.
# ---------------------------------------------------------------------------

def decomposed_backends(model, ncell, blocks, npacket, sim=None):
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd.simulation import (DomainDecomposition,
                                           DomainGpuBackend,
                                           LocalDomainDriver)
    heat = model == "lexington"
    dec = DomainDecomposition((ncell,) * 3, blocks)
    dens = temp = None
    if model == "lexington":
        dens = np.asarray(sim.number_density).reshape((ncell,) * 3)
        temp = np.asarray(sim.temperature).reshape((ncell,) * 3)
    backends = []
    for rank in range(dec.world):
        b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                             track_heating=True if heat else False,
                             export_capacity=npacket)
        off, size = dec.block(rank)
        sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
        configure(b.engine, model, int(np.prod(size)),
                  None if dens is None else dens[sl].ravel(),
                  None if temp is None else temp[sl].ravel())
        backends.append(b)
    return dec, backends, LocalDomainDriver(backends, dec)


def upload_state(dec, backends, sim, ncell):
    """the oracle's cell state into every block"""
    shape = (ncell,) * 3
    dens = np.asarray(sim.number_density).reshape(shape)
    temp = np.asarray(sim.temperature).reshape(shape)
    xs = [np.asarray(x).reshape(shape) for x in sim.x]
    for rank, b in enumerate(backends):
        off, size = dec.block(rank)
        sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
        b.engine.upload_cells(dens[sl].ravel(), temp[sl].ravel(),
                              np.array([x[sl].ravel() for x in xs]))


@pytest.mark.parametrize("tiles", [0, 1])
@pytest.mark.parametrize("blocks", [(2, 2, 2), (3, 2, 1)])
def test_decomposed_diffuse_matches_oracle(oracle, blocks, tiles):
    """stromgren_diffuse.param on 24^3 in blocks, LocalDomainDriver, against
    the oracle on the same seeds: identical packet counters, J_H at 1e-9, and
    the bit-exact closed-form cell update per block."""
    from cmacionize_amd import engine as E
    ncell, npacket = 24, 50000
    sim = oracle.stromgren_simulation(ncell, diffuse=True)
    dec, backends, driver = decomposed_backends("diffuse", ncell, blocks,
                                                npacket)
    for b in backends:
        # tiles: the later generations fly in tile rounds inside each block,
        # flights that leave a block from a tile are handed over
        b.engine.set_tuning(reemit_inline_below=64, tile_rounds=tiles,
                            tile_min_flights=0, tile_min_per_item=0)
    for loop in range(3):
        driver.iteration(loop, npacket, 42, update=False)
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, loop, 0, npacket)
        assert driver.totweight == sim.totweight == npacket
        assert np.array_equal(driver.typecount, sim.typecount)
        assert driver.typecount[1] > 0
        J = assemble(dec, backends, E.FIELD_MEAN_INTENSITY)
        assert np.allclose(J, sim.J[0], rtol=1e-9, atol=1e-12 * sim.J[0].max())
        # the cell update, block by block, from the oracle's integrals
        Jo = np.asarray(sim.J[0]).reshape((ncell,) * 3)
        for rank, b in enumerate(backends):
            off, size = dec.block(rank)
            sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
            b.engine.upload_field(E.FIELD_MEAN_INTENSITY, Jo[sl].ravel())
            b.update_cells(loop, driver.totweight)
        # (a device fault is reported by the device call, not inside the
        # oracle's long CPU section)
        for b in backends:
            b.engine.synchronize()
        sim.update(loop, sim.totweight)
        x = assemble(dec, backends, E.FIELD_IONIC_FRACTION)
        assert np.array_equal(x, sim.x[0])
    assert (x < 0.5).any() and (x > 0.5).any()
    for b in backends:
        b.engine.close()


@pytest.mark.parametrize("tiles", [0, 1])
@pytest.mark.parametrize("blocks", [(2, 2, 2), (3, 1, 2), (1, 1, 1)])
def test_decomposed_periodic_diffuse_matches_oracle(oracle, blocks, tiles):
    """A box that is periodic in x and y (CartesianDensityGrid::is_inside,
    src/CartesianDensityGrid.cpp:187-227), cut in blocks: a flight that
    leaves the whole box through a periodic face goes on in the block on the
    other side with its origin shifted by a box side - also when that block is
    the one it left (one block along the axis). Counters and J against the
    oracle on the same seeds."""
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.simulation import (DomainDecomposition,
                                           DomainGpuBackend,
                                           LocalDomainDriver)
    ncell, npacket = 24, 30000
    periodic = (1, 1, 0)
    source = [[0.31 * S["sides"][0], -0.2 * S["sides"][0],
               0.07 * S["sides"][0]]]
    sim = oracle.OracleSimulation((ncell,) * 3, S["anchor"], S["sides"],
                                  periodic=periodic)
    sim.set_sources(source, [1.], S["luminosity"])
    # thin enough that most packets cross the box several times
    sim.set_homogeneous(S["density"], S["temperature"], xH=2.e-5)
    m = sim.model
    m.spectrum_type = oracle.SPECTRUM_MONOCHROMATIC
    m.mono_frequency = S["frequency"]
    m.xsec_type = oracle.XSEC_FIXED
    m.xsec_fixed[0] = S["sigma_H"]
    m.recomb_type = oracle.RECOMB_FIXED
    m.recomb_fixed[0] = S["alpha_H"]
    m.reemit_type = oracle.REEMIT_PHYSICAL

    dec = DomainDecomposition((ncell,) * 3, blocks)
    backends = []
    for rank in range(dec.world):
        b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                             track_heating=True, export_capacity=4 * npacket,
                             periodic=periodic)
        eng = b.engine
        eng.set_sources(source, [1.], S["luminosity"])
        eng.set_spectrum_monochromatic(S["frequency"])
        sigma = np.zeros(14)
        sigma[0] = S["sigma_H"]
        alpha = np.zeros(14)
        alpha[0] = S["alpha_H"]
        eng.set_cross_sections_fixed(sigma)
        eng.set_recombination_rates_fixed(alpha)
        eng.set_reemission(1)
        eng.set_tuning(reemit_inline_below=64, tile_rounds=tiles,
                       tile_min_flights=0, tile_min_per_item=0)
        backends.append(b)
    driver = LocalDomainDriver(backends, dec)
    upload_state(dec, backends, sim, ncell)
    for loop in range(2):
        driver.iteration(loop, npacket, 9, update=False)
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(9, loop, 0, npacket)
        assert driver.totweight == sim.totweight == npacket
        assert np.array_equal(driver.typecount, sim.typecount)
        assert driver.typecount[1] > 0 and driver.typecount[3] > 0
        J = assemble(dec, backends, E.FIELD_MEAN_INTENSITY)
        assert np.allclose(J, sim.J[0], rtol=1e-9, atol=1e-12 * sim.J[0].max())
        h = assemble(dec, backends, E.FIELD_HEATING)
        assert np.allclose(h, sim.heating[0], rtol=1e-9,
                           atol=1e-12 * np.abs(sim.heating[0]).max())
    for b in backends:
        b.engine.close()


@pytest.mark.parametrize("tiles", [0, 1])
def test_decomposed_lexington_matches_oracle(oracle, tiles):
    """Config 5's physics and decomposition at test size: lexingtonHII40 on
    24^3 as 2 x 2 x 2 blocks through LocalDomainDriver against the ORACLE:
    packet counters, all 14 mean intensities and both heating terms at 1e-6
    (the tolerance of the undivided multi-ion test: device pow/log10 ulps),
    then the cell update (ionization balance, temperature solve from loop 4)
    of every block from the oracle's integrals."""
    from cmacionize_amd import engine as E
    ncell, npacket = 24, 40000
    sim = oracle.lexington_simulation(ncell)
    dec, backends, driver = decomposed_backends("lexington", ncell, (2, 2, 2),
                                                npacket, sim)
    upload_state(dec, backends, sim, ncell)
    for b in backends:
        b.engine.set_tuning(tile_rounds=tiles, tile_min_flights=0, tile_min_per_item=0)
    shape = (ncell,) * 3
    exchanged = 0
    for loop in range(6):
        driver.iteration(loop, npacket, 42, update=False)
        exchanged += driver.flights_exchanged
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, loop, 0, npacket)
        assert driver.totweight == sim.totweight == npacket
        # (a frequency within an ulp of a threshold may fall on the other
        # side on the device - as in the undivided test)
        assert np.abs(driver.typecount - sim.typecount).max() <= 3
        for ion in range(14):
            J = assemble(dec, backends, E.FIELD_MEAN_INTENSITY + ion)
            ref = np.asarray(sim.J[ion])
            assert np.allclose(J, ref, rtol=1e-6, atol=1e-6 * ref.max()), ion
            assert abs(J.sum() - ref.sum()) <= 1e-6 * ref.sum()
        for k in range(2):
            h = assemble(dec, backends, E.FIELD_HEATING + k)
            ref = sim.heating[k]
            assert np.allclose(h, ref, rtol=1e-6,
                               atol=1e-6 * np.abs(ref).max())
        for rank, b in enumerate(backends):
            off, size = dec.block(rank)
            sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
            for ion in range(14):
                b.engine.upload_field(
                    E.FIELD_MEAN_INTENSITY + ion,
                    np.asarray(sim.J[ion]).reshape(shape)[sl].ravel())
            for k in range(2):
                b.engine.upload_field(
                    E.FIELD_HEATING + k,
                    np.asarray(sim.heating[k]).reshape(shape)[sl].ravel())
            b.update_cells(loop, driver.totweight)
        # (a device fault is reported by the device call, not inside the
        # oracle's long CPU section)
        for b in backends:
            b.engine.synchronize()
        sim.update(loop, sim.totweight)
        T = assemble(dec, backends, E.FIELD_TEMPERATURE)
        assert np.allclose(T, sim.temperature, rtol=1e-6, atol=0.), loop
        for ion in range(14):
            x = assemble(dec, backends, E.FIELD_IONIC_FRACTION + ion)
            ref = np.asarray(sim.x[ion])
            ok = np.isclose(x, ref, rtol=1e-5, atol=1e-300) | \
                (np.isnan(x) & np.isnan(ref))
            assert ok.all(), (loop, ion)
        upload_state(dec, backends, sim, ncell)
    assert exchanged > 0  # re-emitted flights did cross block faces
    assert sim.temperature.max() > 6000. and sim.temperature.min() == 500.
    for b in backends:
        b.engine.close()


@pytest.mark.parametrize("model,blocks", [("diffuse", (2, 2, 2)),
                                          ("lexington", (2, 2, 2)),
                                          ("lexington", (3, 1, 2))])
def test_decomposed_matches_task_based_oracle(oracle, model, blocks):
    """The decomposed mode against the oracle of the reference's TASK-BASED
    semantics (oracle/cmio_subgrid.c: DensitySubGrid::interact,
    src/DensitySubGrid.hpp:1137-1274, one subgrid per block): same packets,
    same subgrid changes. The task-based tallies carry the abundance of the
    ion's element and other heating thresholds
    (src/SourceDiscretePhotonTaskContext.hpp:172-180,
    src/DensitySubGrid.hpp:607,611); the engine keeps the classic path's
    definitions in every mode, so the comparison goes through those two exact
    relations (tests/test_oracle_subgrid.py)."""
    from cmacionize_amd import engine as E
    from test_oracle_subgrid import ION_ELEMENT
    ncell, npacket = 24, 40000
    sim = (oracle.lexington_simulation(ncell) if model == "lexington"
           else oracle.stromgren_simulation(ncell, diffuse=True))
    dec, backends, driver = decomposed_backends(model, ncell, blocks, npacket,
                                                sim)
    upload_state(dec, backends, sim, ncell)
    for b in backends:
        b.engine.set_tuning(tile_min_flights=0, tile_min_per_item=0)
    driver.iteration(0, npacket, 42, update=False)
    sim.reset()
    sim.totweight = 0.
    sim.typecount[:] = 0.
    tw, tc, ns, nh = sim.shoot_subgrids(blocks, 42, 0, 0, npacket)
    assert driver.totweight == tw == npacket
    assert np.abs(driver.typecount - tc).max() <= (3 if model == "lexington"
                                                   else 0)
    # every change of subgrid is a flight handed over - except the first,
    # zero-length steps out of the star's corner, which every block of the
    # engine takes itself before asking whose packet it is
    assert 0 < driver.flights_exchanged <= nh
    if model == "diffuse":
        assert driver.nsteps == ns
    gas = np.asarray(sim.number_density) > 0.
    m = sim.model
    tol = 1e-6 if model == "lexington" else 1e-9
    nuH, nuHe = oracle.eV_to_Hz(13.6), oracle.eV_to_Hz(24.6)
    Jc = []
    for ion in range(14 if model == "lexington" else 1):
        J = assemble(dec, backends, E.FIELD_MEAN_INTENSITY + ion)
        Jc.append(J)
        A = 1. if ion == 0 else m.abundance[ION_ELEMENT[ion]]
        ref = np.asarray(sim.J[ion])
        assert np.allclose(A * J[gas], ref[gas], rtol=tol,
                           atol=tol * 1e-3 * max(ref.max(), 1e-300)), ion
        # (cells without gas: tallied by the task-based path only)
        assert np.all(J[~gas] == 0.)
    if model == "lexington":
        hH = assemble(dec, backends, E.FIELD_HEATING)
        ref = np.asarray(sim.heating[0])
        assert np.allclose((hH + Jc[0] * (nuH - 3.288e15))[gas], ref[gas],
                           rtol=1e-6, atol=1e-6 * np.abs(ref).max())
        hHe = assemble(dec, backends, E.FIELD_HEATING + 1)
        ref = np.asarray(sim.heating[1])
        AHe = m.abundance[1]
        assert np.allclose(AHe * (hHe + Jc[1] * (nuHe - 5.948e15))[gas],
                           ref[gas], rtol=1e-6,
                           atol=1e-6 * np.abs(ref).max())
    for b in backends:
        b.engine.close()


def test_group_exchange_equals_python_routing():
    """cmi_gpu_group_exchange_flights (routing kernel on the source device,
    rows written into the owner's inbox) against the Python hand-over of
    LocalDomainDriver (torch sort by owner): same flights reach the same
    blocks - identical counters, integrals at 1e-11."""
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.engine import EngineGroup
    from cmacionize_amd.simulation import (DomainDecomposition,
                                           DomainGpuBackend,
                                           LocalDomainDriver)
    ncell, npacket = 24, 40000
    results = []
    for use_group in (False, True):
        dec = DomainDecomposition((ncell,) * 3, (2, 3, 1))
        backends = []
        for rank in range(dec.world):
            b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                                 export_capacity=npacket)
            configure(b.engine, "diffuse", int(np.prod(dec.block(rank)[1])))
            b.engine.set_tuning(reemit_inline_below=64)
            backends.append(b)
        if not use_group:
            driver = LocalDomainDriver(backends, dec)
            driver.iteration(0, npacket, 42, update=False)
            tw, tc, ns = driver.totweight, driver.typecount, driver.nsteps
            assert driver.flights_exchanged > 0
        else:
            group = EngineGroup([b.engine for b in backends])
            for b in backends:
                b.reset_grid()
                b.shoot(42, 0, 0, npacket)
            rounds = flights = 0
            while True:
                n = group.exchange_flights(42, 0)
                if n == 0:
                    break
                rounds += 1
                flights += n
            assert rounds >= 1 and flights > 0
            tw, tc, ns = 0., np.zeros(4), 0
            for b in backends:
                t, c, n = b.get_counters()
                tw += t
                tc += np.asarray(c)
                ns += n
            group.close()
        J = assemble(dec, backends, E.FIELD_MEAN_INTENSITY)
        results.append((tw, tc, ns, J))
        for b in backends:
            b.engine.close()
    (tw0, tc0, ns0, J0), (tw1, tc1, ns1, J1) = results
    assert tw0 == tw1 == npacket and np.array_equal(tc0, tc1) and ns0 == ns1
    assert np.allclose(J1, J0, rtol=1e-11, atol=1e-13 * J0.max())


def test_group_with_copies_of_the_source_block_matches_oracle(oracle):
    """A star in the middle of one octant of a 2x2x2 decomposition and THREE
    engines for that block: the copies share the block's packets by packet id
    (emission and incoming flights), cmi_gpu_group_reduce_accumulators sums
    their integrals into each (DensitySubGridCreator::update_original_counters,
    src/DensitySubGridCreator.hpp:556-574), each then solves the same cells.
    Counters and J against the oracle on the undivided grid; the copies end
    bit-identical to each other."""
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.engine import EngineGroup
    from cmacionize_amd.simulation import DomainDecomposition, DomainGpuBackend
    ncell, npacket = 24, 40000
    side = S["sides"][0]
    source = [[0.27 * side, -0.23 * side, 0.21 * side]]
    sim = oracle.OracleSimulation((ncell,) * 3, S["anchor"], S["sides"])
    sim.set_sources(source, [1.], S["luminosity"])
    sim.set_homogeneous(S["density"], S["temperature"], xH=1.e-6)
    m = sim.model
    m.spectrum_type = oracle.SPECTRUM_MONOCHROMATIC
    m.mono_frequency = S["frequency"]
    m.xsec_type = oracle.XSEC_FIXED
    m.xsec_fixed[0] = S["sigma_H"]
    m.recomb_type = oracle.RECOMB_FIXED
    m.recomb_fixed[0] = S["alpha_H"]
    m.reemit_type = oracle.REEMIT_PHYSICAL

    dec = DomainDecomposition((ncell,) * 3, (2, 2, 2))
    idx = [int((source[0][a] - S["anchor"][a]) / side * ncell) >= ncell // 2
           for a in range(3)]
    hot = (idx[0] * 2 + idx[1]) * 2 + idx[2]
    ranks = list(range(dec.world)) + [hot, hot]
    backends = []
    for rank in ranks:
        b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                             export_capacity=2 * npacket)
        configure(b.engine, "diffuse", int(np.prod(dec.block(rank)[1])))
        b.engine.set_sources(source, [1.], S["luminosity"])
        b.engine.set_tuning(reemit_inline_below=64, tile_min_flights=0,
                            tile_min_per_item=0)
        backends.append(b)
    group = EngineGroup([b.engine for b in backends])
    for loop in range(3):
        emitted = []
        for b in backends:
            b.reset_grid()
            b.shoot(42, loop, 0, npacket)
            emitted.append(b.get_counters()[2])
        # only the three engines of the source block fly first flights
        assert [n > 0 for n in emitted] == \
            [r == hot for r in ranks], emitted
        while group.exchange_flights(42, loop):
            pass
        group.reduce_accumulators()
        tw, tc = 0., np.zeros(4)
        for b in backends:
            b.synchronize()
            t, c, n = b.get_counters()
            tw += t
            tc += np.asarray(c)
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, loop, 0, npacket)
        assert tw == sim.totweight == npacket
        assert np.array_equal(tc, sim.typecount)
        J = assemble(dec, backends[:dec.world], E.FIELD_MEAN_INTENSITY)
        assert np.allclose(J, sim.J[0], rtol=1e-9,
                           atol=1e-12 * sim.J[0].max()), loop
        copies = [backends[i] for i, r in enumerate(ranks) if r == hot]
        J0 = copies[0].engine.download_field(E.FIELD_MEAN_INTENSITY)
        for c in copies[1:]:
            assert np.array_equal(
                c.engine.download_field(E.FIELD_MEAN_INTENSITY), J0)
        for b in backends:
            b.update_cells(loop, tw)
        # (a device fault is reported by the device call, not inside the
        # oracle's long CPU section)
        for b in backends:
            b.engine.synchronize()
        sim.update(loop, sim.totweight)
        x = assemble(dec, backends[:dec.world], E.FIELD_IONIC_FRACTION)
        assert np.allclose(x, sim.x[0], rtol=1e-8)
        x0 = copies[0].engine.download_field(E.FIELD_IONIC_FRACTION)
        for c in copies[1:]:
            assert np.array_equal(
                c.engine.download_field(E.FIELD_IONIC_FRACTION), x0)
        # keep in lockstep with the oracle: its sums have another order, and
        # 1e-16 in x moves an absorption across a cell wall now and then
        shape = (ncell,) * 3
        for rank, b in zip(ranks, backends):
            off, size = dec.block(rank)
            sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
            b.engine.upload_cells(
                np.asarray(sim.number_density).reshape(shape)[sl].ravel(),
                np.asarray(sim.temperature).reshape(shape)[sl].ravel(),
                np.array([np.asarray(x_).reshape(shape)[sl].ravel()
                          for x_ in sim.x]))
    # every copy did a share of the block's work
    steps = [c.get_counters()[2] for c in copies]
    assert min(steps) > 0.25 * max(steps), steps
    group.close()
    for b in backends:
        b.engine.close()


def test_group_reduce_goes_through_rccl(monkeypatch):
    """The accumulator reduce of a replica group is a grouped ncclAllReduce
    (RCCL, loaded at run time). This box has one GPU, and RCCL refuses two
    ranks on one device, so the collective runs here with a group of ONE
    engine (CMI_GPU_FORCE_RCCL): library loading, communicator, data type and
    reduction codes, in-place call on the engine's stream - a one-rank sum
    must leave the accumulators as they are."""
    from cmacionize_amd import GpuEngine, STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.engine import EngineGroup
    monkeypatch.setenv("CMI_GPU_FORCE_RCCL", "1")
    n = 12
    eng = GpuEngine((n,) * 3, S["anchor"], S["sides"], (0, 0, 0), device=0,
                    track_heating=True)
    configure(eng, "stromgren", n ** 3)
    eng.reset_grid()
    eng.shoot(3, 0, 0, 5000)
    before = [eng.download_field(E.FIELD_MEAN_INTENSITY),
              eng.download_field(E.FIELD_HEATING)]
    group = EngineGroup([eng])
    group.reduce_accumulators()
    eng.synchronize()
    after = [eng.download_field(E.FIELD_MEAN_INTENSITY),
             eng.download_field(E.FIELD_HEATING)]
    assert before[0].max() > 0.
    for a, b in zip(after, before):
        assert np.array_equal(a, b)
    # ... and the gather of the sharded cell update: grouped in-place
    # ncclAllGathers of the 15 state fields and the transport records; with
    # one rank they must leave what the update wrote
    twin = GpuEngine((n,) * 3, S["anchor"], S["sides"], (0, 0, 0), device=0,
                     track_heating=True)
    configure(twin, "stromgren", n ** 3)
    twin.upload_field(E.FIELD_MEAN_INTENSITY, before[0])
    twin.update_cells(0, 5000.)
    group.update_cells(0, 5000.)
    eng.synchronize()
    assert np.array_equal(eng.download_field(E.FIELD_IONIC_FRACTION),
                          twin.download_field(E.FIELD_IONIC_FRACTION))
    for e in (eng, twin):
        e.reset_grid()
        e.shoot(3, 1, 0, 5000)
    ref = twin.download_field(E.FIELD_MEAN_INTENSITY)
    assert np.allclose(eng.download_field(E.FIELD_MEAN_INTENSITY), ref,
                       rtol=1e-11, atol=1e-14 * ref.max())
    group.close()
    eng.close()
    twin.close()


def test_group_sharded_cell_update():
    """cmi_gpu_group_update_cells on three replicas of a 15^3 lexington grid
    (3375 cells: equal slabs; 4 replicas of it: unequal slabs): member r
    solves slab r - ionization balance and, from loop 4, the temperature
    solve - and pulls the other slabs and their transport records. Every
    replica ends bit-identical to a single engine that updated all cells
    (src/IonizationSimulation.cpp:532-618), also in what the next transport
    step deposits."""
    from cmacionize_amd import GpuEngine, STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.engine import EngineGroup
    n, npacket = 15, 30000
    dens, temp = lexington_fields(n)
    for world in (3, 4):
        engines = []
        for k in range(world + 1):
            e = GpuEngine((n,) * 3, S["anchor"], S["sides"], (0, 0, 0),
                          device=0, track_heating=True)
            configure(e, "lexington", n ** 3, dens.ravel(), temp.ravel())
            engines.append(e)
        single, replicas = engines[0], engines[1:]
        group = EngineGroup(replicas)
        for loop in (0, 4, 5):
            single.reset_grid()
            single.shoot(7, loop, 0, npacket)
            tw = single.get_counters()[0]
            first = 0
            for r, e in enumerate(replicas):
                count = npacket // world + (1 if r < npacket % world else 0)
                e.reset_grid()
                e.shoot(7, loop, first, count)
                first += count
            group.reduce_accumulators()
            # same integrals everywhere (the sums differ in order only):
            # start all engines from the single engine's
            for f in range(16):
                J = single.download_field(E.FIELD_MEAN_INTENSITY + f)
                for e in replicas:
                    e.upload_field(E.FIELD_MEAN_INTENSITY + f, J)
            single.update_cells(loop, tw)
            group.update_cells(loop, tw)
            for e in replicas:
                e.synchronize()
                for f in range(15):
                    a = e.download_field(E.FIELD_TEMPERATURE + f)
                    b = single.download_field(E.FIELD_TEMPERATURE + f)
                    assert np.array_equal(a, b, equal_nan=True), (world, loop, f)
        T = single.download_field(E.FIELD_TEMPERATURE)
        assert T.max() > 6000.  # the temperature was solved
        # the transport records were gathered too: same next step
        for e in engines:
            e.reset_grid()
            e.shoot(7, 6, 0, npacket)
        ref = single.download_field(E.FIELD_MEAN_INTENSITY)
        for e in replicas:
            assert np.allclose(e.download_field(E.FIELD_MEAN_INTENSITY), ref,
                               rtol=1e-11, atol=1e-14 * ref.max())
        group.close()
        for e in engines:
            e.close()
