"""GPU parity tests of the sampled spectra, diffuse re-emission, multi-ion
transport, line cooling and thermal balance - through the C ABI, against the
reference's fixtures (at the reference's tolerances) and against the oracle."""
import ctypes as C

import numpy as np
import pytest

from test_oracle_pinning import load, rel_ok
from test_oracle_physics import LEX, lexington_model, p

pytestmark = pytest.mark.gpu


def lexington_engine(ncell, oracle_sim=None, do_temperature=True,
                     track_heating=True):
    from cmacionize_amd import GpuEngine
    import oracle_lib as o
    anchor = (-5. * o.PC,) * 3
    sides = (10. * o.PC,) * 3
    eng = GpuEngine((ncell,) * 3, anchor, sides, (0, 0, 0), device=0,
                    track_heating=track_heating)
    eng.set_sources([[0., 0., 0.]], [1.], 4.26e49)
    eng.set_spectrum_planck(40000.)
    eng.set_cross_sections_verner()
    eng.set_recombination_rates_verner()
    eng.set_abundances(LEX[1:])
    eng.set_reemission(1)
    eng.set_temperature_params(do_temperature_calculation=int(do_temperature),
                               pah_heating_factor=0.)
    if oracle_sim is not None:
        eng.upload_cells(oracle_sim.number_density, oracle_sim.temperature,
                         np.array([np.asarray(x) for x in oracle_sim.x]))
    return eng


def test_spectrum_samplers_match_oracle(oracle):
    """Same uniforms -> same frequencies (log10 / pow differ by ulps)."""
    eng = lexington_engine(4)
    m = oracle.Model()
    m.spectrum_type = oracle.SPECTRUM_PLANCK
    m.planck_temperature = 40000.
    m.xsec_type = oracle.XSEC_VERNER
    m.tables = oracle.lib().cmio_tables_create(C.byref(m))
    n = 200000
    # (temperatures inside the table of the Lyman continuum spectra, in its
    # first and last intervals and outside it on either side)
    for kind, T in ((0, 0.), (1, 8888.), (1, 1600.), (1, 14990.), (2, 8888.),
                    (3, 0.), (1, 1000.), (2, 1567.5), (1, 14932.5),
                    (2, 20000.)):
        got = eng.sample_spectrum(kind, T, 77, n)
        want = np.empty(n)
        oracle.lib().cmio_sample_spectrum(C.byref(m), kind, T, 77, n, p(want))
        assert np.allclose(got, want, rtol=1e-13, atol=0.), kind
    oracle.lib().cmio_tables_free(m.tables)
    eng.close()


def test_planck_histogram_on_gpu():
    """testPhotonSourceSpectrum.cpp:155-190 on the device sampler."""
    from test_oracle_physics import planck_luminosity
    eng = lexington_engine(4)
    x = eng.sample_spectrum(0, 0., 5, 1000000) / 3.288465385e15
    counts = np.bincount(((x - 1.) * 100. / 3.).astype(int), minlength=100)
    enorm = planck_luminosity(1.015) / counts[0]
    for i in range(100):
        nu = 1. + (i + 0.5) * 0.03
        tol = 1.5 * 10. ** (-2.29 + 0.0239001 * (i - 3.))
        assert rel_ok(planck_luminosity(nu), counts[i] * enorm, tol), i
    eng.close()


def test_cooling_and_heating_balance_fixture(oracle):
    """ioneng_testdata.txt through the device balance (reference tolerance
    1e-6) and against the oracle."""
    data = load("ioneng_testdata.txt")
    eng = lexington_engine(4)
    eng.set_temperature_params(do_temperature_calculation=1,
                               pah_heating_factor=1.,
                               cosmic_ray_heating_factor=0.)
    J = data[:, :14]
    heating = data[:, 14:16] * 1.e-7
    T = data[:, 16]
    n = data[:, 19] * 1.e6
    x, _, pair = eng.thermal_probe(0, J, heating, T, n)
    m = lexington_model(oracle)
    for i, row in enumerate(data):
        assert rel_ok(x[i, 0], row[20], 1.e-6) and rel_ok(x[i, 1], row[21], 1.e-6)
        assert rel_ok(pair[i, 0], row[17] * 0.1 * 1.e-20, 1.e-6)
        assert rel_ok(pair[i, 1], row[18] * 0.1 * 1.e-20, 1.e-6)
        for k in range(12):
            assert rel_ok(x[i, 2 + k], row[22 + k], 1.e-6)
        h0, he0, gain, loss = (C.c_double() for _ in range(4))
        xo = np.zeros(14)
        oracle.lib().cmio_cooling_and_heating_balance(
            C.byref(m), C.byref(h0), C.byref(he0), C.byref(gain),
            C.byref(loss), T[i], n[i], 0.5, p(np.ascontiguousarray(J[i])),
            p(np.ascontiguousarray(heating[i])), 1., 0., 0.75, p(xo))
        assert abs(pair[i, 0] - gain.value) <= 1e-10 * gain.value
        assert abs(pair[i, 1] - loss.value) <= 1e-10 * loss.value
        # device pow/exp/log differ from libm by ulps; the fractions are
        # ratios of sums of such terms (some rows cancel strongly)
        assert np.allclose(x[i, 2:], xo[2:], rtol=1e-8, atol=1e-14)
    eng.close()


def test_temperature_solve_fixture(oracle):
    """tbal_testdata.txt through the device temperature solve (reference
    tolerance 1e-4) and against the oracle."""
    data = load("tbal_testdata.txt")
    data = data[data[:, 16] <= 30000.]
    eng = lexington_engine(4)
    eng.set_temperature_params(do_temperature_calculation=1,
                               pah_heating_factor=1.,
                               cosmic_ray_heating_limit=1.,
                               cosmic_ray_heating_scale_length=0.)
    J = data[:, :14]
    heating = data[:, 14:16] * 1.e-7
    T = data[:, 16]
    n = data[:, 17] * 1.e6
    x, Tnew, hout = eng.thermal_probe(1, J, heating, T, n)
    m = lexington_model(oracle, pahfac=1.)
    for i, row in enumerate(data):
        expect = row[18:32].copy()
        expect[0] = min(1., expect[0])
        for k in range(14):
            assert rel_ok(x[i, k], expect[k], 1.e-4), (i, k)
        assert rel_ok(Tnew[i], min(30000., row[32]), 1.e-4)
        xo = np.zeros(14)
        Tc = C.c_double(T[i])
        ho = np.ascontiguousarray(heating[i]).copy()
        oracle.lib().cmio_temperature_cell(
            C.byref(m), 1., 1., n[i], 0.5, C.byref(Tc),
            p(np.ascontiguousarray(J[i])), p(ho), p(xo))
        assert abs(Tnew[i] - Tc.value) <= 1e-7 * Tc.value
        assert np.allclose(x[i], xo, rtol=1e-6, atol=1e-300)
    eng.close()


@pytest.mark.parametrize("tuning", [
    dict(reemit_passes=1),
    dict(reemit_passes=0),
    # later generations in tile rounds, down to the last flight / with a
    # hand-over to the transport kernel / refilling lane by lane
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0),
    dict(tile_rounds=1, tile_min_flights=700, tile_min_per_item=0),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0,
         tile_counting_sort=0),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=3,
         reemit_inline_below=0),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0, tile_refill_threshold=1,
         max_packets_per_launch=25000),
    dict(tile_rounds=0),
    dict(reemit_passes=1, reemit_inline_below=0, reemit_max_passes=3,
         refill_threshold_reemit=1, max_packets_per_launch=25000),
    dict(reemit_passes=1, exact_dda=1, aggregate=0, sort_packets=0),
    # round 5: absorbed packets parked through the queue's counter instead of
    # at their position; parked at their position with lane-by-lane refills
    # and split launches; the rows of the live flights copied into fresh rows
    # whenever there is a dead slot (default for these runs: never)
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0,
         park_in_place=0),
    dict(tile_rounds=1, tile_min_flights=500, tile_min_per_item=0,
         refill_threshold=5, max_packets_per_launch=17000),
    dict(tile_rounds=0, refill_threshold=20, sort_packets=0),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0,
         tile_compact_ratio=1),
])
def test_diffuse_stromgren_shoot_matches_oracle(oracle, tuning):
    """benchmarks/stromgren_diffuse.param: physical re-emission with fixed
    cross sections (H branch only) - packets, counters and J against the
    oracle on the same seeds."""
    from cmacionize_amd import engine as E
    from test_gpu_transport import make_engine
    ncell, npacket = 32, 60000
    eng = make_engine(ncell)
    eng.set_reemission(1)
    eng.set_tuning(**tuning)
    sim = oracle.stromgren_simulation(ncell, diffuse=True)
    for loop in range(3):
        eng.reset_grid()
        eng.shoot(42, loop, 0, npacket)
        tw, tc, ns = eng.get_counters()
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, loop, 0, npacket)
        assert tw == sim.totweight == npacket
        # a re-emission decision is a comparison of a uniform with a
        # probability: identical unless within an ulp -> counters identical
        assert np.array_equal(tc, sim.typecount), (tc, sim.typecount)
        assert tc[1] > 0  # some packets end as diffuse H photons leaving
        J = eng.download_field(E.FIELD_MEAN_INTENSITY)
        assert np.allclose(J, sim.J[0], rtol=1e-9, atol=1e-12 * sim.J[0].max())
        eng.upload_field(E.FIELD_MEAN_INTENSITY, sim.J[0])
        eng.upload_field(E.FIELD_HEATING, sim.heating[0])
        eng.update_cells(loop, tw)
        eng.synchronize()  # device errors surface here, not in the oracle
        sim.update(loop, sim.totweight)
        assert np.array_equal(eng.download_field(E.FIELD_IONIC_FRACTION),
                              sim.x[0])
    eng.close()


@pytest.mark.parametrize("tuning", [
    dict(),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0,
         tile_counting_sort=0),
    # cross sections inside the transport / re-emission kernels instead of in
    # the key kernel / a kernel of their own
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0,
         pre_emission=0),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0,
         defer_weights=0),
    dict(pre_emission=0, defer_weights=0),
    dict(tile_rounds=0),
    # round 5: rows never copied / copied at every dead slot; absorbed
    # packets parked through the queue's counter
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0,
         tile_compact_ratio=0),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0,
         tile_compact_ratio=1, park_in_place=0),
    # range classes in the sort key, as at 1e8 packets
    dict(sort_tau_bits=3),
    dict(sort_tau_bits=2, tile_rounds=0),
])
def test_lexington_iteration_matches_oracle(oracle, tuning):
    """benchmarks/lexingtonHII40.param at 24^3: Planck source, Verner cross
    sections, 14 mean intensities + 2 heating terms, physical re-emission with
    He channels, then the cell update (ionization balance for loop <= 3,
    temperature solve after). With the later generations as tile rounds (8^3
    tiles, 16 accumulators per cell in LDS), as passes of the transport
    kernel, and the default mix."""
    from cmacionize_amd import engine as E
    ncell, npacket = 24, 40000
    sim = oracle.lexington_simulation(ncell)
    eng = lexington_engine(ncell, sim)
    if tuning:
        eng.set_tuning(**tuning)
    for loop in range(6):
        eng.reset_grid()
        eng.shoot(42, loop, 0, npacket)
        tw, tc, ns = eng.get_counters()
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, loop, 0, npacket)
        assert tw == sim.totweight == npacket
        # device pow/log10 differ from libm by ulps, so a frequency can land
        # on the other side of a threshold once in a long while
        assert np.abs(tc - sim.typecount).max() <= 3, (tc, sim.typecount)
        for ion in range(14):
            J = eng.download_field(E.FIELD_MEAN_INTENSITY + ion)
            ref = np.asarray(sim.J[ion])
            assert np.allclose(J, ref, rtol=1e-6, atol=1e-6 * ref.max()), ion
            assert abs(J.sum() - ref.sum()) <= 1e-6 * ref.sum()
        for k in range(2):
            h = eng.download_field(E.FIELD_HEATING + k)
            ref = sim.heating[k]
            assert np.allclose(h, ref, rtol=1e-6, atol=1e-6 * np.abs(ref).max())
        # cell update from identical integrals
        for ion in range(14):
            eng.upload_field(E.FIELD_MEAN_INTENSITY + ion, sim.J[ion])
        for k in range(2):
            eng.upload_field(E.FIELD_HEATING + k, sim.heating[k])
        eng.update_cells(loop, tw)
        eng.synchronize()  # device errors surface here, not in the oracle
        sim.update(loop, sim.totweight)
        T = eng.download_field(E.FIELD_TEMPERATURE)
        assert np.allclose(T, sim.temperature, rtol=1e-6, atol=0.), loop
        for ion in range(14):
            x = eng.download_field(E.FIELD_IONIC_FRACTION + ion)
            ref = np.asarray(sim.x[ion])
            ok = np.isclose(x, ref, rtol=1e-5, atol=1e-300) | \
                (np.isnan(x) & np.isnan(ref))
            assert ok.all(), (loop, ion)
        # keep both in the same state for the next iteration
        eng.upload_cells(sim.number_density, sim.temperature,
                         np.array([np.asarray(x) for x in sim.x]))
    assert sim.temperature.max() > 6000. and sim.temperature.min() == 500.
    eng.close()


@pytest.mark.parametrize("tuning", [
    dict(),                              # begin -> finish, no pipeline step
    dict(temperature_finish_slots=0),    # every step as pipeline kernels
    dict(temperature_finish_slots=100),  # pipeline steps, then the finish
    dict(temperature_pipeline=0),        # the solve as one kernel
])
def test_temperature_solve_on_eight_small_engines_at_once(oracle, tuning):
    """TemperatureCalculator::calculate_temperature
    (src/TemperatureCalculator.cpp:567-931) of eight 12^3 engines enqueued
    back to back on eight streams of one device - for a grid this small the
    pipeline is temp_begin_kernel -> read-back -> temp_finish_kernel with no
    step in between, and cmi_gpu_update_cells returns with the last kernel in
    flight - while the oracle solves the same cells on the host's cores. Every
    engine must then hold the oracle's temperatures and fractions; the
    engines' results are equal bit for bit."""
    from cmacionize_amd import engine as E
    ncell, npacket, nengine = 12, 20000, 8
    sim = oracle.lexington_simulation(ncell)
    for loop in range(4):
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, loop, 0, npacket)
        sim.update(loop, sim.totweight)
    engines = [lexington_engine(ncell, sim) for _ in range(nengine)]
    for eng in engines:
        if tuning:
            eng.set_tuning(**tuning)
    for loop in (4, 5):
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, loop, 0, npacket)
        for eng in engines:
            for ion in range(14):
                eng.upload_field(E.FIELD_MEAN_INTENSITY + ion, sim.J[ion])
            for k in range(2):
                eng.upload_field(E.FIELD_HEATING + k, sim.heating[k])
        # all eight solves in flight at once, the oracle beside them
        for eng in engines:
            eng.update_cells(loop, sim.totweight)
        sim.update(loop, sim.totweight)
        for eng in engines:
            eng.synchronize()
        T0 = engines[0].download_field(E.FIELD_TEMPERATURE)
        assert np.allclose(T0, sim.temperature, rtol=1e-6, atol=0.), loop
        x0 = []
        for ion in range(14):
            x = engines[0].download_field(E.FIELD_IONIC_FRACTION + ion)
            ref = np.asarray(sim.x[ion])
            ok = np.isclose(x, ref, rtol=1e-5, atol=1e-300) | \
                (np.isnan(x) & np.isnan(ref))
            assert ok.all(), (loop, ion)
            x0.append(x)
        for eng in engines[1:]:
            assert np.array_equal(eng.download_field(E.FIELD_TEMPERATURE), T0)
            for ion in range(14):
                assert np.array_equal(
                    eng.download_field(E.FIELD_IONIC_FRACTION + ion),
                    x0[ion], equal_nan=True), (loop, ion)
        for eng in engines:
            eng.upload_cells(sim.number_density, sim.temperature,
                             np.array([np.asarray(x) for x in sim.x]))
    assert sim.temperature.max() > 6000.
    for eng in engines:
        eng.close()


def test_lexington_converged_state_within_one_percent_of_oracle(oracle):
    """The same bar for the multi-ion benchmark: engine and oracle run 8
    iterations of lexingtonHII40 on their own (ionization balance, then the
    temperature solve from iteration 4) and must agree on shell averages of
    the temperature and of the H, He, O+ and N+ fractions within 1 %."""
    from cmacionize_amd import engine as E
    ncell, npacket, iterations = 24, 40000, 8
    sim = oracle.lexington_simulation(ncell)
    eng = lexington_engine(ncell, sim)
    for loop in range(iterations):
        eng.reset_grid()
        eng.shoot(42, loop, 0, npacket)
        tw, _, _ = eng.get_counters()
        eng.update_cells(loop, tw)
    sim.run(npacket, iterations, seed=42)
    ax = (np.arange(ncell) + 0.5) / ncell - 0.5
    r = np.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 +
                ax[None, None, :] ** 2).ravel()
    shells = np.minimum((r / 0.5 * 10).astype(int), 12)
    fields = [(E.FIELD_TEMPERATURE, sim.temperature)]
    for ion in (0, 1, 5, 8):
        fields.append((E.FIELD_IONIC_FRACTION + ion, np.asarray(sim.x[ion])))
    for field, ref in fields:
        got = eng.download_field(field)
        for s in range(shells.max() + 1):
            m = (shells == s) & np.isfinite(ref)
            if not m.any():
                continue
            a, b = got[m].mean(), ref[m].mean()
            assert abs(a - b) <= 0.01 * abs(b) + 1e-300, (field, s, a, b)
    assert np.asarray(sim.temperature).max() > 6000.
    eng.close()


@pytest.mark.parametrize("passes", [1, 0])
def test_fixed_value_reemission_matches_oracle(oracle, passes):
    """FixedValueDiffuseReemissionHandler (src/FixedValueDiffuseReemission
    Handler.hpp:73-86): re-emission with a fixed probability at a fixed
    frequency - through the interaction kernel (passes) and in place."""
    from cmacionize_amd import engine as E
    from test_gpu_transport import make_engine
    ncell, npacket = 24, 50000
    eng = make_engine(ncell, track_heating=False)
    nu = 1.01 * 3.288465385e15
    eng.set_reemission(2, 0.42, nu)
    eng.set_tuning(reemit_passes=passes, reemit_inline_below=64)
    sim = oracle.stromgren_simulation(ncell)
    sim.model.reemit_type = 2
    sim.model.reemit_fixed_probability = 0.42
    sim.model.reemit_fixed_frequency = nu
    for loop in range(3):
        eng.reset_grid()
        eng.shoot(9, loop, 0, npacket)
        tw, tc, ns = eng.get_counters()
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(9, loop, 0, npacket)
        assert tw == sim.totweight == npacket
        assert np.array_equal(tc, sim.typecount)
        assert tc[1] > 0 and tc[3] > 0
        J = eng.download_field(E.FIELD_MEAN_INTENSITY)
        assert np.allclose(J, sim.J[0], rtol=1e-9, atol=1e-12 * sim.J[0].max())
        eng.upload_field(E.FIELD_MEAN_INTENSITY, sim.J[0])
        eng.update_cells(loop, tw)
        eng.synchronize()  # device errors surface here, not in the oracle
        sim.update(loop, sim.totweight)
        assert np.array_equal(eng.download_field(E.FIELD_IONIC_FRACTION),
                              sim.x[0])
    eng.close()


@pytest.mark.parametrize("npacket", [1, 65, 700])
def test_lexington_ragged_packet_counts(oracle, npacket):
    """Multi-ion launches that do not fill a wave or a quarter of one: the
    walk's idle rows must contribute exact zeros to all 16 accumulators."""
    from cmacionize_amd import engine as E
    ncell = 12
    sim = oracle.lexington_simulation(ncell)
    eng = lexington_engine(ncell, sim)
    for loop, tuning in enumerate((dict(aggregate=2), dict(aggregate=0))):
        eng.set_tuning(**tuning)
        eng.reset_grid()
        eng.shoot(5, loop, 3, npacket)
        tw, tc, ns = eng.get_counters()
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(5, loop, 3, npacket)
        assert tw == sim.totweight == npacket
        assert np.abs(tc - sim.typecount).max() <= 1
        for ion in range(14):
            J = eng.download_field(E.FIELD_MEAN_INTENSITY + ion)
            ref = np.asarray(sim.J[ion])
            assert np.isfinite(J).all()
            assert np.allclose(J, ref, rtol=1e-6,
                               atol=1e-6 * max(ref.max(), 1e-300)), ion
        for k in range(2):
            h = eng.download_field(E.FIELD_HEATING + k)
            assert np.isfinite(h).all()
    eng.close()


@pytest.mark.parametrize("tuning", [
    dict(tile_rounds=0),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0, exact_dda=1),
])
def test_periodic_diffuse_shoot_matches_oracle(oracle, tuning):
    """A box that is periodic in x and y (CartesianDensityGrid::is_inside,
    src/CartesianDensityGrid.cpp:187-227) with physical re-emission: packets
    and their re-emissions wrap around the box, in the tile rounds across
    clipped tiles (20 cells = one 16-cell tile + one 4-cell tile per axis).
    Counters and J against the oracle on the same seeds."""
    from cmacionize_amd import GpuEngine, STROMGREN as S
    from cmacionize_amd import engine as E
    ncell, npacket = 20, 30000
    periodic = (1, 1, 0)
    eng = GpuEngine((ncell,) * 3, S["anchor"], S["sides"], periodic, device=0,
                    track_heating=True)
    source = [[0.31 * S["sides"][0], -0.2 * S["sides"][0], 0.07 * S["sides"][0]]]
    eng.set_sources(source, [1.], S["luminosity"])
    eng.set_spectrum_monochromatic(S["frequency"])
    sigma = np.zeros(14)
    sigma[0] = S["sigma_H"]
    alpha = np.zeros(14)
    alpha[0] = S["alpha_H"]
    eng.set_cross_sections_fixed(sigma)
    eng.set_recombination_rates_fixed(alpha)
    eng.set_reemission(1)
    eng.set_tuning(**tuning)
    n = ncell ** 3
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
    eng.upload_cells(sim.number_density, sim.temperature,
                     np.array([np.asarray(x) for x in sim.x]))
    for loop in range(2):
        eng.reset_grid()
        eng.shoot(9, loop, 0, npacket)
        tw, tc, ns = eng.get_counters()
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(9, loop, 0, npacket)
        assert tw == sim.totweight == npacket
        assert np.array_equal(tc, sim.typecount), (tc, sim.typecount)
        assert tc[1] > 0 and tc[3] > 0
        # the source sits 4 cells from the +x face: many paths wrap around
        assert ns > 1.5 * ncell * npacket
        J = eng.download_field(E.FIELD_MEAN_INTENSITY)
        assert np.allclose(J, sim.J[0], rtol=1e-9, atol=1e-12 * sim.J[0].max())
        h = eng.download_field(E.FIELD_HEATING)
        assert np.allclose(h, sim.heating[0], rtol=1e-9,
                           atol=1e-12 * np.abs(sim.heating[0]).max())
    eng.close()


def test_temperature_pipeline_equals_single_kernel(oracle):
    """The temperature solve as a pipeline of kernels (default) and as one
    kernel run the same functions in the same order: identical results, bit
    for bit, from the same accumulators - on a grid the engine brought to the
    iteration where the solve starts."""
    from cmacionize_amd import engine as E
    ncell, npacket = 20, 60000
    sim = oracle.lexington_simulation(ncell)
    eng = lexington_engine(ncell, sim)
    for loop in range(5):
        eng.reset_grid()
        eng.shoot(42, loop, 0, npacket)
        tw, _, _ = eng.get_counters()
        if loop < 4:
            eng.update_cells(loop, tw)
    fields = [E.FIELD_NUMBER_DENSITY, E.FIELD_TEMPERATURE] + \
        [E.FIELD_IONIC_FRACTION + i for i in range(14)] + \
        [E.FIELD_MEAN_INTENSITY + i for i in range(14)] + \
        [E.FIELD_HEATING, E.FIELD_HEATING + 1]
    before = {f: eng.download_field(f) for f in fields}
    results = []
    # the single kernel; every step through the pipeline; the pipeline with
    # its last 300 cells / all cells finished by the straggler kernel
    for pipeline, finish in ((0, 0), (1, 0), (1, 300), (1, 1 << 20)):
        for f in fields:
            eng.upload_field(f, before[f])
        eng.set_tuning(temperature_pipeline=pipeline,
                       temperature_finish_slots=finish)
        eng.update_cells(4, tw)
        results.append({f: eng.download_field(f) for f in fields})
    solved = results[0][E.FIELD_TEMPERATURE]
    assert (solved != before[E.FIELD_TEMPERATURE]).sum() > 1000
    assert 5000. < solved[solved > 600.].mean() < 15000.
    for f in fields:
        for other in results[1:]:
            assert np.array_equal(results[0][f], other[f]), f
    eng.close()


def test_lexington_benchmark_global_quantities():
    """benchmarks/lexingtonHII40.param end to end against what the
    literature gives for this model (the Lexington / Meudon HII40 benchmark,
    Ferland 1995, Pequignot et al. 2001; medians of the participating codes
    as tabulated by Ercolano et al. 2003: outer radius 1.46e19 cm,
    <He+>/<H+> 0.77, mean electron temperature ~8000 K). The reference
    repository ships no expected values for its benchmark (its
    lexingtonHII40.py only plots), so these are the one check of the
    multi-ion path - transport of 14 ions, He and H re-emission, the metal
    balance, line cooling and the thermal balance together - that does not go
    through the oracle. Bounds wider than the published codes' spread; the
    radius also follows from the photon budget: (3 Q / (4 pi n^2 alpha_B))^1/3
    = 1.49e19 cm for hydrogen alone at 8000 K, helium takes a few per cent of
    the photons."""
    from cmacionize_amd import engine as E
    import oracle_lib as o
    ncell = 64
    eng = lexington_engine(ncell)
    ax = -5. * o.PC + (np.arange(ncell) + 0.5) * (10. * o.PC / ncell)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    r = np.sqrt(X * X + Y * Y + Z * Z).ravel()
    gas = r > 3.e16
    n = ncell ** 3
    x = np.zeros((14, n))
    x[0] = 1.e-6
    x[1] = 1.e-6
    eng.upload_cells(np.where(gas, 1.e8, 0.), np.where(gas, 8000., 0.), x)
    for loop in range(14):
        eng.reset_grid()
        eng.shoot(42, loop, 0, 2000000)
        tw, _, _ = eng.get_counters()
        eng.update_cells(loop, tw)
    xH = eng.download_field(E.FIELD_IONIC_FRACTION)
    xHe = eng.download_field(E.FIELD_IONIC_FRACTION + 1)
    T = eng.download_field(E.FIELD_TEMPERATURE)
    eng.close()
    # outer radius: where the shell-mean neutral fraction passes 0.5
    edges = np.linspace(3.e16, 5. * o.PC, 61)
    which = np.digitize(r, edges) - 1
    shell = np.array([xH[(which == k) & gas].mean() for k in range(60)])
    k = int(np.argmax(shell > 0.5))
    mid = 0.5 * (edges[:-1] + edges[1:])
    r_out = np.interp(0.5, shell[k - 1:k + 1], mid[k - 1:k + 1])
    assert abs(r_out * 100. / 1.46e19 - 1.) < 0.03, r_out * 100.
    ionized = gas & (xH < 0.5)
    he_over_h = (1. - xHe[ionized]).sum() / (1. - xH[ionized]).sum()
    # (measured at 64^3, 2e6 packets: 0.70 - at the low end of the published
    # codes; the helium front is two cells wide at this resolution)
    assert 0.65 < he_over_h < 0.82, he_over_h
    # (uniform density: the volume mean is the n_e n_p weighted mean up to
    # the ionization fractions)
    t_mean = (T[ionized] * (1. - xH[ionized])).sum() / \
        (1. - xH[ionized]).sum()
    assert 7600. < t_mean < 8600., t_mean
    # the temperature rises outwards (hardening of the radiation field)
    inner = ionized & (r < 1.6 * o.PC)
    outer = ionized & (r > 3.5 * o.PC)
    assert T[inner].mean() < t_mean < T[outer].mean()
    print("lexingtonHII40 64^3: R_out %.3e cm, <He+>/<H+> %.3f, <T> %.0f K, "
          "T inner %.0f K, T outer %.0f K" % (r_out * 100., he_over_h, t_mean,
                                               T[inner].mean(),
                                               T[outer].mean()))
