"""Continuous photon sources: IsotropicContinuousPhotonSource
on the simulation box, the mix with the discrete sources and the packet
weights that come with it (src/PhotonSource.cpp:104-130,208-249,
src/IsotropicContinuousPhotonSource.hpp:95-191).

CPU part: the oracle's restatement against the invariants the reference's
testIsotropicContinuousPhotonSource.cpp checks (origin on the box, direction
into it) and the PhotonSource ctor's rules. GPU part: the engine against the
oracle on the same seeds - emission, tallies that are sums of weights."""
import ctypes as C

import numpy as np
import pytest

FREQ_C = 4.2e15


def emit(oracle, sim, seed, n):
    ph = oracle.Photon()
    t = C.c_double()
    out = []
    for i in range(n):
        oracle.lib().cmio_emit(C.byref(sim.model), seed, 0, i, C.byref(ph),
                               C.byref(t), None)
        out.append((list(ph.position), list(ph.direction), ph.energy,
                    ph.weight, t.value))
    return out


def test_source_mix_rules(oracle):
    """PhotonSource ctor, src/PhotonSource.cpp:104-130."""
    from cmacionize_amd import STROMGREN as S
    sim = oracle.stromgren_simulation(8)
    m = sim.model
    assert m.continuous_probability == 0.
    # both kinds: half of the packets each, continuous weight Lc / Ld
    sim.set_continuous_source(3. * S["luminosity"], frequency=FREQ_C)
    assert m.continuous_probability == 0.5
    assert m.discrete_photon_weight == 1.
    assert m.continuous_photon_weight == 3.
    assert m.total_luminosity == 4. * S["luminosity"]
    # continuous only
    m.nsource = 0
    oracle.lib().cmio_mix_sources(C.byref(m))
    assert m.continuous_probability == 1.
    assert (m.discrete_photon_weight, m.continuous_photon_weight) == (0., 1.)
    assert m.total_luminosity == 3. * S["luminosity"]


def test_isotropic_source_starts_on_the_box_and_points_inwards(oracle):
    """test/testIsotropicContinuousPhotonSource.cpp: every packet starts on
    the box (inside it: upper faces exclusive) and flies into it; the six
    faces of a cube get the same share."""
    from cmacionize_amd import STROMGREN as S
    sim = oracle.stromgren_simulation(8)
    sim.model.nsource = 0
    sim.set_continuous_source(1.e49, frequency=FREQ_C)
    lo = np.array(S["anchor"])
    side = np.array(S["sides"])
    hi = lo + side
    faces = np.zeros(6)
    n = 6000
    for pos, dirn, nu, w, tau in emit(oracle, sim, 5, n):
        pos, dirn = np.array(pos), np.array(dirn)
        assert nu == FREQ_C and w == 1.
        assert np.all(pos >= lo) and np.all(pos < hi)
        # (focus + l * direction lands on the face up to rounding)
        on_lo = np.abs(pos - lo) <= 4 * np.finfo(float).eps * side
        on_hi = np.abs(pos - hi) <= 4 * np.finfo(float).eps * side
        assert on_lo.sum() + on_hi.sum() >= 1
        for a in range(3):
            if on_lo[a]:
                assert dirn[a] > 0.
                faces[2 * a] += 1
            if on_hi[a]:
                assert dirn[a] < 0.
                faces[2 * a + 1] += 1
        assert abs(np.dot(dirn, dirn) - 1.) < 1e-14
    assert np.all(np.abs(faces / n - 1. / 6.) < 5. * np.sqrt(1. / 6. / n))


def continuous_pair(oracle, ncell, discrete, planck, reemit):
    """(engine, oracle simulation) with an isotropic continuous source and,
    optionally, the benchmark's star."""
    from cmacionize_amd import GpuEngine, STROMGREN as S
    from cmacionize_amd import engine as E
    eng = GpuEngine((ncell,) * 3, S["anchor"], S["sides"], (0, 0, 0),
                    device=0, track_heating=True)
    sim = oracle.OracleSimulation((ncell,) * 3, S["anchor"], S["sides"])
    sim.set_homogeneous(S["density"], S["temperature"], xH=3.e-5)
    m = sim.model
    m.spectrum_type = oracle.SPECTRUM_MONOCHROMATIC
    m.mono_frequency = S["frequency"]
    m.xsec_type = oracle.XSEC_FIXED
    m.xsec_fixed[0] = S["sigma_H"]
    m.recomb_type = oracle.RECOMB_FIXED
    m.recomb_fixed[0] = S["alpha_H"]
    sigma = np.zeros(14)
    sigma[0] = S["sigma_H"]
    alpha = np.zeros(14)
    alpha[0] = S["alpha_H"]
    eng.set_cross_sections_fixed(sigma)
    eng.set_recombination_rates_fixed(alpha)
    Lc = 2.5 * S["luminosity"]
    if discrete:
        pos = [[0.2 * S["sides"][0], -0.1 * S["sides"][0], 0.05 * S["sides"][0]]]
        sim.set_sources(pos, [1.], S["luminosity"])
        eng.set_sources(pos, [1.], S["luminosity"])
        eng.set_spectrum_monochromatic(S["frequency"])
    else:
        eng.set_sources([], [], 0.)
    if planck:
        sim.set_continuous_source(Lc, planck_temperature=45000.)
        eng.set_continuous_spectrum_planck(45000.)
    else:
        sim.set_continuous_source(Lc, frequency=FREQ_C)
        eng.set_continuous_spectrum_monochromatic(FREQ_C)
    eng.set_continuous_source(E.CONTINUOUS_ISOTROPIC, Lc)
    if reemit:
        m.reemit_type = oracle.REEMIT_PHYSICAL
        eng.set_reemission(1)
    eng.upload_cells(sim.number_density, sim.temperature,
                     np.array([np.asarray(x) for x in sim.x]))
    return eng, sim


@pytest.mark.gpu
@pytest.mark.parametrize("discrete,planck", [(True, False), (False, True)])
def test_continuous_emission_matches_oracle(oracle, discrete, planck):
    """cmi_gpu_emit_packets: same stream -> same packets, from either kind of
    source."""
    eng, sim = continuous_pair(oracle, 8, discrete, planck, False)
    if planck:
        sim.build_tables()
    n = 2048
    pos, dirn, nu, sig, tau = eng.emit_packets(11, 0, 0, n)
    ref = emit(oracle, sim, 11, n)
    kinds = set()
    for i, (p, d, f, w, t) in enumerate(ref):
        assert np.allclose(dirn[i], d, rtol=0, atol=4e-16)
        # (the entry point: focus + l * direction, with the direction's ulps)
        assert np.allclose(pos[i], p, rtol=0, atol=1e-15 * sim.grid.sides[0])
        assert abs(nu[i] - f) <= 1e-15 * f
        assert abs(tau[i] - t) <= 4e-16 * abs(t)
        kinds.add(w)
    assert kinds == ({1., 2.5} if discrete else {1.})
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tuning", [
    dict(tile_rounds=0),
    dict(tile_rounds=1, tile_min_flights=0, tile_min_per_item=0),
    dict(reemit_passes=0),
])
@pytest.mark.parametrize("discrete", [True, False])
def test_weighted_tallies_match_oracle(oracle, discrete, tuning):
    """A star plus an isotropic background 2.5 times as luminous: packets of
    weight 1 and 2.5. Mean intensity, heating, totweight and the per-type
    counts are sums of weights (src/DensityGrid.hpp:150-197,
    src/IonizationPhotonShootJob.hpp:143-144), re-emitted packets keep their
    weight - against the oracle; then the cell update with
    L_discrete + L_continuous."""
    from cmacionize_amd import engine as E
    ncell, npacket = 20, 40000
    eng, sim = continuous_pair(oracle, ncell, discrete, False, True)
    eng.set_tuning(reemit_inline_below=64, **tuning)
    for loop in range(2):
        eng.reset_grid()
        eng.shoot(21, loop, 0, npacket)
        tw, tc, ns = eng.get_counters()
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(21, loop, 0, npacket)
        assert abs(tw - sim.totweight) <= 1e-12 * sim.totweight
        assert np.allclose(tc, sim.typecount, rtol=1e-12, atol=0.)
        if discrete:
            assert tw > 1.5 * npacket  # about (1 + 2.5) / 2 per packet
        else:
            assert tw == npacket
        assert tc[1] > 0 and tc[3] > 0
        J = eng.download_field(E.FIELD_MEAN_INTENSITY)
        assert np.allclose(J, sim.J[0], rtol=1e-9, atol=1e-12 * sim.J[0].max())
        h = eng.download_field(E.FIELD_HEATING)
        assert np.allclose(h, sim.heating[0], rtol=1e-9,
                           atol=1e-12 * np.abs(sim.heating[0]).max())
        eng.upload_field(E.FIELD_MEAN_INTENSITY, sim.J[0])
        eng.update_cells(loop, sim.totweight)
        eng.synchronize()  # device errors surface here, not in the oracle
        sim.update(loop, sim.totweight)
        assert np.array_equal(eng.download_field(E.FIELD_IONIC_FRACTION),
                              sim.x[0])
    eng.close()


@pytest.mark.gpu
def test_continuous_source_on_a_decomposed_grid(oracle):
    """Packets of the continuous source enter through every face of the box:
    every block emits those that start in it."""
    from cmacionize_amd import STROMGREN as S
    from cmacionize_amd import engine as E
    from cmacionize_amd.engine import EngineGroup
    from cmacionize_amd.simulation import DomainDecomposition, DomainGpuBackend
    ncell, npacket = 24, 30000
    eng, sim = continuous_pair(oracle, ncell, True, False, True)
    eng.close()
    dec = DomainDecomposition((ncell,) * 3, (2, 1, 2))
    backends = []
    Lc = 2.5 * S["luminosity"]
    pos = [[0.2 * S["sides"][0], -0.1 * S["sides"][0], 0.05 * S["sides"][0]]]
    shape = (ncell,) * 3
    for rank in range(dec.world):
        b = DomainGpuBackend(dec, rank, S["anchor"], S["sides"], device=0,
                             track_heating=True, export_capacity=4 * npacket)
        e = b.engine
        sigma = np.zeros(14)
        sigma[0] = S["sigma_H"]
        alpha = np.zeros(14)
        alpha[0] = S["alpha_H"]
        e.set_cross_sections_fixed(sigma)
        e.set_recombination_rates_fixed(alpha)
        e.set_sources(pos, [1.], S["luminosity"])
        e.set_spectrum_monochromatic(S["frequency"])
        e.set_continuous_spectrum_monochromatic(FREQ_C)
        e.set_continuous_source(E.CONTINUOUS_ISOTROPIC, Lc)
        e.set_reemission(1)
        e.set_tuning(reemit_inline_below=64, tile_min_flights=0,
                     tile_min_per_item=0)
        off, size = dec.block(rank)
        sl = tuple(slice(off[a], off[a] + size[a]) for a in range(3))
        e.upload_cells(
            np.asarray(sim.number_density).reshape(shape)[sl].ravel(),
            np.asarray(sim.temperature).reshape(shape)[sl].ravel(),
            np.array([np.asarray(x).reshape(shape)[sl].ravel()
                      for x in sim.x]))
        backends.append(b)
    group = EngineGroup([b.engine for b in backends])
    for b in backends:
        b.reset_grid()
        b.shoot(21, 0, 0, npacket)
    while group.exchange_flights(21, 0):
        pass
    tw, tc = 0., np.zeros(4)
    J = np.zeros(shape)
    for rank, b in enumerate(backends):
        b.synchronize()
        t, c, n = b.get_counters()
        tw += t
        tc += np.asarray(c)
        off, size = dec.block(rank)
        J[off[0]:off[0] + size[0], off[1]:off[1] + size[1],
          off[2]:off[2] + size[2]] = \
            b.engine.download_field(E.FIELD_MEAN_INTENSITY).reshape(size)
    sim.reset()
    sim.totweight = 0.
    sim.typecount[:] = 0.
    sim.shoot(21, 0, 0, npacket)
    assert abs(tw - sim.totweight) <= 1e-12 * sim.totweight
    assert np.allclose(tc, sim.typecount, rtol=1e-12, atol=0.)
    assert np.allclose(J.ravel(), sim.J[0], rtol=1e-9,
                       atol=1e-12 * sim.J[0].max())
    group.close()
    for b in backends:
        b.engine.close()


# ---------------------------------------------------------------------------
# PlanarContinuousPhotonSource (src/PlanarContinuousPhotonSource.hpp:96-196)
# ---------------------------------------------------------------------------

def planar_setup(oracle, sim, axis):
    from cmacionize_amd import STROMGREN as S
    side = S["sides"][0]
    intercept = 0.125 * side          # on a cell wall of a 16^3 / 24^3 grid
    anchor = (-0.3 * side, -0.2 * side)
    sides = (0.5 * side, 0.35 * side)
    L = 1.5 * S["luminosity"]
    sim.set_planar_continuous_source(axis, intercept, anchor, sides, L, FREQ_C)
    return intercept, anchor, sides, L


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_planar_source_starts_on_its_rectangle(oracle, axis):
    """Every packet starts on the rectangle of the plane x[axis] = intercept
    (the two other axes in their natural order, get_non_fixed_index, :69-76)
    in an isotropic direction; the source has its own luminosity, so with a
    star of luminosity L the packets weigh 1.5 L / L."""
    sim = oracle.stromgren_simulation(8)
    intercept, anchor, sides, L = planar_setup(oracle, sim, axis)
    assert sim.model.continuous_probability == 0.5
    assert sim.model.continuous_photon_weight == 1.5
    others = [a for a in range(3) if a != axis]
    n = 4000
    dirs = []
    seen = 0
    for pos, dirn, nu, w, tau in emit(oracle, sim, 3, n):
        if w == 1.:
            continue  # a packet of the star
        seen += 1
        assert w == 1.5 and nu == FREQ_C
        assert pos[axis] == intercept
        for k, a in enumerate(others):
            assert anchor[k] <= pos[a] <= anchor[k] + sides[k]
        dirs.append(dirn)
    dirs = np.array(dirs)
    assert abs(seen / n - 0.5) < 4. * np.sqrt(0.25 / n)
    assert np.all(np.abs(dirs.mean(axis=0)) < 5. / np.sqrt(seen))


@pytest.mark.gpu
@pytest.mark.parametrize("axis", [0, 2])
def test_planar_source_matches_oracle(oracle, axis):
    """Emission (same stream -> same packets) and one transport step with a
    star and a planar source, tile rounds on: weighted tallies against the
    oracle."""
    from cmacionize_amd import GpuEngine, STROMGREN as S
    from cmacionize_amd import engine as E
    ncell, npacket = 16, 30000
    eng = GpuEngine((ncell,) * 3, S["anchor"], S["sides"], (0, 0, 0),
                    device=0, track_heating=True)
    sim = oracle.OracleSimulation((ncell,) * 3, S["anchor"], S["sides"])
    sim.set_homogeneous(S["density"], S["temperature"], xH=3.e-5)
    m = sim.model
    m.spectrum_type = oracle.SPECTRUM_MONOCHROMATIC
    m.mono_frequency = S["frequency"]
    m.xsec_type = oracle.XSEC_FIXED
    m.xsec_fixed[0] = S["sigma_H"]
    m.recomb_type = oracle.RECOMB_FIXED
    m.recomb_fixed[0] = S["alpha_H"]
    m.reemit_type = oracle.REEMIT_PHYSICAL
    star = [[0.2 * S["sides"][0], -0.1 * S["sides"][0], 0.05 * S["sides"][0]]]
    sim.set_sources(star, [1.], S["luminosity"])
    intercept, anchor, sides, L = planar_setup(oracle, sim, axis)
    sigma = np.zeros(14)
    sigma[0] = S["sigma_H"]
    alpha = np.zeros(14)
    alpha[0] = S["alpha_H"]
    eng.set_cross_sections_fixed(sigma)
    eng.set_recombination_rates_fixed(alpha)
    eng.set_sources(star, [1.], S["luminosity"])
    eng.set_spectrum_monochromatic(S["frequency"])
    eng.set_continuous_spectrum_monochromatic(FREQ_C)
    eng.set_continuous_source_planar(axis, intercept, anchor, sides, L)
    eng.set_reemission(1)
    eng.set_tuning(reemit_inline_below=64, tile_min_flights=0,
                   tile_min_per_item=0)
    eng.upload_cells(sim.number_density, sim.temperature,
                     np.array([np.asarray(x) for x in sim.x]))
    pos, dirn, nu, sig, tau = eng.emit_packets(5, 0, 0, 1024)
    for i, (p, d, f, w, t) in enumerate(emit(oracle, sim, 5, 1024)):
        assert np.array_equal(pos[i], p)
        assert np.allclose(dirn[i], d, rtol=0, atol=4e-16)
        assert nu[i] == f
    eng.reset_grid()
    eng.shoot(5, 1, 0, npacket)
    tw, tc, ns = eng.get_counters()
    sim.reset()
    sim.totweight = 0.
    sim.typecount[:] = 0.
    sim.shoot(5, 1, 0, npacket)
    assert abs(tw - sim.totweight) <= 1e-12 * sim.totweight
    assert np.allclose(tc, sim.typecount, rtol=1e-12, atol=0.)
    assert tw > 1.2 * npacket
    J = eng.download_field(E.FIELD_MEAN_INTENSITY)
    assert np.allclose(J, sim.J[0], rtol=1e-9, atol=1e-12 * sim.J[0].max())
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tiles", [0, 1])
def test_weighted_tallies_multi_ion(oracle, tiles):
    """lexingtonHII40's physics (Verner cross sections, 14 ions, both heating
    terms, physical re-emission) with its star PLUS an isotropic Planck
    background 1.7 times as luminous: packets of weight 1 and 1.7 through the
    multi-ion kernels (transposed walk, combining table, tile rounds). All 16
    accumulator fields and the weighted counters against the oracle."""
    from cmacionize_amd import GpuEngine, STROMGREN as S
    from cmacionize_amd import engine as E
    from test_gpu_domain import configure, lexington_fields
    ncell, npacket = 16, 40000
    sim = oracle.lexington_simulation(ncell)
    Lc = 1.7 * 4.26e49
    sim.model.discrete_luminosity = 4.26e49
    sim.set_continuous_source(Lc, planck_temperature=30000.)
    sim.build_tables()
    assert sim.model.continuous_photon_weight == 1.7
    eng = GpuEngine((ncell,) * 3, S["anchor"], S["sides"], (0, 0, 0),
                    device=0, track_heating=True)
    configure(eng, "lexington", ncell ** 3,
              np.asarray(sim.number_density), np.asarray(sim.temperature))
    eng.set_continuous_spectrum_planck(30000.)
    eng.set_continuous_source(E.CONTINUOUS_ISOTROPIC, Lc)
    eng.set_tuning(tile_rounds=tiles, tile_min_flights=0, tile_min_per_item=0,
                   reemit_inline_below=64)
    eng.upload_cells(sim.number_density, sim.temperature,
                     np.array([np.asarray(x) for x in sim.x]))
    for loop in range(2):
        eng.reset_grid()
        eng.shoot(8, loop, 0, npacket)
        tw, tc, ns = eng.get_counters()
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(8, loop, 0, npacket)
        assert abs(tw - sim.totweight) <= 1e-9 * sim.totweight
        assert tw > 1.2 * npacket
        # (a frequency within an ulp of a threshold may fall on the other
        # side on the device: up to a few packets change type)
        assert np.abs(np.asarray(tc) - sim.typecount).max() <= 3 * 1.7
        for ion in range(14):
            J = eng.download_field(E.FIELD_MEAN_INTENSITY + ion)
            ref = np.asarray(sim.J[ion])
            assert np.allclose(J, ref, rtol=1e-6, atol=1e-6 * ref.max()), ion
        for k in range(2):
            h = eng.download_field(E.FIELD_HEATING + k)
            ref = np.asarray(sim.heating[k])
            assert np.allclose(h, ref, rtol=1e-6,
                               atol=1e-6 * np.abs(ref).max())
        for f in range(16):
            eng.upload_field(E.FIELD_MEAN_INTENSITY + f,
                             np.asarray(sim.J[f]) if f < 14
                             else np.asarray(sim.heating[f - 14]))
        eng.update_cells(loop, sim.totweight)
        eng.synchronize()  # device errors surface here, not in the oracle
        sim.update(loop, sim.totweight)
        x = eng.download_field(E.FIELD_IONIC_FRACTION)
        assert np.allclose(x, sim.x[0], rtol=1e-6)
    eng.close()
