"""GPU parity of the trackers (cmi_gpu_set_spectrum_trackers /
cmi_gpu_set_trackers: the
tracker hook of DensityGrid::update_integrals, src/DensityGrid.hpp:188-191,
with SpectrumTracker::count_photon, src/SpectrumTracker.hpp:176-212): the same
packets cross the same cells in engine and oracle, so the counts - integers -
must be equal."""
import numpy as np
import pytest

from test_gpu_physics import lexington_engine

pytestmark = pytest.mark.gpu


def cell_of(position, anchor, side, ncell):
    i = [int((position[a] - anchor) / side * ncell) for a in range(3)]
    return (i[0] * ncell + i[1]) * ncell + i[2]


@pytest.mark.parametrize("tuning", [dict(), dict(reemit_passes=0)])
def test_tracker_counts_match_oracle(oracle, tuning):
    ncell, npacket = 20, 40000
    sim = oracle.lexington_simulation(ncell)
    eng = lexington_engine(ncell, sim)
    eng.set_tuning(**tuning)
    pc = oracle.PC
    anchor, side = -5. * pc, 10. * pc
    positions = np.array([[1.3 * pc, 0.4 * pc, -0.7 * pc],
                          [-2.1 * pc, 1.9 * pc, 0.2 * pc],
                          [1.3 * pc, 0.4 * pc, -0.7 * pc],   # same cell, cone
                          [0.01 * pc, 0.01 * pc, 0.01 * pc],  # in the hole
                          [3.6 * pc, -3.2 * pc, 2.9 * pc]])
    angles = np.array([np.pi, np.pi, 0.5, np.pi, 1.2])
    directions = np.array([[0., 0., 0.], [0., 0., 0.], [1., 0.3, -0.5],
                           [0., 0., 0.], [1., -1., 1.]])
    nbins = 40
    eng.set_spectrum_trackers(positions, nbins, angles, directions)
    cells = [cell_of(p, anchor, side, ncell) for p in positions]
    # a first iteration without trackers (they are added for the last one);
    # then both go on from the oracle's state
    eng.reset_grid()
    eng.shoot(42, 0, 0, npacket)
    sim.run(npacket, 1, seed=42)
    assert not eng.get_tracker_counts().any()
    eng.upload_cells(sim.number_density, sim.temperature,
                     np.array([np.asarray(x) for x in sim.x]))
    eng.enable_trackers(True)
    eng.reset_grid()
    eng.shoot(42, 1, 0, npacket)
    tw, tc, ns = eng.get_counters()
    got = eng.get_tracker_counts()
    with oracle.Trackers(cells, nbins, angles, directions) as t:
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, 1, 0, npacket)
    assert tw == sim.totweight and np.array_equal(tc, sim.typecount)
    assert np.array_equal(got, t.counts)
    # primaries and re-emitted hydrogen photons were seen; the cone sees a
    # part of what the open tracker of the same cell sees; no gas, no count
    assert got[0, 0].sum() > 50 and got[0, 1].sum() > 5
    assert 0 < got[2].sum() < got[0].sum()
    assert not got[3].any()
    # switched off again: nothing more is counted, and the fast path is back
    eng.enable_trackers(False)
    eng.reset_grid()
    eng.shoot(42, 2, 0, npacket)
    assert np.array_equal(eng.get_tracker_counts(), got)
    eng.set_spectrum_trackers(np.zeros((0, 3)))
    eng.close()


def test_tracker_arguments():
    from cmacionize_amd import engine as E
    eng = lexington_engine(4)
    with pytest.raises(E.EngineError, match="not inside grid"):
        eng.set_spectrum_trackers([[1.e30, 0., 0.]])
    with pytest.raises(E.EngineError, match="at most"):
        eng.set_spectrum_trackers(np.zeros((17, 3)))
    eng.close()


@pytest.mark.parametrize("tuning", [dict(), dict(reemit_passes=0)])
def test_absorption_trackers_match_oracle(oracle, tuning):
    """AbsorptionTrackers (src/AbsorptionTracker.hpp:49-235, the hook of
    DensitySubGrid::update_intensity_counters, src/DensitySubGrid.hpp:592-617)
    beside a SpectrumTracker: path length x cross section x weight per ion and
    photon type for every packet that crosses the cell. Summed over the types
    a tracker holds the cell's own mean-intensity sums."""
    from cmacionize_amd import engine as E
    ncell, npacket = 20, 40000
    sim = oracle.lexington_simulation(ncell)
    eng = lexington_engine(ncell, sim)
    eng.set_tuning(**tuning)
    pc = oracle.PC
    anchor, side = -5. * pc, 10. * pc
    positions = np.array([[1.3 * pc, 0.4 * pc, -0.7 * pc],
                          [-2.1 * pc, 1.9 * pc, 0.2 * pc],
                          [1.3 * pc, 0.4 * pc, -0.7 * pc],   # same cell
                          [0.01 * pc, 0.01 * pc, 0.01 * pc]])  # in the hole
    kinds = [E.TRACKER_ABSORPTION, E.TRACKER_ABSORPTION, E.TRACKER_SPECTRUM,
             E.TRACKER_ABSORPTION]
    nbins = 30
    eng.set_trackers(positions, kinds, nbins)
    cells = [cell_of(p, anchor, side, ncell) for p in positions]
    sim.run(npacket, 1, seed=42)
    eng.upload_cells(sim.number_density, sim.temperature,
                     np.array([np.asarray(x) for x in sim.x]))
    eng.enable_trackers(True)
    eng.reset_grid()
    eng.shoot(42, 1, 0, npacket)
    got_counts = eng.get_tracker_counts()
    got = eng.get_tracker_absorption()
    with oracle.Trackers(cells, nbins, kinds=kinds) as t:
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, 1, 0, npacket)
    assert np.array_equal(got_counts, t.counts)
    assert got_counts[2].sum() > 50 and not got_counts[[0, 1, 3]].any()
    assert np.allclose(got, t.absorption, rtol=1e-9,
                       atol=1e-12 * t.absorption.max())
    # source photons and re-emitted hydrogen photons were absorbed; nothing
    # flies with the type "absorbed"; no gas, no absorption
    assert got[0, 0, 0] > 0. and got[0, 1, 0] > 0.
    assert not got[:, 3].any() and not got[3].any() and not got[2].any()
    # all types together: the cell's mean-intensity sums of this iteration
    for k in (0, 1):
        for ion in range(14):
            J = eng.download_field(E.FIELD_MEAN_INTENSITY + ion)[cells[k]]
            assert abs(got[k, :, ion].sum() - J) <= 1e-9 * J + 1e-300, ion
    eng.close()


def test_trackers_on_a_decomposed_grid(oracle):
    """Trackers on the blocks of a decomposed grid (TrackerManager::
    add_trackers for a DensitySubGridCreator, src/TrackerManager.hpp:243-297):
    every block is given all trackers and counts those in its own cells - in
    the incremental marcher, whose state the hand-overs carry; the caller adds
    the blocks' counts (TrackerManager::normalize, :307-318). Against the
    oracle on the undivided grid."""
    from cmacionize_amd import engine as E
    from test_gpu_domain import decomposed_backends, upload_state
    ncell, npacket = 24, 40000
    sim = oracle.lexington_simulation(ncell)
    sim.run(npacket, 1, seed=42)
    dec, backends, driver = decomposed_backends("lexington", ncell, (2, 2, 2),
                                                npacket, sim)
    upload_state(dec, backends, sim, ncell)
    pc = oracle.PC
    anchor, side = -5. * pc, 10. * pc
    positions = np.array([[1.3 * pc, 0.4 * pc, -0.7 * pc],
                          [-2.1 * pc, 1.9 * pc, 0.2 * pc],
                          [-2.1 * pc, 1.9 * pc, 0.2 * pc],
                          [3.6 * pc, -3.2 * pc, 2.9 * pc]])
    kinds = [E.TRACKER_SPECTRUM, E.TRACKER_SPECTRUM, E.TRACKER_ABSORPTION,
             E.TRACKER_ABSORPTION]
    angles = np.array([np.pi, 0.9, np.pi, np.pi])
    directions = np.array([[0., 0., 0.], [-1., 1., 0.2], [0., 0., 0.],
                           [0., 0., 0.]])
    nbins = 25
    for b in backends:
        b.engine.set_trackers(positions, kinds, nbins, angles, directions)
        b.engine.enable_trackers(True)
    driver.iteration(1, npacket, 42, update=False)
    counts = sum(b.engine.get_tracker_counts().astype(np.int64)
                 for b in backends)
    absorption = sum(b.engine.get_tracker_absorption() for b in backends)
    cells = [cell_of(p, anchor, side, ncell) for p in positions]
    with oracle.Trackers(cells, nbins, angles, directions, kinds=kinds) as t:
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, 1, 0, npacket)
    assert driver.totweight == sim.totweight
    # (the incremental marcher crosses the same cells as the exact one except
    # on exact corner ties: a count may differ once in a long while)
    assert np.abs(counts - t.counts.astype(np.int64)).max() <= 1
    assert counts[0].sum() > 50 and counts[1].sum() > 0
    assert np.allclose(absorption, t.absorption, rtol=1e-6,
                       atol=1e-9 * t.absorption.max())
    assert absorption[2, 0, 0] > 0. and absorption[3, 0, 0] > 0.
    for b in backends:
        b.engine.close()


def test_trackers_with_their_own_numbers_of_bins(oracle):
    """The reference's trackers each have their own `number of bins`
    (test/test_tracker_manager.yml: 100, 1000, 100): counts tracker after
    tracker, equal to the oracle's."""
    from cmacionize_amd import engine as E
    ncell, npacket = 20, 40000
    sim = oracle.lexington_simulation(ncell)
    eng = lexington_engine(ncell, sim)
    pc = oracle.PC
    anchor, side = -5. * pc, 10. * pc
    positions = np.array([[1.3 * pc, 0.4 * pc, -0.7 * pc],
                          [-2.1 * pc, 1.9 * pc, 0.2 * pc],
                          [1.3 * pc, 0.4 * pc, -0.7 * pc],
                          [3.6 * pc, -3.2 * pc, 2.9 * pc]])
    bins = [40, 1000, 7, 100]
    eng.set_trackers(positions, [E.TRACKER_SPECTRUM] * 4, bins)
    cells = [cell_of(p, anchor, side, ncell) for p in positions]
    sim.run(npacket, 1, seed=42)
    eng.upload_cells(sim.number_density, sim.temperature,
                     np.array([np.asarray(x) for x in sim.x]))
    eng.enable_trackers(True)
    eng.reset_grid()
    eng.shoot(42, 1, 0, npacket)
    got = eng.get_tracker_counts()
    with oracle.Trackers(cells, bins) as t:
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, 1, 0, npacket)
    assert [g.shape for g in got] == [(3, b) for b in bins]
    for g, want in zip(got, t.counts):
        assert np.array_equal(g, want)
    # the same packets in coarser and finer bins
    assert got[0].sum() == got[2].sum() > 50
    assert got[1].sum() > 0 and got[3].sum() > 0
    eng.close()


def test_weighted_spectrum_trackers_match_oracle(oracle):
    """WeightedSpectrumTrackers (src/WeightedSpectrumTracker.hpp:44-446) with
    LinearFrequencyBins and LevelFrequencyBins beside a SpectrumTracker in the
    same cell: every crossing adds 1 / projected area to the bin of the
    packet's frequency. Engine and oracle see the same crossings; the sums
    differ by the order of the additions only."""
    from cmacionize_amd import engine as E
    ncell, npacket = 20, 40000
    sim = oracle.lexington_simulation(ncell)
    eng = lexington_engine(ncell, sim)
    pc = oracle.PC
    anchor, side = -5. * pc, 10. * pc
    positions = np.array([[1.3 * pc, 0.4 * pc, -0.7 * pc],
                          [1.3 * pc, 0.4 * pc, -0.7 * pc],
                          [1.3 * pc, 0.4 * pc, -0.7 * pc],
                          [-2.1 * pc, 1.9 * pc, 0.2 * pc],
                          [0.01 * pc, 0.01 * pc, 0.01 * pc]])  # in the hole
    W = E.TRACKER_WEIGHTED_SPECTRUM
    kinds = [W, E.TRACKER_SPECTRUM, W, W, W]
    bins = [100, 60, 14, 25, 10]
    frequency_bins = [None, None, ("Level",), ("Linear", 3.5e15, 6.e15), None]
    eng.set_trackers(positions, kinds, bins)
    eng.set_tracker_frequency_bins(2, "Level")
    eng.set_tracker_frequency_bins(3, "Linear", 3.5e15, 6.e15)
    with pytest.raises(E.EngineError, match="not a weighted"):
        eng.set_tracker_frequency_bins(1, "Level")
    with pytest.raises(E.EngineError, match="level bins are 14"):
        eng.set_tracker_frequency_bins(0, "Level")
    cells = [cell_of(p, anchor, side, ncell) for p in positions]
    sim.run(npacket, 1, seed=42)
    eng.upload_cells(sim.number_density, sim.temperature,
                     np.array([np.asarray(x) for x in sim.x]))
    eng.enable_trackers(True)
    eng.reset_grid()
    eng.shoot(42, 1, 0, npacket)
    got = eng.get_tracker_flux()
    counts = eng.get_tracker_counts()
    with oracle.Trackers(cells, bins, kinds=kinds,
                         frequency_bins=frequency_bins) as t:
        sim.reset()
        sim.totweight = 0.
        sim.typecount[:] = 0.
        sim.shoot(42, 1, 0, npacket)
    assert [g.shape for g in got] == [(4, b) for b in bins]
    for k, (g, want) in enumerate(zip(got, t.flux)):
        assert np.allclose(g, want, rtol=1e-12, atol=0.), k
        # an empty bin is empty in both
        assert np.array_equal(g == 0., want == 0.), k
    for c, want in zip(counts, t.counts):
        assert np.array_equal(c, want)
    # the weighted trackers of the cell saw the SpectrumTracker's packets (and
    # those outside its frequency range, in their end bins): at least one per
    # crossing / sqrt(3), at most one per crossing
    crossings = counts[1].sum()
    assert crossings > 50
    for k in (0, 2):
        total = got[k][:3].sum()
        assert total >= crossings / np.sqrt(3.) * (1. - 1e-12)
        assert abs(got[k].sum() - got[0].sum()) <= 1e-12 * got[0].sum()
    assert got[0][0].sum() > 0. and got[0][1].sum() > 0.
    assert not got[1].any() and not got[4].any() and not got[0][3].any()
    # the narrow linear bins collect what lies outside them at their ends
    assert got[3][:, 0].sum() > 0. and got[3][:, -1].sum() > 0.
    eng.close()
