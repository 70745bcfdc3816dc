"""ctypes binding of the CPU oracle (oracle/libcmio.so).

Test infrastructure only: imported by tests/, by __graft_entry__.smoke() and
by the cpu_baseline leg of bench.py. The product package (cmacionize_amd)
never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
NION = 14
NELEMENT = 7
NTYPE = 4

ION_NAMES = ["H_n", "He_n", "C_p1", "C_p2", "N_n", "N_p1", "N_p2", "O_n",
             "O_p1", "Ne_n", "Ne_p1", "S_p1", "S_p2", "S_p3"]

SPECTRUM_MONOCHROMATIC, SPECTRUM_PLANCK, SPECTRUM_TABLE = 0, 1, 2
XSEC_FIXED, XSEC_VERNER, XSEC_TABLE = 0, 1, 2
RECOMB_FIXED, RECOMB_VERNER, RECOMB_TABLE = 0, 1, 2
TABLE_LINEAR, TABLE_LOGLOG = 0, 1
REEMIT_NONE, REEMIT_PHYSICAL, REEMIT_FIXED = 0, 1, 2

dp = C.POINTER(C.c_double)


class Grid(C.Structure):
    _fields_ = [("anchor", C.c_double * 3), ("sides", C.c_double * 3),
                ("ncell", C.c_int32 * 3), ("periodic", C.c_int32 * 3)]


class Cells(C.Structure):
    _fields_ = [("number_density", dp), ("temperature", dp),
                ("ionic_fraction", dp * NION), ("mean_intensity", dp * NION),
                ("heating", dp * 2)]


class Table(C.Structure):
    """cmio_table: a plugin sampled on a grid of its argument."""
    _fields_ = [("x", dp), ("y", dp), ("n", C.c_int32),
                ("interpolation", C.c_int32)]


class Model(C.Structure):
    _fields_ = [
        ("nsource", C.c_int32),
        ("source_position", dp),
        ("source_cumulative", dp),
        ("total_luminosity", C.c_double),
        ("spectrum_type", C.c_int32),
        ("mono_frequency", C.c_double),
        ("planck_temperature", C.c_double),
        ("xsec_type", C.c_int32),
        ("xsec_fixed", C.c_double * NION),
        ("recomb_type", C.c_int32),
        ("recomb_fixed", C.c_double * NION),
        ("abundance", C.c_double * NELEMENT),
        ("reemit_type", C.c_int32),
        ("reemit_fixed_probability", C.c_double),
        ("reemit_fixed_frequency", C.c_double),
        ("do_temperature", C.c_int32),
        ("t_min_iteration", C.c_int32),
        ("t_epsilon", C.c_double),
        ("t_max_iterations", C.c_int32),
        ("pahfac", C.c_double),
        ("crfac", C.c_double),
        ("crlim", C.c_double),
        ("crscale", C.c_double),
        ("t_min_ionized", C.c_double),
        ("tables", C.c_void_p),
        ("continuous_type", C.c_int32),
        ("continuous_spectrum_type", C.c_int32),
        ("continuous_mono_frequency", C.c_double),
        ("continuous_planck_temperature", C.c_double),
        ("continuous_box_anchor", C.c_double * 3),
        ("continuous_box_sides", C.c_double * 3),
        ("discrete_luminosity", C.c_double),
        ("continuous_luminosity", C.c_double),
        ("continuous_probability", C.c_double),
        ("discrete_photon_weight", C.c_double),
        ("continuous_photon_weight", C.c_double),
        ("continuous_axis", C.c_int32),
        ("continuous_intercept", C.c_double),
        ("continuous_anchor", C.c_double * 2),
        ("continuous_side", C.c_double * 2),
        ("spectrum_table", Table * 2),
        ("xsec_table", Table),
        ("recomb_table", Table),
    ]


class Photon(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("direction", C.c_double * 3),
                ("inverse_direction", C.c_double * 3), ("energy", C.c_double),
                ("cross_section", C.c_double * NION),
                ("cross_section_He_corr", C.c_double),
                ("weight", C.c_double), ("type", C.c_int32)]


_lib = None


class OracleError(RuntimeError):
    """An error the oracle recorded (oracle/cmio_error.c) where the reference
    would have raised cmac_error."""


def _errcheck(result, func, args):
    msg = _raw_last_error()
    if msg:
        text = msg.decode()
        _raw_clear_error()
        raise OracleError(text)
    return result


class _CheckedCDLL(C.CDLL):
    """Every function of the library reports the oracle's error flag as a
    Python exception of the call that set it."""

    def __getitem__(self, name):
        f = super().__getitem__(name)
        if name not in ("cmio_last_error", "cmio_clear_error"):
            f.errcheck = _errcheck
        return f


_raw_last_error = _raw_clear_error = None


def build():
    """(Re)build oracle/libcmio.so with gcc; cheap when up to date."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ORACLE_DIR, "libcmio.so")
    if not os.path.exists(path):
        build()
    global _raw_last_error, _raw_clear_error
    L = _CheckedCDLL(path)
    _raw_last_error = L.cmio_last_error
    _raw_last_error.restype = C.c_char_p
    _raw_last_error.argtypes = []
    _raw_clear_error = L.cmio_clear_error
    _raw_clear_error.restype = None
    _raw_clear_error.argtypes = []
    L.cmio_rng_uniform.restype = C.c_double
    L.cmio_rng_uniform.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64,
                                   C.c_uint32]
    L.cmio_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
    L.cmio_interact.restype = C.c_int64
    L.cmio_interact.argtypes = [C.POINTER(Grid), C.POINTER(Model),
                                C.POINTER(Cells), C.POINTER(Photon),
                                C.c_double, C.POINTER(C.c_int64), dp,
                                C.c_int64, C.POINTER(C.c_int64)]
    L.cmio_shoot.argtypes = [C.POINTER(Grid), C.POINTER(Model),
                             C.POINTER(Cells), C.c_uint32, C.c_uint32,
                             C.c_uint64, C.c_uint64, dp, dp]
    L.cmio_shoot_fast.argtypes = L.cmio_shoot.argtypes
    L.cmio_subgrid_shoot.argtypes = [C.POINTER(Grid), C.POINTER(C.c_int32),
                                     C.POINTER(Model), C.POINTER(Cells),
                                     C.c_uint32, C.c_uint32, C.c_uint64,
                                     C.c_uint64, dp, dp,
                                     C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint64)]
    L.cmio_emit.argtypes = [C.POINTER(Model), C.c_uint32, C.c_uint32,
                            C.c_uint64, C.POINTER(Photon), dp,
                            C.POINTER(C.c_uint32)]
    L.cmio_mix_sources.argtypes = [C.POINTER(Model)]
    L.cmio_line_strengths.argtypes = [C.c_double, C.c_double, dp, dp]
    L.cmio_balmer_jump.argtypes = [C.c_double, dp]
    L.cmio_emissivities.argtypes = [C.POINTER(Model), C.c_double, C.c_double,
                                    dp, dp]
    L.cmio_reset_grid.argtypes = [C.POINTER(Grid), C.POINTER(Cells)]
    L.cmio_update_cells.argtypes = [C.POINTER(Grid), C.POINTER(Model),
                                    C.POINTER(Cells), C.c_uint32, C.c_double]
    L.cmio_update_cells_range.argtypes = [C.POINTER(Grid), C.POINTER(Model),
                                          C.POINTER(Cells), C.c_uint32,
                                          C.c_double, C.c_int64, C.c_int64]
    L.cmio_wall_intersection.argtypes = [dp, dp, dp, dp, dp,
                                         C.POINTER(C.c_int32), dp, dp]
    L.cmio_ionization_state_hydrogen.restype = C.c_double
    L.cmio_ionization_state_hydrogen.argtypes = [C.c_double] * 3
    L.cmio_ionization_states_hydrogen_helium.restype = C.c_int
    L.cmio_ionization_states_hydrogen_helium.argtypes = [C.c_double] * 7 + [
        dp, dp]
    L.cmio_ionization_state_cell.argtypes = [C.POINTER(Model), C.c_double,
                                             C.c_double, C.c_double,
                                             C.c_double, dp, dp, dp]
    L.cmio_num_threads.restype = C.c_int
    L.cmio_tables_create.restype = C.c_void_p
    L.cmio_tables_create.argtypes = [C.POINTER(Model)]
    L.cmio_tables_free.argtypes = [C.c_void_p]
    L.cmio_sample_spectrum.argtypes = [C.POINTER(Model), C.c_int, C.c_double,
                                       C.c_uint32, C.c_uint64, dp]
    L.cmio_he2pc_integral.restype = C.c_double
    L.cmio_reemission_probabilities.argtypes = [C.c_double, dp]
    L.cmio_solve_5x5.restype = C.c_int
    L.cmio_solve_5x5.argtypes = [dp, dp]
    L.cmio_line_cooling.restype = C.c_double
    L.cmio_line_cooling.argtypes = [C.c_double, C.c_double, dp]
    for name in ("cmio_lc_energy_difference",
                 "cmio_lc_transition_probability",
                 "cmio_lc_statistical_weight"):
        f = getattr(L, name)
        f.restype = C.c_double
        f.argtypes = [C.c_int, C.c_int]
    L.cmio_cooling_and_heating_balance.argtypes = [
        C.POINTER(Model), dp, dp, dp, dp, C.c_double, C.c_double, C.c_double,
        dp, dp, C.c_double, C.c_double, C.c_double, dp]
    L.cmio_temperature_cell.argtypes = [
        C.POINTER(Model), C.c_double, C.c_double, C.c_double, C.c_double, dp,
        dp, dp, dp]
    for name in ("cmio_verner_cross_section",
                 "cmio_verner_recombination_rate",
                 "cmio_ct_recombination_rate_H", "cmio_ct_ionization_rate_H",
                 "cmio_ct_recombination_rate_He"):
        f = getattr(L, name)
        f.restype = C.c_double
        f.argtypes = [C.c_int, C.c_double]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(dp)


# ---------------------------------------------------------------------------
# unit conversions of the reference (src/UnitConverter.hpp:98-160)
PC = 3.086e16
ELECTRONVOLT = 1.6021766208e-19
PLANCK = 6.626070040e-34


def eV_to_Hz(ev):
    return ev * ELECTRONVOLT * (1. / PLANCK) / 1.


class OracleSimulation:
    """SoA state + model for the oracle; numpy arrays are the storage.

    Mirrors the subset of IonizationSimulation (src/IonizationSimulation.cpp)
    that is on the hot path: reset_grid -> shoot -> update_cells per
    iteration.
    """

    def __init__(self, ncell, anchor, sides, periodic=(0, 0, 0),
                 compact=False, accumulators=None):
        """compact=True aliases the storage of all metal ions (fractions and
        mean intensities) to one scratch array each: valid only when their
        cross sections are zero (H-only runs); saves 24 of 32 arrays."""
        self.grid = Grid()
        for a in range(3):
            self.grid.anchor[a] = anchor[a]
            self.grid.sides[a] = sides[a]
            self.grid.ncell[a] = ncell[a]
            self.grid.periodic[a] = int(periodic[a])
        self.ncell = tuple(int(n) for n in ncell)
        n = int(np.prod(self.ncell))
        self.n = n
        self.number_density = np.zeros(n)
        self.temperature = np.zeros(n)
        nstore = 3 if compact else NION
        self._xs = np.zeros((nstore, n))
        if accumulators is not None:
            # caller-owned contiguous [16][n] block (14 J + 2 heating), the
            # layout of the engine's accumulator block
            assert accumulators.shape == (NION + 2, n) and not compact
            self._Js = accumulators[:NION]
            self.heating = accumulators[NION:]
        else:
            self._Js = np.zeros((nstore, n))
            self.heating = np.zeros((2, n))
        if compact:
            self.x = [self._xs[0], self._xs[1]] + [self._xs[2]] * 12
            self.J = [self._Js[0], self._Js[1]] + [self._Js[2]] * 12
        else:
            self.x = self._xs
            self.J = self._Js
        self.cells = Cells()
        self.cells.number_density = _ptr(self.number_density)
        self.cells.temperature = _ptr(self.temperature)
        for i in range(NION):
            self.cells.ionic_fraction[i] = _ptr(self.x[i])
            self.cells.mean_intensity[i] = _ptr(self.J[i])
        for i in range(2):
            self.cells.heating[i] = _ptr(self.heating[i])
        self.model = Model()
        m = self.model
        m.t_min_iteration = 3
        m.t_epsilon = 1.e-3
        m.t_max_iterations = 100
        m.crlim = 0.75
        m.crscale = 1.33333 * 3.086e19
        m.t_min_ionized = 4000.
        self.totweight = 0.
        self.typecount = np.zeros(NTYPE)

    def set_sources(self, positions, weights, luminosity):
        self._src_pos = np.ascontiguousarray(positions, dtype=np.float64)
        self._src_cum = np.cumsum(np.asarray(weights, dtype=np.float64))
        self._src_cum[-1] = 1.
        self.model.nsource = len(self._src_cum)
        self.model.source_position = _ptr(self._src_pos)
        self.model.source_cumulative = _ptr(self._src_cum)
        self.model.total_luminosity = luminosity
        self.model.discrete_luminosity = luminosity
        if self.model.continuous_type:
            lib().cmio_mix_sources(C.byref(self.model))

    def set_planar_continuous_source(self, axis, intercept, anchor, sides,
                                     luminosity, frequency):
        """PlanarContinuousPhotonSource with a monochromatic spectrum."""
        m = self.model
        m.continuous_type = 2
        m.continuous_axis = axis
        m.continuous_intercept = intercept
        for k in range(2):
            m.continuous_anchor[k] = anchor[k]
            m.continuous_side[k] = sides[k]
        m.continuous_spectrum_type = SPECTRUM_MONOCHROMATIC
        m.continuous_mono_frequency = frequency
        m.continuous_luminosity = luminosity
        lib().cmio_mix_sources(C.byref(m))

    def set_continuous_source(self, luminosity, frequency=None,
                              planck_temperature=None):
        """IsotropicContinuousPhotonSource on the simulation box with a
        monochromatic or Planck ContinuousPhotonSourceSpectrum; the mix with
        the discrete sources as the PhotonSource ctor computes it
        (src/PhotonSource.cpp:104-130)."""
        m = self.model
        m.continuous_type = 1
        for a in range(3):
            m.continuous_box_anchor[a] = self.grid.anchor[a]
            m.continuous_box_sides[a] = self.grid.sides[a]
        if frequency is not None:
            m.continuous_spectrum_type = SPECTRUM_MONOCHROMATIC
            m.continuous_mono_frequency = frequency
        else:
            m.continuous_spectrum_type = SPECTRUM_PLANCK
            m.continuous_planck_temperature = planck_temperature
        m.continuous_luminosity = luminosity
        lib().cmio_mix_sources(C.byref(m))

    def _set_table(self, table, x, y, interpolation):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        self._keep = getattr(self, "_keep", []) + [x, y]
        table.x = _ptr(x)
        table.y = _ptr(y)
        table.n = len(x)
        table.interpolation = interpolation

    def set_spectrum_table(self, frequency, cumulative, role=0,
                           interpolation=TABLE_LINEAR):
        """A PhotonSourceSpectrum given as its quantile function (role 0: the
        discrete sources', 1: the continuous source's)."""
        self._set_table(self.model.spectrum_table[role], cumulative,
                        frequency, interpolation)
        if role == 0:
            self.model.spectrum_type = SPECTRUM_TABLE
        else:
            self.model.continuous_spectrum_type = SPECTRUM_TABLE

    def set_cross_sections_table(self, frequency, sigma,
                                 interpolation=TABLE_LINEAR):
        assert np.shape(sigma) == (NION, len(frequency))
        self._set_table(self.model.xsec_table, frequency, sigma,
                        interpolation)
        self.model.xsec_type = XSEC_TABLE

    def set_recombination_rates_table(self, temperature, alpha,
                                      interpolation=TABLE_LOGLOG):
        assert np.shape(alpha) == (NION, len(temperature))
        self._set_table(self.model.recomb_table, temperature, alpha,
                        interpolation)
        self.model.recomb_type = RECOMB_TABLE

    def set_homogeneous(self, density, temperature, xH=1.e-6, xHe=1.e-6):
        """src/HomogeneousDensityFunction.hpp:99-107"""
        self.number_density[:] = density
        self.temperature[:] = temperature
        self._xs[:] = 0.
        self.x[0][:] = xH
        self.x[1][:] = xHe

    def reset(self):
        lib().cmio_reset_grid(C.byref(self.grid), C.byref(self.cells))

    def build_tables(self):
        """(Re)build the sampling tables after the spectrum / cross sections
        / re-emission settings of the model are final."""
        if self.model.tables:
            lib().cmio_tables_free(self.model.tables)
        self.model.tables = lib().cmio_tables_create(C.byref(self.model))

    def shoot(self, seed, iteration, first_packet, n_packets):
        if not self.model.tables and (
                self.model.spectrum_type == SPECTRUM_PLANCK or
                (self.model.continuous_type and
                 self.model.continuous_spectrum_type == SPECTRUM_PLANCK) or
                self.model.reemit_type == REEMIT_PHYSICAL):
            self.build_tables()
        tw = C.c_double(0.)
        tc = np.zeros(NTYPE)
        lib().cmio_shoot(C.byref(self.grid), C.byref(self.model),
                         C.byref(self.cells), seed, iteration, first_packet,
                         n_packets, C.byref(tw), _ptr(tc))
        self.totweight += tw.value
        self.typecount += tc
        return tw.value, tc

    def shoot_subgrids(self, nsub, seed, iteration, first_packet, n_packets):
        """cmio_subgrid_shoot: the reference's task-based semantics
        (DensitySubGrid::interact on nsub subgrids per axis) for the same
        packets; returns (totweight, typecount, cell crossings, subgrid
        changes)."""
        if not self.model.tables and (
                self.model.spectrum_type == SPECTRUM_PLANCK or
                self.model.reemit_type == REEMIT_PHYSICAL):
            self.build_tables()
        tw = C.c_double(0.)
        tc = np.zeros(NTYPE)
        ns, nh = C.c_uint64(0), C.c_uint64(0)
        lib().cmio_subgrid_shoot(C.byref(self.grid), (C.c_int32 * 3)(*nsub),
                                 C.byref(self.model), C.byref(self.cells),
                                 seed, iteration, first_packet, n_packets,
                                 C.byref(tw), _ptr(tc), C.byref(ns),
                                 C.byref(nh))
        self.totweight += tw.value
        self.typecount += tc
        return tw.value, tc, ns.value, nh.value

    def shoot_fast(self, seed, iteration, first_packet, n_packets):
        """cmio_shoot_fast: the CPU-baseline organisation of the same loop
        (array-of-structures cells, per-cell lock); same tallies up to the
        order of the additions."""
        if not self.model.tables and (
                self.model.spectrum_type == SPECTRUM_PLANCK or
                self.model.reemit_type == REEMIT_PHYSICAL):
            self.build_tables()
        tw = C.c_double(0.)
        tc = np.zeros(NTYPE)
        lib().cmio_shoot_fast(C.byref(self.grid), C.byref(self.model),
                              C.byref(self.cells), seed, iteration,
                              first_packet, n_packets, C.byref(tw), _ptr(tc))
        self.totweight += tw.value
        self.typecount += tc
        return tw.value, tc

    def update(self, loop, totweight):
        lib().cmio_update_cells(C.byref(self.grid), C.byref(self.model),
                                C.byref(self.cells), loop, totweight)

    def update_range(self, loop, totweight, first, count):
        """The cell update for cells [first, first + count) only."""
        lib().cmio_update_cells_range(C.byref(self.grid), C.byref(self.model),
                                      C.byref(self.cells), loop, totweight,
                                      first, count)

    def run(self, n_packets, n_iterations, seed=42):
        for loop in range(n_iterations):
            self.reset()
            self.totweight = 0.
            self.typecount[:] = 0.
            self.shoot(seed, loop, 0, n_packets)
            self.update(loop, self.totweight)


NEMISSIONLINE = 42
# EmissivityValues.hpp:36-81
EMISSION_LINES = [
    "HAlpha", "HBeta", "HII", "BALMER_JUMP_LOW", "BALMER_JUMP_HIGH",
    "OI_6300", "OI_6364", "OII_3727", "OIII_5007", "OIII_4959", "OIII_4363",
    "OIII_52mu", "OIII_88mu", "NII_5755", "NII_6548", "NII_6584",
    "NeIII_3869", "NeIII_3968", "SII_6725", "SII_4072", "SIII_9405",
    "SIII_6312", "SIII_19mu", "SIII_33mu", "avg_T", "avg_T_count",
    "avg_nH_nHe", "avg_nH_nHe_count", "NeII_12mu", "NIII_57mu", "NeIII_15mu",
    "NII_122mu", "CII_158mu", "CII_2325", "CIII_1908", "OII_7325", "SIV_10mu",
    "HeI_5876", "Hrec_s", "WFC2_F439W", "WFC2_F555W", "WFC2_F675W"]


def line_strengths(T, ne, abundances):
    """LineCoolingData::get_line_strengths: 103 values (10 x 10 + 3)."""
    a = np.ascontiguousarray(abundances, dtype=np.float64)
    out = np.zeros(103)
    lib().cmio_line_strengths(T, ne, _ptr(a), _ptr(out))
    return out


def balmer_jump(T):
    out = np.zeros(4)
    lib().cmio_balmer_jump(T, _ptr(out))
    return out


def emissivities(model, n, T, x):
    xx = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(NEMISSIONLINE)
    lib().cmio_emissivities(C.byref(model), n, T, _ptr(xx), _ptr(out))
    return out


def num_threads():
    return int(lib().cmio_num_threads())


def set_num_threads(n):
    lib().cmio_set_num_threads(int(n))


def stromgren_simulation(ncell=64, diffuse=False, compact=False,
                         accumulators=None):
    """benchmarks/stromgren.param (and stromgren_diffuse.param)."""
    sim = OracleSimulation((ncell,) * 3, (-5. * PC,) * 3, (10. * PC,) * 3,
                           compact=compact, accumulators=accumulators)
    sim.set_sources([[0., 0., 0.]], [1.], 4.26e49)
    sim.set_homogeneous(100. * 1.e6, 8000.)
    m = sim.model
    m.spectrum_type = SPECTRUM_MONOCHROMATIC
    m.mono_frequency = eV_to_Hz(13.6)
    m.xsec_type = XSEC_FIXED
    m.xsec_fixed[0] = 6.3e-18 * 1.e-4
    m.recomb_type = RECOMB_FIXED
    m.recomb_fixed[0] = 4.e-13 * 1.e-6
    m.reemit_type = REEMIT_PHYSICAL if diffuse else REEMIT_NONE
    m.do_temperature = 0
    return sim


def block_syntax_density(ncell, anchor, sides, blocks):
    """BlockSyntaxDensityFunction::operator() on the cell midpoints
    (src/BlockSyntaxDensityFunction.hpp:150-190, BlockSyntaxBlock.hpp:91-105).
    blocks: list of (origin[3], sides[3], exponent, density, temperature,
    neutral_fraction_H); later blocks override earlier ones."""
    ax = [anchor[a] + (np.arange(ncell[a]) + 0.5) * (sides[a] / ncell[a])
          for a in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    pos = [X.ravel(), Y.ravel(), Z.ravel()]
    n = np.full(X.size, -1.)
    T = np.full(X.size, -1.)
    xH = np.full(X.size, -1.)
    for origin, bsides, exponent, density, temperature, nfH in blocks:
        r = np.zeros(X.size)
        for a in range(3):
            x = 2. * np.abs(pos[a] - origin[a]) / bsides[a]
            if exponent < 10.:
                r += x ** exponent
            else:
                r = np.maximum(r, x)
        if exponent < 10.:
            r = r ** (1. / exponent)
        inside = r <= 1.
        n[inside] = density
        T[inside] = temperature
        xH[inside] = nfH
    assert n.min() >= 0. and T.min() >= 0.
    return n, T, xH


LEXINGTON_ABUNDANCES = dict(He=0.1, C=2.2e-4, N=4.e-5, O=3.3e-4, Ne=5.e-5,
                            S=9.e-6)


def lexington_simulation(ncell=32, star_temperature=40000.):
    """benchmarks/lexingtonHII40.param + lexingtonHII40.yml."""
    anchor = (-5. * PC,) * 3
    sides = (10. * PC,) * 3
    sim = OracleSimulation((ncell,) * 3, anchor, sides)
    sim.set_sources([[0., 0., 0.]], [1.], 4.26e49)
    n, T, xH = block_syntax_density(
        (ncell,) * 3, anchor, sides,
        [((0., 0., 0.), (10. * PC,) * 3, 10., 100. * 1.e6, 8000., 1.e-6),
         ((0., 0., 0.), (6.e18 * 0.01,) * 3, 2., 0., 0., 1.e-6)])
    sim.number_density[:] = n
    sim.temperature[:] = T
    sim._xs[:] = 0.
    sim.x[0][:] = xH
    sim.x[1][:] = 1.e-6
    m = sim.model
    m.spectrum_type = SPECTRUM_PLANCK
    m.planck_temperature = star_temperature
    m.xsec_type = XSEC_VERNER
    m.recomb_type = RECOMB_VERNER
    for i, el in enumerate(("He", "C", "N", "O", "Ne", "S")):
        m.abundance[1 + i] = LEXINGTON_ABUNDANCES[el]
    m.reemit_type = REEMIT_PHYSICAL
    m.do_temperature = 1
    m.pahfac = 0.
    return sim


def projected_area(direction):
    """WeightedSpectrumTracker::get_projected_area (cmio_projected_area)"""
    L = lib()
    L.cmio_projected_area.argtypes = [dp]
    L.cmio_projected_area.restype = C.c_double
    d = np.ascontiguousarray(direction, dtype=np.float64)
    return L.cmio_projected_area(_ptr(d))


def frequency_bin(kind, nbins, minimum, maximum, frequency):
    """FrequencyBins::get_bin_number (cmio_frequency_bin) of the kind "Linear"
    or "Level" of bins"""
    L = lib()
    L.cmio_frequency_bin.argtypes = [C.c_int32, C.c_int32, C.c_double,
                                     C.c_double, C.c_double]
    L.cmio_frequency_bin.restype = C.c_int32
    return L.cmio_frequency_bin(int(kind == "Level"), nbins, minimum, maximum,
                                frequency)


class Trackers:
    """SpectrumTrackers on cells of the oracle's grid (cmio_set_trackers):
    installed while the object is used as a context manager."""

    def __init__(self, cells, nbins=100, opening_angles=None,
                 reference_directions=None, kinds=None, frequency_bins=None):
        self.cells = np.ascontiguousarray(cells, dtype=np.int64)
        n = len(self.cells)
        # kinds[k] == 2: a WeightedSpectrumTracker, sums in flux[k][type][bin]
        # (a list of views); frequency_bins[k] = ("Linear", minimum, maximum)
        # or ("Level",) - default: Linear from 13.6 eV to 54.4 eV
        ev = 1.6021766208e-19 / 6.626070040e-34
        fb = [None] * n if frequency_bins is None else list(frequency_bins)
        fb = [("Linear", 13.6 * ev, 54.4 * ev) if f is None else tuple(f)
              for f in fb]
        self.bins_type = np.array([f[0] == "Level" for f in fb],
                                  dtype=np.int32)
        self.bins_min = np.array([f[1] if len(f) > 1 else 0. for f in fb])
        self.bins_max = np.array([f[2] if len(f) > 2 else 0. for f in fb])
        # nbins: one number for all trackers, or one per tracker (counts is
        # then a list of [3][bins] views)
        self.bins = None
        if not np.isscalar(nbins):
            self.bins = np.ascontiguousarray(nbins, dtype=np.int32)
            nbins = int(self.bins.max())
        # kinds[k] != 0: an AbsorptionTracker, sums in absorption[k][type][ion]
        self.kinds = None if kinds is None else \
            np.ascontiguousarray(kinds, dtype=np.int32)
        self.absorption = np.zeros((n, 4, NION))
        self.nbins = nbins
        ang = np.full(n, np.pi) if opening_angles is None else \
            np.asarray(opening_angles, dtype=np.float64)
        self.cosang = np.ascontiguousarray(np.cos(ang))
        d = np.zeros((n, 3)) if reference_directions is None else \
            np.array(reference_directions, dtype=np.float64).reshape(n, 3)
        norm = np.sqrt((d * d).sum(axis=1))
        d[norm > 0.] /= norm[norm > 0., None]
        self.directions = np.ascontiguousarray(d)
        if self.bins is None:
            self.counts = np.zeros((n, 3, nbins), dtype=np.uint64)
            self._flat = self.counts
        else:
            self._flat = np.zeros(3 * int(self.bins.sum()), dtype=np.uint64)
            self.counts, at = [], 0
            for b in self.bins:
                self.counts.append(self._flat[at:at + 3 * b].reshape(3, b))
                at += 3 * int(b)
        each = [nbins] * n if self.bins is None else [int(b) for b in self.bins]
        self._flux = np.zeros(4 * max(sum(each), 1))
        self.flux, at = [], 0
        for b in each:
            self.flux.append(self._flux[at:at + 4 * b].reshape(4, b))
            at += 4 * b

    def __enter__(self):
        L = lib()
        L.cmio_set_trackers.argtypes = [
            C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double),
            C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.cmio_set_trackers.restype = None
        L.cmio_set_trackers(
            len(self.cells), self.nbins,
            self.cells.ctypes.data_as(C.POINTER(C.c_int64)),
            _ptr(self.cosang), _ptr(self.directions),
            self._flat.ctypes.data_as(C.POINTER(C.c_uint64)))
        if self.bins is not None:
            L.cmio_set_tracker_bins.argtypes = [C.POINTER(C.c_int32)]
            L.cmio_set_tracker_bins.restype = None
            L.cmio_set_tracker_bins(
                self.bins.ctypes.data_as(C.POINTER(C.c_int32)))
        if self.kinds is not None:
            L.cmio_set_tracker_kinds.argtypes = [C.POINTER(C.c_int32), dp]
            L.cmio_set_tracker_kinds.restype = None
            L.cmio_set_tracker_kinds(
                self.kinds.ctypes.data_as(C.POINTER(C.c_int32)),
                _ptr(self.absorption))
            L.cmio_set_tracker_weighted.argtypes = [
                dp, C.POINTER(C.c_int32), dp, dp]
            L.cmio_set_tracker_weighted.restype = None
            L.cmio_set_tracker_weighted(
                _ptr(self._flux),
                self.bins_type.ctypes.data_as(C.POINTER(C.c_int32)),
                _ptr(self.bins_min), _ptr(self.bins_max))
        return self

    def __exit__(self, *exc):
        lib().cmio_set_trackers(0, 0, None, None, None, None)
        return False
