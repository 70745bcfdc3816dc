"""ctypes binding of the engine's C ABI (include/cmi_gpu.h, libcmi_gpu.so).

This is plumbing only: every method is one call through the C ABI. There is no
Python or CPU implementation of the path behind it - if the HIP library is
missing or no device is present, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CMI_GPU_LIBRARY: load another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("CMI_GPU_LIBRARY",
                          os.path.join(_HERE, "libcmi_gpu.so"))

NION = 14
TRACKER_SPECTRUM, TRACKER_ABSORPTION, TRACKER_WEIGHTED_SPECTRUM = 0, 1, 2
NACC = 16
NTYPE = 4

FIELD_NUMBER_DENSITY = 0
FIELD_TEMPERATURE = 1
FIELD_IONIC_FRACTION = 2
FIELD_MEAN_INTENSITY = 16
FIELD_HEATING = 30

REEMIT_NONE, REEMIT_PHYSICAL, REEMIT_FIXED = 0, 1, 2
CONTINUOUS_NONE, CONTINUOUS_ISOTROPIC, CONTINUOUS_PLANAR = 0, 1, 2
# cmi_gpu_set_*_table (the generic lowering of a plugin into a table)
ROLE_SOURCE, ROLE_CONTINUOUS = 0, 1
TABLE_LINEAR, TABLE_LOGLOG = 0, 1

_dp = C.POINTER(C.c_double)


class Config(C.Structure):
    _fields_ = [("anchor", C.c_double * 3), ("sides", C.c_double * 3),
                ("ncell", C.c_int32 * 3), ("periodic", C.c_int32 * 3),
                ("device", C.c_int32), ("track_heating", C.c_int32),
                ("stream", C.c_void_p),
                ("external_accumulators", C.c_void_p),
                ("sub_offset", C.c_int32 * 3), ("sub_ncell", C.c_int32 * 3)]


class TemperatureParams(C.Structure):
    _fields_ = [("do_temperature_calculation", C.c_int32),
                ("minimum_number_of_iterations", C.c_int32),
                ("epsilon_convergence", C.c_double),
                ("maximum_number_of_iterations", C.c_int32),
                ("pah_heating_factor", C.c_double),
                ("cosmic_ray_heating_factor", C.c_double),
                ("cosmic_ray_heating_limit", C.c_double),
                ("cosmic_ray_heating_scale_length", C.c_double),
                ("minimum_ionized_temperature", C.c_double)]


class EngineError(RuntimeError):
    pass


# every symbol include/cmi_gpu.h declares
EXPORTED_SYMBOLS = [
    "cmi_gpu_create", "cmi_gpu_destroy", "cmi_gpu_last_error",
    "cmi_gpu_synchronize", "cmi_gpu_number_of_cells", "cmi_gpu_set_sources",
    "cmi_gpu_set_spectrum_monochromatic", "cmi_gpu_set_spectrum_planck",
    "cmi_gpu_set_continuous_source",
    "cmi_gpu_set_continuous_source_planar",
    "cmi_gpu_set_continuous_spectrum_monochromatic",
    "cmi_gpu_set_continuous_spectrum_planck",
    "cmi_gpu_set_cross_sections_fixed", "cmi_gpu_set_cross_sections_verner",
    "cmi_gpu_set_spectrum_table", "cmi_gpu_set_cross_sections_table",
    "cmi_gpu_set_recombination_rates_table",
    "cmi_gpu_set_recombination_rates_fixed",
    "cmi_gpu_set_recombination_rates_verner", "cmi_gpu_set_abundances",
    "cmi_gpu_set_reemission", "cmi_gpu_set_temperature_params",
    "cmi_gpu_upload_cells", "cmi_gpu_upload_field", "cmi_gpu_download_field",
    "cmi_gpu_field_device_pointer", "cmi_gpu_reset_grid", "cmi_gpu_shoot",
    "cmi_gpu_get_counters", "cmi_gpu_update_cells", "cmi_gpu_emit_packets",
    "cmi_gpu_trace_packets", "cmi_gpu_get_timing", "cmi_gpu_set_tuning",
    "cmi_gpu_get_atomic_count", "cmi_gpu_sample_spectrum",
    "cmi_gpu_thermal_probe", "cmi_gpu_accumulator_layout",
    "cmi_gpu_get_kernel_timing", "cmi_gpu_get_wave_steps",
    "cmi_gpu_get_launch_times", "cmi_gpu_set_export_buffer",
    "cmi_gpu_get_export_count", "cmi_gpu_reset_exports",
    "cmi_gpu_shoot_flights", "cmi_gpu_download_exports",
    "cmi_gpu_shoot_flights_host", "cmi_gpu_physics_probe",
    "cmi_gpu_update_cells_range", "cmi_gpu_refresh_transport_records",
    "cmi_gpu_get_launch_steps", "cmi_gpu_group_create",
    "cmi_gpu_group_destroy", "cmi_gpu_group_reduce_accumulators",
    "cmi_gpu_group_update_cells",
    "cmi_gpu_group_exchange_flights", "cmi_gpu_group_exchange_stats",
    "cmi_gpu_compute_emissivities",
    "cmi_gpu_set_spectrum_trackers", "cmi_gpu_enable_trackers",
    "cmi_gpu_get_tracker_counts", "cmi_gpu_set_trackers",
    "cmi_gpu_get_tracker_absorption", "cmi_gpu_set_tracker_frequency_bins",
    "cmi_gpu_get_tracker_flux", "cmi_gpu_projected_areas",
]

# the emission lines of EmissivityValues (src/EmissivityValues.hpp:36-81), in
# the order cmi_gpu_compute_emissivities numbers them
EMISSION_LINES = [
    "HAlpha", "HBeta", "HII", "BALMER_JUMP_LOW", "BALMER_JUMP_HIGH",
    "OI_6300", "OI_6364", "OII_3727", "OIII_5007", "OIII_4959", "OIII_4363",
    "OIII_52mu", "OIII_88mu", "NII_5755", "NII_6548", "NII_6584",
    "NeIII_3869", "NeIII_3968", "SII_6725", "SII_4072", "SIII_9405",
    "SIII_6312", "SIII_19mu", "SIII_33mu", "avg_T", "avg_T_count",
    "avg_nH_nHe", "avg_nH_nHe_count", "NeII_12mu", "NIII_57mu", "NeIII_15mu",
    "NII_122mu", "CII_158mu", "CII_2325", "CIII_1908", "OII_7325", "SIV_10mu",
    "HeI_5876", "Hrec_s", "WFC2_F439W", "WFC2_F555W", "WFC2_F675W"]

_lib = None


def load_library():
    """Load libcmi_gpu.so (built by __graft_entry__.build()); no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            "HIP engine library not found at %s - build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C cmacionize_amd/csrc`. There is no CPU fallback." %
            LIB_PATH)
    # torch ships its own copy of the HIP runtime; if the engine's library
    # pulls in the system one first, torch cannot initialise the GPU later in
    # the same process. Let torch (the owner of streams and of the buffers it
    # shares with the engine) come first when it is there.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.cmi_gpu_last_error.restype = C.c_char_p
    L.cmi_gpu_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.cmi_gpu_destroy.argtypes = [vp]
    L.cmi_gpu_synchronize.argtypes = [vp]
    L.cmi_gpu_number_of_cells.restype = C.c_int64
    L.cmi_gpu_number_of_cells.argtypes = [vp]
    L.cmi_gpu_set_sources.argtypes = [vp, C.c_int32, _dp, _dp, C.c_double]
    L.cmi_gpu_set_spectrum_monochromatic.argtypes = [vp, C.c_double]
    L.cmi_gpu_set_spectrum_planck.argtypes = [vp, C.c_double]
    L.cmi_gpu_set_continuous_source.argtypes = [vp, C.c_int32, C.c_double]
    L.cmi_gpu_set_continuous_source_planar.argtypes = [
        vp, C.c_int32, C.c_double, _dp, _dp, C.c_double]
    L.cmi_gpu_set_continuous_spectrum_monochromatic.argtypes = [vp, C.c_double]
    L.cmi_gpu_set_continuous_spectrum_planck.argtypes = [vp, C.c_double]
    L.cmi_gpu_set_cross_sections_fixed.argtypes = [vp, _dp]
    L.cmi_gpu_set_spectrum_table.argtypes = [vp, C.c_int32, C.c_int32, _dp,
                                             _dp, C.c_int32]
    L.cmi_gpu_set_cross_sections_table.argtypes = [vp, C.c_int32, _dp, _dp,
                                                   C.c_int32]
    L.cmi_gpu_set_recombination_rates_table.argtypes = [vp, C.c_int32, _dp,
                                                        _dp, C.c_int32]
    L.cmi_gpu_set_cross_sections_verner.argtypes = [vp]
    L.cmi_gpu_set_recombination_rates_fixed.argtypes = [vp, _dp]
    L.cmi_gpu_set_recombination_rates_verner.argtypes = [vp]
    L.cmi_gpu_set_abundances.argtypes = [vp, _dp]
    L.cmi_gpu_set_reemission.argtypes = [vp, C.c_int32, C.c_double,
                                         C.c_double]
    L.cmi_gpu_set_temperature_params.argtypes = [
        vp, C.POINTER(TemperatureParams)]
    L.cmi_gpu_upload_cells.argtypes = [vp, _dp, _dp, _dp]
    L.cmi_gpu_upload_field.argtypes = [vp, C.c_int32, _dp]
    L.cmi_gpu_download_field.argtypes = [vp, C.c_int32, _dp]
    L.cmi_gpu_field_device_pointer.restype = C.c_void_p
    L.cmi_gpu_field_device_pointer.argtypes = [vp, C.c_int32]
    L.cmi_gpu_reset_grid.argtypes = [vp]
    L.cmi_gpu_shoot.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64,
                                C.c_uint64]
    L.cmi_gpu_get_counters.argtypes = [vp, _dp, _dp, C.POINTER(C.c_uint64)]
    L.cmi_gpu_update_cells.argtypes = [vp, C.c_uint32, C.c_double]
    L.cmi_gpu_emit_packets.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64,
                                       C.c_uint64, _dp, _dp, _dp, _dp, _dp]
    L.cmi_gpu_trace_packets.argtypes = [
        vp, C.c_uint64, _dp, _dp, _dp, _dp, _dp, C.c_int32,
        C.POINTER(C.c_int64), _dp, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
        _dp]
    L.cmi_gpu_get_timing.argtypes = [vp, C.c_int32, _dp, C.POINTER(C.c_uint64),
                                     _dp, C.POINTER(C.c_uint64)]
    L.cmi_gpu_get_kernel_timing.argtypes = [vp, _dp, C.POINTER(C.c_uint64)]
    L.cmi_gpu_set_tuning.argtypes = [vp, C.c_char_p, C.c_int64]
    L.cmi_gpu_get_atomic_count.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.cmi_gpu_get_wave_steps.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.cmi_gpu_set_export_buffer.argtypes = [vp, vp, C.c_uint64]
    L.cmi_gpu_get_export_count.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.cmi_gpu_reset_exports.argtypes = [vp]
    L.cmi_gpu_download_exports.argtypes = [vp, _dp, C.c_uint64,
                                           C.POINTER(C.c_uint64)]
    L.cmi_gpu_shoot_flights_host.argtypes = [vp, C.c_uint32, C.c_uint32,
                                             C.c_uint64, _dp, C.c_uint64]
    L.cmi_gpu_shoot_flights.argtypes = [vp, C.c_uint32, C.c_uint32,
                                        C.c_uint64, vp, C.c_uint64]
    L.cmi_gpu_get_launch_times.argtypes = [vp, C.c_uint64, _dp,
                                           C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint64)]
    L.cmi_gpu_sample_spectrum.argtypes = [vp, C.c_int32, C.c_double,
                                          C.c_uint32, C.c_uint64, _dp]
    L.cmi_gpu_accumulator_layout.argtypes = [vp, C.POINTER(C.c_int64),
                                             C.POINTER(C.c_int64)]
    L.cmi_gpu_thermal_probe.argtypes = [vp, C.c_int64, C.c_int32, _dp, _dp,
                                        _dp, _dp, _dp, _dp, _dp]
    L.cmi_gpu_physics_probe.argtypes = [vp, C.c_int32, C.c_int64, _dp, _dp]
    L.cmi_gpu_compute_emissivities.argtypes = [
        vp, C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.c_int64,
        C.POINTER(C.c_double)]
    L.cmi_gpu_set_spectrum_trackers.argtypes = [vp, C.c_int32, _dp, C.c_int32,
                                                _dp, _dp]
    L.cmi_gpu_set_trackers.argtypes = [vp, C.c_int32, _dp,
                                       C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32), _dp, _dp]
    L.cmi_gpu_get_tracker_absorption.argtypes = [vp, _dp]
    L.cmi_gpu_enable_trackers.argtypes = [vp, C.c_int32]
    L.cmi_gpu_set_tracker_frequency_bins.argtypes = [
        vp, C.c_int32, C.c_int32, C.c_double, C.c_double]
    L.cmi_gpu_get_tracker_flux.argtypes = [vp, _dp]
    L.cmi_gpu_projected_areas.argtypes = [_dp, C.c_int64, _dp]
    L.cmi_gpu_get_tracker_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.cmi_gpu_update_cells_range.argtypes = [vp, C.c_uint32, C.c_double,
                                             C.c_int64, C.c_int64]
    L.cmi_gpu_refresh_transport_records.argtypes = [vp]
    L.cmi_gpu_get_launch_steps.argtypes = [vp, C.c_uint64,
                                           C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint64)]
    L.cmi_gpu_group_create.argtypes = [C.c_int32, C.POINTER(vp),
                                       C.POINTER(vp)]
    L.cmi_gpu_group_destroy.argtypes = [vp]
    L.cmi_gpu_group_reduce_accumulators.argtypes = [vp]
    L.cmi_gpu_group_update_cells.argtypes = [vp, C.c_uint32, C.c_double]
    L.cmi_gpu_group_exchange_flights.argtypes = [
        vp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
    L.cmi_gpu_group_exchange_stats.argtypes = [
        vp, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_int32]
    _lib = L
    return L


def projected_areas(directions):
    """WeightedSpectrumTracker::get_projected_area of unit vectors ([n][3]),
    by the function the kernels call, run on the host"""
    d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
    out = np.zeros(len(d))
    rc = load_library().cmi_gpu_projected_areas(_p(d), len(d), _p(out))
    if rc != 0:
        raise RuntimeError("cmi_gpu_projected_areas failed (%d)" % rc)
    return out


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class EngineGroup:
    """Several engines driven by this process (cmi_gpu_group_*): replicas
    (reduce_accumulators) or the blocks of a decomposed grid
    (exchange_flights)."""

    def __init__(self, engines):
        self._lib = load_library()
        self.engines = list(engines)
        handles = (C.c_void_p * len(self.engines))(
            *[e._h for e in self.engines])
        self._h = C.c_void_p()
        rc = self._lib.cmi_gpu_group_create(len(self.engines), handles,
                                            C.byref(self._h))
        if rc != 0:
            raise EngineError(self._lib.cmi_gpu_last_error().decode())

    def _check(self, rc):
        if rc != 0:
            raise EngineError(self._lib.cmi_gpu_last_error().decode())

    def reduce_accumulators(self):
        self._check(self._lib.cmi_gpu_group_reduce_accumulators(self._h))

    def update_cells(self, loop, totweight):
        """Sharded cell update of every class of the group (slab r solved
        by member r, then gathered into every member)."""
        self._check(self._lib.cmi_gpu_group_update_cells(self._h, loop,
                                                         totweight))

    def exchange_flights(self, seed, iteration, first_packet=0):
        total = C.c_uint64()
        self._check(self._lib.cmi_gpu_group_exchange_flights(
            self._h, seed, iteration, first_packet, C.byref(total)))
        return total.value

    def exchange_stats(self, reset=True):
        """Host-side cost of the exchange rounds since the last reset:
        {rounds, counts_us, threads_us, total_us} (sums over the rounds)."""
        rounds = C.c_uint64()
        us = (C.c_double * 3)()
        self._check(self._lib.cmi_gpu_group_exchange_stats(
            self._h, C.byref(rounds), us, 1 if reset else 0))
        return dict(rounds=rounds.value, counts_us=us[0], threads_us=us[1],
                    total_us=us[2])

    def close(self):
        if self._h:
            self._lib.cmi_gpu_group_destroy(self._h)
            self._h = C.c_void_p()


class GpuEngine:
    """One engine handle = one grid on one GPU."""

    def __init__(self, ncell, anchor, sides, periodic=(0, 0, 0), device=0,
                 track_heating=False, stream=None, external_accumulators=None,
                 sub_offset=None, sub_ncell=None):
        """ncell/anchor/sides describe the whole grid; sub_offset/sub_ncell
        make the engine hold one block of it (domain decomposition)."""
        self._lib = load_library()
        cfg = Config()
        if sub_ncell is not None:
            for a in range(3):
                cfg.sub_offset[a] = int(sub_offset[a])
                cfg.sub_ncell[a] = int(sub_ncell[a])
        for a in range(3):
            cfg.anchor[a] = anchor[a]
            cfg.sides[a] = sides[a]
            cfg.ncell[a] = int(ncell[a])
            cfg.periodic[a] = int(bool(periodic[a]))
        cfg.device = device
        cfg.track_heating = int(bool(track_heating))
        cfg.stream = stream
        cfg.external_accumulators = external_accumulators
        self._h = C.c_void_p()
        self._check(self._lib.cmi_gpu_create(C.byref(cfg), C.byref(self._h)))
        self.ncell = tuple(int(n) for n in
                           (sub_ncell if sub_ncell is not None else ncell))
        self.n = int(np.prod(self.ncell))
        # tests, bench and tools read device timings; the engine records them
        # only on request
        self.set_tuning(timing=1)

    def _check(self, rc):
        if rc != 0:
            raise EngineError("cmi_gpu error %d: %s" % (
                rc, self._lib.cmi_gpu_last_error().decode()))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.cmi_gpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # plugin descriptors -----------------------------------------------------
    def set_sources(self, positions, weights, luminosity):
        w = _f64(weights)
        if len(w) == 0:  # no discrete sources
            self._check(self._lib.cmi_gpu_set_sources(self._h, 0, None, None,
                                                      0.))
            return
        pos = _f64(positions).reshape(-1, 3)
        self._check(self._lib.cmi_gpu_set_sources(self._h, len(w), _p(pos),
                                                  _p(w), luminosity))

    def set_continuous_source(self, kind, luminosity):
        """ContinuousPhotonSource (CONTINUOUS_ISOTROPIC on the box) with the
        luminosity the PhotonSource ctor computes for it."""
        self._check(self._lib.cmi_gpu_set_continuous_source(self._h, kind,
                                                            luminosity))

    def set_continuous_source_planar(self, axis, intercept, anchor, sides,
                                     luminosity):
        """PlanarContinuousPhotonSource: the rectangle [anchor, anchor +
        sides] of the plane x[axis] = intercept."""
        a = _f64(anchor)
        s = _f64(sides)
        self._check(self._lib.cmi_gpu_set_continuous_source_planar(
            self._h, axis, intercept, _p(a), _p(s), luminosity))

    def set_continuous_spectrum_monochromatic(self, frequency):
        self._check(self._lib.cmi_gpu_set_continuous_spectrum_monochromatic(
            self._h, frequency))

    def set_continuous_spectrum_planck(self, temperature):
        self._check(self._lib.cmi_gpu_set_continuous_spectrum_planck(
            self._h, temperature))

    def set_spectrum_monochromatic(self, frequency):
        self._check(self._lib.cmi_gpu_set_spectrum_monochromatic(self._h,
                                                                 frequency))

    def set_spectrum_planck(self, temperature):
        self._check(self._lib.cmi_gpu_set_spectrum_planck(self._h,
                                                          temperature))

    def set_spectrum_table(self, frequency, cumulative, role=ROLE_SOURCE,
                           interpolation=TABLE_LINEAR):
        """Generic lowering of a PhotonSourceSpectrum: its quantile function
        (cumulative[n] from 0 to 1 -> frequency[n] in Hz)."""
        f, c = _f64(frequency), _f64(cumulative)
        assert f.shape == c.shape and f.ndim == 1
        self._check(self._lib.cmi_gpu_set_spectrum_table(
            self._h, role, len(f), _p(f), _p(c), interpolation))

    def set_cross_sections_table(self, frequency, sigma,
                                 interpolation=TABLE_LINEAR):
        """Generic lowering of CrossSections: sigma[14][n] (m^2) on the
        frequencies frequency[n] (Hz)."""
        f, s = _f64(frequency), _f64(sigma)
        assert s.shape == (NION, len(f))
        self._check(self._lib.cmi_gpu_set_cross_sections_table(
            self._h, len(f), _p(f), _p(s), interpolation))

    def set_recombination_rates_table(self, temperature, alpha,
                                      interpolation=TABLE_LOGLOG):
        """Generic lowering of RecombinationRates: alpha[14][n] (m^3 s^-1) on
        the temperatures temperature[n] (K)."""
        t, a = _f64(temperature), _f64(alpha)
        assert a.shape == (NION, len(t))
        self._check(self._lib.cmi_gpu_set_recombination_rates_table(
            self._h, len(t), _p(t), _p(a), interpolation))

    def set_cross_sections_fixed(self, sigma):
        s = _f64(sigma)
        assert s.shape == (NION,)
        self._check(self._lib.cmi_gpu_set_cross_sections_fixed(self._h, _p(s)))

    def set_cross_sections_verner(self):
        self._check(self._lib.cmi_gpu_set_cross_sections_verner(self._h))

    def set_recombination_rates_fixed(self, alpha):
        a = _f64(alpha)
        assert a.shape == (NION,)
        self._check(self._lib.cmi_gpu_set_recombination_rates_fixed(self._h,
                                                                    _p(a)))

    def set_recombination_rates_verner(self):
        self._check(self._lib.cmi_gpu_set_recombination_rates_verner(self._h))

    def set_abundances(self, abundances):
        a = _f64(abundances)
        assert a.shape == (6,)
        self._check(self._lib.cmi_gpu_set_abundances(self._h, _p(a)))

    def set_reemission(self, kind, probability=0., frequency=0.):
        self._check(self._lib.cmi_gpu_set_reemission(self._h, kind,
                                                     probability, frequency))

    def set_temperature_params(self, **kw):
        p = TemperatureParams(0, 3, 1.e-3, 100, 0., 0., 0.75,
                              1.33333 * 3.086e19, 4000.)
        for k, v in kw.items():
            setattr(p, k, v)
        self._check(self._lib.cmi_gpu_set_temperature_params(self._h,
                                                             C.byref(p)))

    # cell data --------------------------------------------------------------
    def upload_cells(self, number_density, temperature, ionic_fractions=None):
        n = _f64(number_density).ravel()
        t = _f64(temperature).ravel()
        assert n.size == self.n and t.size == self.n
        x = None
        if ionic_fractions is not None:
            x = _f64(ionic_fractions).reshape(NION, self.n)
        self._check(self._lib.cmi_gpu_upload_cells(
            self._h, _p(n), _p(t), _p(x) if x is not None else None))

    def upload_field(self, field, values):
        v = _f64(values).ravel()
        assert v.size == self.n
        self._check(self._lib.cmi_gpu_upload_field(self._h, field, _p(v)))

    def download_field(self, field):
        out = np.empty(self.n)
        self._check(self._lib.cmi_gpu_download_field(self._h, field, _p(out)))
        return out

    def accumulator_layout(self):
        fs, cs = C.c_int64(), C.c_int64()
        self._check(self._lib.cmi_gpu_accumulator_layout(
            self._h, C.byref(fs), C.byref(cs)))
        return fs.value, cs.value

    def field_device_pointer(self, field):
        return self._lib.cmi_gpu_field_device_pointer(self._h, field)

    # iteration body -----------------------------------------------------------
    def field_tensor(self, field):
        """Zero-copy torch view of a state field ([ncell] doubles on the
        engine's device) - for collectives on the engine's memory."""
        import torch
        ptr = self.field_device_pointer(field)
        n = self.n

        class _View:
            __cuda_array_interface__ = {
                "shape": (n,), "typestr": "<f8", "data": (int(ptr), False),
                "version": 2}
        return torch.as_tensor(_View(), device="cuda")

    def refresh_transport_records(self):
        self._check(self._lib.cmi_gpu_refresh_transport_records(self._h))

    def reset_grid(self):
        self._check(self._lib.cmi_gpu_reset_grid(self._h))

    def shoot(self, seed, iteration, first_packet, n_packets):
        self._check(self._lib.cmi_gpu_shoot(self._h, seed, iteration,
                                            first_packet, n_packets))

    def get_counters(self):
        tw = C.c_double()
        tc = np.zeros(NTYPE)
        ns = C.c_uint64()
        self._check(self._lib.cmi_gpu_get_counters(self._h, C.byref(tw),
                                                   _p(tc), C.byref(ns)))
        return tw.value, tc, ns.value

    def get_atomic_count(self):
        n = C.c_uint64()
        self._check(self._lib.cmi_gpu_get_atomic_count(self._h, C.byref(n)))
        return n.value

    def get_launch_times(self):
        """[(ms, flights)] of the transport launches since the last
        get_timing(reset=True)."""
        n = C.c_uint64()
        self._check(self._lib.cmi_gpu_get_launch_times(self._h, 0, None, None,
                                                       C.byref(n)))
        ms = np.zeros(max(n.value, 1))
        pk = np.zeros(max(n.value, 1), dtype=np.uint64)
        self._check(self._lib.cmi_gpu_get_launch_times(
            self._h, n.value, _p(ms),
            pk.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(n)))
        return list(zip(ms[:n.value].tolist(), pk[:n.value].tolist()))

    def get_launch_steps(self):
        """The DDA step counter (steps since the last reset_grid) after each
        transport launch since the last get_timing(reset=True)."""
        n = C.c_uint64()
        self._check(self._lib.cmi_gpu_get_launch_steps(self._h, 0, None,
                                                       C.byref(n)))
        st = np.zeros(max(n.value, 1), dtype=np.uint64)
        self._check(self._lib.cmi_gpu_get_launch_steps(
            self._h, n.value, st.ctypes.data_as(C.POINTER(C.c_uint64)),
            C.byref(n)))
        return st[:n.value].tolist()

    # decomposed grids ---------------------------------------------------------
    def set_export_buffer(self, device_pointer, capacity):
        self._check(self._lib.cmi_gpu_set_export_buffer(
            self._h, device_pointer, int(capacity)))

    def get_export_count(self):
        n = C.c_uint64()
        self._check(self._lib.cmi_gpu_get_export_count(self._h, C.byref(n)))
        return n.value

    def reset_exports(self):
        self._check(self._lib.cmi_gpu_reset_exports(self._h))

    def shoot_flights(self, seed, iteration, first_packet, device_pointer, n):
        self._check(self._lib.cmi_gpu_shoot_flights(
            self._h, seed, iteration, int(first_packet), device_pointer,
            int(n)))

    def get_wave_steps(self):
        n = C.c_uint64()
        self._check(self._lib.cmi_gpu_get_wave_steps(self._h, C.byref(n)))
        return n.value

    def update_cells(self, loop, totweight):
        self._check(self._lib.cmi_gpu_update_cells(self._h, loop, totweight))

    def update_cells_range(self, loop, totweight, first_cell, ncell):
        self._check(self._lib.cmi_gpu_update_cells_range(
            self._h, loop, totweight, first_cell, ncell))

    def compute_emissivities(self, lines=None, first_cell=0, ncell=None):
        """EmissivityCalculator::calculate_emissivities
        (src/EmissivityCalculator.cpp:439-470) for the cells [first_cell,
        first_cell + ncell): {line name: array over those cells}. `lines` are
        names from EMISSION_LINES (all of them by default)."""
        names = list(EMISSION_LINES if lines is None else lines)
        idx = np.array([EMISSION_LINES.index(n) for n in names],
                       dtype=np.int32)
        if ncell is None:
            ncell = self.n - first_cell
        out = np.empty((len(names), ncell))
        self._check(self._lib.cmi_gpu_compute_emissivities(
            self._h, len(names), idx.ctypes.data_as(C.POINTER(C.c_int32)),
            first_cell, ncell, out.ctypes.data_as(C.POINTER(C.c_double))))
        return dict(zip(names, out))

    def set_spectrum_trackers(self, positions, nbins=100, opening_angles=None,
                              reference_directions=None):
        """SpectrumTrackers in the cells that hold `positions` ([n][3])."""
        pos = _f64(positions).reshape(-1, 3)
        n = len(pos)
        ang = None if opening_angles is None else _f64(opening_angles)
        ref = None if reference_directions is None else \
            _f64(reference_directions).reshape(-1, 3)
        self._check(self._lib.cmi_gpu_set_spectrum_trackers(
            self._h, n, _p(pos) if n else None, nbins,
            None if ang is None else _p(ang),
            None if ref is None else _p(ref)))
        self._trackers = (n, nbins)

    def set_trackers(self, positions, kinds, nbins=100, opening_angles=None,
                     reference_directions=None):
        """Trackers of the given kinds (TRACKER_SPECTRUM / TRACKER_ABSORPTION)
        in the cells that hold `positions` ([n][3]); nbins: one number for
        all, or one per tracker."""
        pos = _f64(positions).reshape(-1, 3)
        n = len(pos)
        kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        assert kinds.size == n
        bins = np.ascontiguousarray(
            np.broadcast_to(np.asarray(nbins, dtype=np.int32), (n,)))
        ang = None if opening_angles is None else _f64(opening_angles)
        ref = None if reference_directions is None else \
            _f64(reference_directions).reshape(-1, 3)
        self._check(self._lib.cmi_gpu_set_trackers(
            self._h, n, _p(pos) if n else None,
            kinds.ctypes.data_as(C.POINTER(C.c_int32)),
            bins.ctypes.data_as(C.POINTER(C.c_int32)),
            None if ang is None else _p(ang),
            None if ref is None else _p(ref)))
        self._trackers = (n, bins.tolist())

    def get_tracker_absorption(self):
        """absorption[tracker][photon type (4)][ion (14)]: the sums of an
        AbsorptionTracker (zero rows for spectrum trackers)"""
        n, _ = self._trackers
        out = np.zeros((n, 4, NION))
        self._check(self._lib.cmi_gpu_get_tracker_absorption(self._h, _p(out)))
        return out

    def set_tracker_frequency_bins(self, tracker, kind="Linear",
                                   minimum_frequency=0., maximum_frequency=0.):
        """FrequencyBins of a weighted spectrum tracker: "Linear" between
        the two frequencies (Hz) or "Level" (one bin per ion)."""
        self._check(self._lib.cmi_gpu_set_tracker_frequency_bins(
            self._h, tracker, {"Linear": 0, "Level": 1}[kind],
            minimum_frequency, maximum_frequency))

    def get_tracker_flux(self):
        """flux[tracker][photon type (4)][bin]: the sums of the weighted
        spectrum trackers (zeros for the other kinds); a list of [4][bins]
        arrays"""
        n, nbins = self._trackers
        bins = [nbins] * n if np.isscalar(nbins) else list(nbins)
        flat = np.zeros(4 * max(sum(bins), 1))
        self._check(self._lib.cmi_gpu_get_tracker_flux(self._h, _p(flat)))
        out, at = [], 0
        for b in bins:
            out.append(flat[at:at + 4 * b].reshape(4, b))
            at += 4 * b
        return out

    def enable_trackers(self, on=True):
        self._check(self._lib.cmi_gpu_enable_trackers(self._h, int(on)))

    def get_tracker_counts(self):
        """counts[tracker][type (primary, diffuse H, diffuse He)][bin]: one
        array when all trackers have the same number of bins, else a list of
        [3][bins] arrays."""
        n, nbins = self._trackers
        bins = [nbins] * n if np.isscalar(nbins) else list(nbins)
        flat = np.zeros(3 * max(sum(bins), 1), dtype=np.uint64)
        self._check(self._lib.cmi_gpu_get_tracker_counts(
            self._h, flat.ctypes.data_as(C.POINTER(C.c_uint64))))
        out, at = [], 0
        for b in bins:
            out.append(flat[at:at + 3 * b].reshape(3, b))
            at += 3 * b
        if len(set(bins)) <= 1:
            return np.array(out).reshape(n, 3, bins[0] if bins else 0)
        return out

    def set_tuning(self, **kw):
        for k, v in kw.items():
            self._check(self._lib.cmi_gpu_set_tuning(self._h, k.encode(),
                                                     int(v)))

    def synchronize(self):
        self._check(self._lib.cmi_gpu_synchronize(self._h))

    # probes -------------------------------------------------------------------
    def emit_packets(self, seed, iteration, first_packet, n):
        pos = np.empty((n, 3))
        dirn = np.empty((n, 3))
        nu = np.empty(n)
        sig = np.empty((n, NION))
        tau = np.empty(n)
        self._check(self._lib.cmi_gpu_emit_packets(
            self._h, seed, iteration, first_packet, n, _p(pos), _p(dirn),
            _p(nu), _p(sig), _p(tau)))
        return pos, dirn, nu, sig, tau

    def trace_packets(self, position, direction, tau, sigma_H, sigma_He_corr,
                      max_steps):
        pos = _f64(position).reshape(-1, 3)
        n = pos.shape[0]
        dirn = _f64(direction).reshape(n, 3)
        tau = _f64(tau).reshape(n)
        sh = _f64(sigma_H).reshape(n)
        she = _f64(sigma_He_corr).reshape(n)
        cells = np.empty((n, max_steps), dtype=np.int64)
        ds = np.empty((n, max_steps))
        nsteps = np.empty(n, dtype=np.int32)
        last = np.empty(n, dtype=np.int64)
        final = np.empty((n, 3))
        self._check(self._lib.cmi_gpu_trace_packets(
            self._h, n, _p(pos), _p(dirn), _p(tau), _p(sh), _p(she), max_steps,
            cells.ctypes.data_as(C.POINTER(C.c_int64)), _p(ds),
            nsteps.ctypes.data_as(C.POINTER(C.c_int32)),
            last.ctypes.data_as(C.POINTER(C.c_int64)), _p(final)))
        return cells, ds, nsteps, last, final

    def sample_spectrum(self, kind, temperature, seed, n):
        out = np.empty(n)
        self._check(self._lib.cmi_gpu_sample_spectrum(
            self._h, kind, temperature, seed, n, _p(out)))
        return out

    def thermal_probe(self, solve, J, heating, temperature, number_density):
        J = _f64(J).reshape(-1, NION)
        n = J.shape[0]
        heating = _f64(heating).reshape(n, 2)
        T = _f64(temperature).reshape(n)
        dens = _f64(number_density).reshape(n)
        x = np.empty((n, NION))
        Tout = np.empty(n)
        pair = np.empty((n, 2))
        self._check(self._lib.cmi_gpu_thermal_probe(
            self._h, n, int(solve), _p(J), _p(heating), _p(T), _p(dens),
            _p(x), _p(Tout), _p(pair)))
        return x, Tout, pair

    def physics_probe(self, kind, rows):
        """Atomic-data functions on the device for the given input rows; kind
        0 cross sections (nu), 1 recombination rates (T), 2 line cooling
        ({T, n_e, 13 abundances}), 3 re-emission probabilities (T), 4 charge
        transfer rates (T4)."""
        width_in = (1, 1, 15, 1, 1)[kind]
        width_out = (14, 14, 1, 5, 42)[kind]
        rows = _f64(rows).reshape(-1, width_in)
        out = np.zeros((rows.shape[0], width_out))
        self._check(self._lib.cmi_gpu_physics_probe(
            self._h, kind, rows.shape[0], _p(rows), _p(out)))
        return out

    def get_timing(self, reset=True):
        s = C.c_double()
        u = C.c_double()
        k = C.c_double()
        ns = C.c_uint64()
        nu = C.c_uint64()
        nk = C.c_uint64()
        self._check(self._lib.cmi_gpu_get_kernel_timing(
            self._h, C.byref(k), C.byref(nk)))
        self._check(self._lib.cmi_gpu_get_timing(
            self._h, int(reset), C.byref(s), C.byref(ns), C.byref(u),
            C.byref(nu)))
        return {"shoot_ms": s.value, "shoot_launches": ns.value,
                "update_ms": u.value, "update_launches": nu.value,
                "kernel_ms": k.value, "kernel_launches": nk.value}
