"""One sequence of C-ABI calls for two libraries: the engine
(cmacionize_amd/libcmi_gpu.so, entry points cmi_gpu_*) and its CPU twin
(oracle/libcmi_cpu.so, cmi_cpu_*: test infrastructure on top of the oracle).
Plain ctypes, typed from include/cmi_gpu.h - no use of the package's binding,
so that what is compared is the ABI itself."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINE = os.path.join(ROOT, "cmacionize_amd", "libcmi_gpu.so")
TWIN = os.path.join(ROOT, "oracle", "libcmi_cpu.so")
PC = 3.086e16
dp = C.POINTER(C.c_double)
vp = C.c_void_p


class Config(C.Structure):  # cmi_gpu_config
    _fields_ = [("anchor", C.c_double * 3), ("sides", C.c_double * 3),
                ("ncell", C.c_int32 * 3), ("periodic", C.c_int32 * 3),
                ("device", C.c_int32), ("track_heating", C.c_int32),
                ("stream", C.c_void_p),
                ("external_accumulators", C.c_void_p),
                ("sub_offset", C.c_int32 * 3), ("sub_ncell", C.c_int32 * 3)]


class TemperatureParams(C.Structure):  # cmi_gpu_temperature_params
    _fields_ = [("do_temperature_calculation", C.c_int32),
                ("minimum_number_of_iterations", C.c_int32),
                ("epsilon_convergence", C.c_double),
                ("maximum_number_of_iterations", C.c_int32),
                ("pah_heating_factor", C.c_double),
                ("cosmic_ray_heating_factor", C.c_double),
                ("cosmic_ray_heating_limit", C.c_double),
                ("cosmic_ray_heating_scale_length", C.c_double),
                ("minimum_ionized_temperature", C.c_double)]


# name -> argument types after the handle (include/cmi_gpu.h)
CALLS = {
    "set_sources": [C.c_int32, dp, dp, C.c_double],
    "set_spectrum_monochromatic": [C.c_double],
    "set_spectrum_planck": [C.c_double],
    "set_cross_sections_fixed": [dp],
    "set_cross_sections_verner": [],
    "set_recombination_rates_fixed": [dp],
    "set_recombination_rates_verner": [],
    "set_abundances": [dp],
    "set_reemission": [C.c_int32, C.c_double, C.c_double],
    "set_temperature_params": [C.POINTER(TemperatureParams)],
    "upload_cells": [dp, dp, dp],
    "upload_field": [C.c_int32, dp],
    "download_field": [C.c_int32, dp],
    "reset_grid": [],
    "shoot": [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64],
    "get_counters": [dp, dp, C.POINTER(C.c_uint64)],
    "update_cells": [C.c_uint32, C.c_double],
    "update_cells_range": [C.c_uint32, C.c_double, C.c_int64, C.c_int64],
    "synchronize": [],
    "destroy": [],
}


def _p(a):
    return a.ctypes.data_as(dp)


class AbiError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "%d: %s" % (code, message))
        self.code = code


class Abi:
    """The core entry points of one library, by their ABI names without the
    prefix; every call checks the return code."""

    def __init__(self, path, prefix, ncell, track_heating=True):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        for name, args in CALLS.items():
            f = getattr(self.lib, prefix + name)
            f.argtypes = [vp] + args
            f.restype = C.c_int
        self._last_error = getattr(self.lib, prefix + "last_error")
        self._last_error.restype = C.c_char_p
        create = getattr(self.lib, prefix + "create")
        create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        create.restype = C.c_int
        cfg = Config()
        for a in range(3):
            cfg.anchor[a] = -5. * PC
            cfg.sides[a] = 10. * PC
            cfg.ncell[a] = ncell
        cfg.track_heating = int(track_heating)
        self.h = vp()
        self.n = ncell ** 3
        rc = create(C.byref(cfg), C.byref(self.h))
        if rc:
            raise AbiError(rc, self._last_error().decode())

    def call(self, name, *args):
        rc = getattr(self.lib, self.prefix + name)(self.h, *args)
        if rc:
            raise AbiError(rc, self._last_error().decode())

    def field(self, field):
        out = np.empty(self.n)
        self.call("download_field", field, _p(out))
        return out

    def counters(self):
        tw = C.c_double()
        tc = np.zeros(4)
        ns = C.c_uint64()
        self.call("get_counters", C.byref(tw), _p(tc), C.byref(ns))
        return tw.value, tc

    def close(self):
        if self.h:
            self.call("destroy")
            self.h = vp()


def engine(ncell, **kw):
    return Abi(ENGINE, "cmi_gpu_", ncell, **kw)


def twin(ncell, **kw):
    return Abi(TWIN, "cmi_cpu_", ncell, **kw)


def run_benchmark(abi, kind, ncell, npacket, iterations, seed=42):
    """benchmarks/{stromgren, stromgren_diffuse, lexingtonHII40}.param at
    ncell^3 through the ABI: the set-up calls, then `iterations` times reset /
    shoot / counters / update. Returns per iteration (totweight, typecount,
    J_H, J_He, heating_H) and the final (x_H, x_He, T)."""
    n = ncell ** 3
    src = np.zeros(3)
    one = np.ones(1)
    abi.call("set_sources", 1, _p(src), _p(one), 4.26e49)
    x = np.zeros((14, n))
    x[0] = 1.e-6
    x[1] = 1.e-6
    if kind == "lexington":
        abi.call("set_spectrum_planck", 40000.)
        abi.call("set_cross_sections_verner")
        abi.call("set_recombination_rates_verner")
        ab = np.array([0.1, 2.2e-4, 4.e-5, 3.3e-4, 5.e-5, 9.e-6])
        abi.call("set_abundances", _p(ab))
        abi.call("set_reemission", 1, 0., 0.)
        tp = TemperatureParams(1, 3, 1.e-3, 100, 0., 0., 0.75,
                               1.33333 * 3.086e19, 4000.)
        abi.call("set_temperature_params", C.byref(tp))
        ax = -5. * PC + (np.arange(ncell) + 0.5) * (10. * PC / ncell)
        X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
        gas = np.sqrt(X * X + Y * Y + Z * Z).ravel() > 3.e16
        dens = np.where(gas, 1.e8, 0.)
        temp = np.where(gas, 8000., 0.)
    else:
        abi.call("set_spectrum_monochromatic",
                 13.6 * 1.6021766208e-19 * (1. / 6.626070040e-34) / 1.)
        sigma = np.zeros(14)
        sigma[0] = 6.3e-18 * 1.e-4
        alpha = np.zeros(14)
        alpha[0] = 4.e-13 * 1.e-6
        abi.call("set_cross_sections_fixed", _p(sigma))
        abi.call("set_recombination_rates_fixed", _p(alpha))
        if kind == "stromgren_diffuse":
            abi.call("set_reemission", 1, 0., 0.)
        dens = np.full(n, 1.e8)
        temp = np.full(n, 8000.)
    abi.call("upload_cells", _p(dens), _p(temp), _p(x))
    steps = []
    for loop in range(iterations):
        abi.call("reset_grid")
        abi.call("shoot", seed, loop, 0, npacket)
        tw, tc = abi.counters()
        steps.append((tw, tc, abi.field(16), abi.field(17), abi.field(30)))
        abi.call("update_cells", loop, tw)
        abi.call("synchronize")
    return steps, (abi.field(2), abi.field(3), abi.field(1))
