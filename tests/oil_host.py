"""TEST INFRASTRUCTURE: ctypes access to the per-element arithmetic of opendrift_amd/csrc/odr_oil.hip.h compiled for the
host (g++ -ffp-contract=off), see oil_host.cpp."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'oracle', '_build', 'oil_host.so')
SRC = [os.path.join(HERE, 'oil_host.cpp'), os.path.join(ROOT, 'opendrift_amd', 'csrc', 'odr_oil.hip.h')]
_dp, _fp, _lp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_longlong)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(OUT) or any(os.path.getmtime(OUT) < os.path.getmtime(s) for s in SRC):
            os.makedirs(os.path.dirname(OUT), exist_ok=True)
            subprocess.check_call(['g++', '-O1', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', '-o', OUT, SRC[0]])
        _lib = C.CDLL(OUT)
    return _lib


def _f(a, n):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float32), (n,)))


def elements(x_wind, y_wind, temperature, salinity, diameter, density, viscosity, film, diameter_if_entrained,
             interfacial_tension, sea_water_density, dt_mix, droplets, hs=None, tp=None, hs_mode=1, tp_mode=3, to_kelvin=True):
    n = len(x_wind)
    arrs = [_f(a, n) for a in (x_wind, y_wind, temperature, salinity, 0 if hs is None else hs, 0 if tp is None else tp,
                               diameter, density, viscosity, film, diameter_if_entrained)]
    out = dict(prob=np.empty(n), w_now=np.empty(n), w_if=np.empty(n), nyw=np.empty(n, np.float32), dv50=np.empty(n),
               zb=np.empty(n, np.float32))
    lib().oilh_elements(C.c_longlong(n), *[a.ctypes.data_as(_fp) for a in arrs], C.c_double(interfacial_tension),
                        C.c_double(sea_water_density), C.c_double(dt_mix), droplets, hs_mode, tp_mode, int(to_kelvin),
                        out['prob'].ctypes.data_as(_dp), out['w_now'].ctypes.data_as(_dp), out['w_if'].ctypes.data_as(_dp),
                        out['nyw'].ctypes.data_as(_fp), out['dv50'].ctypes.data_as(_dp), out['zb'].ctypes.data_as(_fp))
    return out


def choice(dv50, uniforms):
    u = np.ascontiguousarray(uniforms, dtype=np.float64)
    d, idx = np.empty(len(u)), np.empty(len(u), np.int64)
    lib().oilh_choice(C.c_double(dv50), C.c_longlong(len(u)), u.ctypes.data_as(_dp), d.ctypes.data_as(_dp),
                      idx.ctypes.data_as(_lp))
    return d, idx
