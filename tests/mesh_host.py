"""TEST INFRASTRUCTURE: ctypes access to opendrift_amd/csrc/odr_mesh.h compiled for the host (g++), see mesh_host.cpp."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'oracle', '_build', 'mesh_host.so')
SRC = [os.path.join(HERE, 'mesh_host.cpp'), os.path.join(ROOT, 'opendrift_amd', 'csrc', 'odr_mesh.h')]
_dp = C.POINTER(C.c_double)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(OUT) or any(os.path.getmtime(OUT) < os.path.getmtime(s) for s in SRC):
            os.makedirs(os.path.dirname(OUT), exist_ok=True)
            subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-o', OUT, SRC[0]])
        _lib = C.CDLL(OUT)
        _lib.mesh_build.restype = C.c_void_p
    return _lib


class HostMesh:
    def __init__(self, lon2d, lat2d):
        lo = np.ascontiguousarray(lon2d, dtype=np.float64)
        la = np.ascontiguousarray(lat2d, dtype=np.float64)
        fl, nt, err = C.c_longlong(), C.c_longlong(), C.create_string_buffer(256)
        h = lib().mesh_build(lo.ctypes.data_as(_dp), la.ctypes.data_as(_dp), lo.shape[0], lo.shape[1], C.byref(fl),
                             C.byref(nt), err, 256)
        if not h:
            raise ValueError(err.value.decode())
        self.h, self.flips, self.ntri = C.c_void_p(h), fl.value, nt.value

    def triangles(self):
        v = np.empty((self.ntri, 3), np.int32)
        n = np.empty((self.ntri, 3), np.int32)
        lib().mesh_tris(self.h, v.ctypes.data_as(C.c_void_p), n.ctypes.data_as(C.c_void_p))
        return v, n

    def locate(self, lon, lat):
        lon = np.ascontiguousarray(lon, dtype=np.float64)
        lat = np.ascontiguousarray(lat, dtype=np.float64)
        x, y = np.empty(len(lon)), np.empty(len(lon))
        lib().mesh_locate(self.h, C.c_longlong(len(lon)), lon.ctypes.data_as(_dp), lat.ctypes.data_as(_dp),
                          x.ctypes.data_as(_dp), y.ctypes.data_as(_dp))
        return x, y

    def __del__(self):
        if getattr(self, 'h', None):
            lib().mesh_free(self.h)
            self.h = None


def meshes():
    """Synthetic curvilinear node arrays: name -> (lon2d, lat2d, needs_flips)."""
    X, Y = np.meshgrid(np.arange(40.), np.arange(30.))
    c, s = np.cos(.35), np.sin(.35)
    return {
        'rotated_stretched': (5 + 0.04 * (np.cos(.5) * X - np.sin(.5) * Y), 60 + 0.02 * (np.sin(.5) * X + np.cos(.5) * Y), False),
        'bent_sheared': (5 + 0.03 * X + 0.0004 * Y * Y, 60 + 0.02 * Y + 0.0003 * X * X - 0.00001 * X * Y, True),
        'rectilinear': (5 + 0.03 * X, 60 + 0.02 * Y, False),
        'left_handed': (5 - 0.03 * X + 0.0002 * Y * Y, 60 + 0.02 * Y, False),
        # a metric grid rotated 20 degrees against the meridians at 70N: the 1/cos(lat) stretch of longitude shears it
        'rot20_at_70N': ((c * X - s * Y) * 0.01 / 0.34 + 10, 70 + 0.01 * (s * X + c * Y), True),
    }
