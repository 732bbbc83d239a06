"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/c18_windsea_swell_profile.npz from the REFERENCE ITSELF:
`stokes_drift_profile_windsea_swell` (models/physics_methods.py:418-456; Breivik & Christensen 2020), the function
`stokes_drift` calls for drift:stokes_drift_profile = 'windsea_swell' (:831-841), evaluated on float32 environment-like
arrays (what self.environment holds) and float64 depths.  No stock model of the reference lists the six swell / wind-sea
variables among its required_variables, so a model run with that profile cannot be generated from the reference; the
profile function is the part of the path that can be pinned.

    python oracle/gen_golden_windsea.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (installs the shim)
from opendrift.models import physics_methods as pm  # noqa: E402


def main():
    rng = np.random.default_rng(18)
    n = 4000
    f = lambda a: np.asarray(a, dtype=np.float32)
    sx, sy = f(rng.normal(0, 0.08, n)), f(rng.normal(0, 0.08, n))
    sx[:20], sy[:20] = 0, 0                                   # zero surface drift
    swell_dir, ww_dir = f(rng.uniform(0, 360, n)), f(rng.uniform(0, 360, n))
    swell_tp, swell_hs = f(rng.uniform(7, 16, n)), f(rng.uniform(0.2, 3, n))
    ww_tm, ww_hs = f(rng.uniform(2, 8, n)), f(rng.uniform(0.1, 4, n))
    z = -rng.uniform(0, 30, n)
    z[20:200] = 0.0
    with np.errstate(all='ignore'):
        u, v, s = pm.stokes_drift_profile_windsea_swell(sx, sy, swell_dir, swell_tp, swell_hs, ww_dir, ww_tm, ww_hs, z)
    print('dtype', u.dtype, 'finite', np.isfinite(u).mean(), 'max speed', np.nanmax(s))
    np.savez_compressed(os.path.join(gg.GOLD, 'c18_windsea_swell_profile.npz'), sx=sx, sy=sy, swell_dir=swell_dir,
                        swell_tp=swell_tp, swell_hs=swell_hs, ww_dir=ww_dir, ww_tm=ww_tm, ww_hs=ww_hs, z=z,
                        stokes_u=np.asarray(u, dtype=np.float64), stokes_v=np.asarray(v, dtype=np.float64))


if __name__ == '__main__':
    main()
