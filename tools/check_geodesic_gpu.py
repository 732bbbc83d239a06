import sys, numpy as np
sys.path.insert(0, '.')
from opendrift_amd.device import Context
from oracle import oracle as orc
rng = np.random.default_rng(1)
n = 20000
lon = rng.uniform(-180, 180, n); lat = rng.uniform(-89, 89, n)
speed = 10 ** rng.uniform(-4, 2.5, n); ang = rng.uniform(-np.pi, np.pi, n)
u, v = speed * np.sin(ang), speed * np.cos(ang)
ctx = Context(0)
P = ctx.particles(n); P.append(lon, lat)
P.update_positions(u, v, 3600.0)
got = P.download()
lo, la = lon.copy(), lat.copy()
orc.update_positions(lo, la, u, v, np.ones(n, np.int32), 3600.0)
d = np.abs(got['lon'] - lo); d = np.minimum(d, 360 - d)
idx = np.argsort(d)[-8:]
for i in idx:
    print(i, 'lon %.6f lat %.6f az %.6f dist %.3f -> dlon %.3e dlat %.3e  got lat %.8f' % (lon[i], lat[i], np.degrees(np.arctan2(u[i], v[i])), speed[i]*3600, d[i], got['lat'][i]-la[i], got['lat'][i]))
print('frac > 1e-11:', np.mean(d > 1e-11), 'lat max', np.abs(got['lat']-la).max())
moving = (rng.uniform(size=n) > 0.05).astype(np.int32)
for dtype, dt in ((np.float64, 3600.0), (np.float32, 900.0), (np.float64, -3600.0)):
    P = ctx.particles(n); P.append(lon, lat, moving=moving)
    uu, vv = u.astype(dtype), v.astype(dtype)
    P.update_positions(uu, vv, dt)
    got = P.download()
    lo, la = lon.copy(), lat.copy()
    orc.update_positions(lo, la, uu, vv, moving, dt)
    d = np.abs(got['lon'] - lo); d = np.minimum(d, 360 - d)
    i = np.argmax(d)
    print(dtype.__name__, dt, 'max dlon', d.max(), 'n>1e-11', (d > 1e-11).sum(), 'worst: moving', moving[i], 'lat', lat[i], 'u,v', uu[i], vv[i], 'dlat', got['lat'][i]-la[i])
    bad = np.nonzero(d > 1e-11)[0][:5]
    for i in bad: print('   ', i, moving[i], lon[i], lat[i], uu[i], vv[i], d[i])
