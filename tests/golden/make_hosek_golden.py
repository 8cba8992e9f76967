"""Generates tests/golden/hosek_golden.npz from the REFERENCE's own code: the Hosek-Wilkie sample implementation the reference
ships (src/runtime/skysun/model/ArHosekSkyModel.cpp), compiled from where it lies by oracle/Makefile into oracle/_ref/libhosek_ref.so.
Inputs: a seeded set of (channel, turbidity, albedo, elevation, theta, gamma); output: arhosek_tristim_skymodel_radiance of the state
arhosek_rgb_skymodelstate_alloc_init builds. Run in the build container (where /root/reference exists): python tests/golden/make_hosek_golden.py"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhosek_ref.so"))
init = getattr(ref, "_Z36arhosek_rgb_skymodelstate_alloc_initddd")
init.restype = C.c_void_p
init.argtypes = [C.c_double] * 3
rad = getattr(ref, "_Z33arhosek_tristim_skymodel_radianceP20ArHosekSkyModelStateddi")
rad.restype = C.c_double
rad.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]

rng = np.random.default_rng(2012)
n = 512
turb = np.concatenate([rng.uniform(1, 10, n - 6), [1.0, 2.0, 3.0, 9.5, 10.0, 10.0]])
alb = np.concatenate([rng.uniform(0, 1, n - 6), [0.0, 1.0, 0.8, 0.3, 0.0, 1.0]])
elev = rng.uniform(0.0, np.pi / 2, n)
theta = rng.uniform(0.0, np.pi / 2 * 0.999, n)
gamma = rng.uniform(0.0, np.pi, n)
chan = rng.integers(0, 3, n)
out = np.empty(n)
for i in range(n):
    st = init(turb[i], alb[i], elev[i])
    out[i] = rad(st, theta[i], gamma[i], int(chan[i]))
np.savez(os.path.join(ROOT, "tests", "golden", "hosek_golden.npz"), channel=chan.astype(np.int32), turbidity=turb, albedo=alb, elevation=elev,
         theta=theta, gamma=gamma, radiance=out)
print("wrote", n, "vectors; radiance range", out.min(), out.max())
