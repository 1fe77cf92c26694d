"""Is the three-instruction fp32 division by a launch constant (Markstein form) exact?  Exhaustive over all 2^32 x for the
divisors 1 + 4a the diffuse sweep meets (a = dt * visc * W * H for the presets' viscosities and the benchmark grids) and a
few hundred random ones.  Probe build: python -m fluidsolver_amd.build --probe; FS_PROBE_LIB=1 python scripts/probe_fast_division.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FS_PROBE_LIB"] = "1"
from fluidsolver_amd import lib as L
lib = C.CDLL(L.LIB_PATH)
lib.fs_probe_divide_fast.argtypes = [C.c_float, C.POINTER(C.c_ulonglong)]
F = np.float32
dens = []
for visc in (1.5e-4, 1e-5, 1e-4, 5e-5, 2e-4, 1e-3, 1e-2):
    for W, H in ((3072, 3072), (256, 256), (512, 512), (16384, 16384), (8192, 8192), (1024, 768)):
        a = F(F(F(F(0.5) * F(visc)) * F(W)) * F(H))
        dens.append((float(F(1.0) + F(4.0) * a), "visc %g, %dx%d" % (visc, W, H)))
rng = np.random.default_rng(1)
n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dens += [(float(F(1.0) + F(4.0) * F(x)), "random a") for x in np.exp(rng.uniform(np.log(1e-3), np.log(1e5), n_rand))]
bad_dens = 0
for den, what in dens:
    out = (C.c_ulonglong * 3)()
    rc = lib.fs_probe_divide_fast(den, out)
    assert rc == 0, rc
    if out[0] or what != "random a":
        print("den = %-14.9g (%s): %d of 2^32 x differ from the IEEE quotient inside [2^-100, 2^100], %d outside%s"
              % (den, what, out[0], out[1], "; e.g. x = bits 0x%08x" % out[2] if out[0] else ""), flush=True)
    bad_dens += 1 if out[0] else 0
print("%d of %d divisors have x inside the safe exponent range for which the three-instruction form is NOT the correctly rounded quotient"
      % (bad_dens, len(dens)))
