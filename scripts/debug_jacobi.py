import ctypes as C, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_binding import Oracle, full_grid
from gpu_util import dev_planes, grid, host_planes, ptr, stream, sync
from fluidsolver_amd import lib as L
lib = L.load(); o = Oracle(); F = np.float32
for (W, H) in [(256, 256), (177, 131)]:
    for n in (1, 2, 8):
        for pz in (0, 1):
            rng = np.random.default_rng(10)
            div = rng.uniform(-1, 1, (H, W)).astype(F); p0 = rng.uniform(-2, 2, (H, W)).astype(F)
            a = (np.zeros((H, W), F) if pz else p0.copy()); b = np.zeros((H, W), F)
            for _ in range(n):
                o.pressureSolve(full_grid(W, H), div, a, b); a, b = b, a
            g = grid(W, H); dd = dev_planes(div); pa = dev_planes(p0); pb = dev_planes(np.full((H, W), 7.0, F))
            in_b = C.c_int(-1)
            L.check(lib.fs_k_pressureSolve_n(C.byref(g), ptr(dd), ptr(pa), ptr(pb), n, pz, C.byref(in_b), stream()))
            sync()
            got = host_planes(pb if in_b.value else pa, W)[0]
            bad = np.argwhere(got != a)
            print(W, H, "n", n, "pz", pz, "in_b", in_b.value, "nbad", len(bad))
            if len(bad):
                ys, xs = bad[:, 0], bad[:, 1]
                print("  y range", ys.min(), ys.max(), "x range", xs.min(), xs.max())
                print("  first", bad[:6].tolist(), [(float(got[y, x]), float(a[y, x])) for y, x in bad[:6]])
                print("  distinct rows", np.unique(ys)[:20].tolist(), "distinct cols", np.unique(xs)[:20].tolist())
