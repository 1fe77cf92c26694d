"""CPU backend for SlabSolver built on the oracle -- TEST INFRASTRUCTURE ONLY (lets the gloo tests run
the real slab/halo logic of fluidsolver_amd/slabs.py without a GPU).  Same interface as HipBackend."""
import numpy as np
import torch

from oracle_binding import Grid as OGrid, Oracle

F = np.float32


class OracleBackend:
    def __init__(self):
        self.o = Oracle()

    def alloc(self, planes, rows, pitch, device):
        return torch.zeros((planes, rows, pitch), dtype=torch.float32)

    def alloc_int(self, device):
        return torch.zeros(1, dtype=torch.int32)

    @staticmethod
    def _og(g):
        return OGrid(g.width, g.height, g.y0, g.rows, g.yb, g.ye)

    @staticmethod
    def _dense(t, W):
        return np.ascontiguousarray(t[:, :, :W].numpy())

    @staticmethod
    def _back(t, a, W):
        t[:, :, :W] = torch.from_numpy(a)

    def addForces(self, g, vel, dye, pos, force, color, radius, dt):
        W = g.width
        v, d = self._dense(vel, W), self._dense(dye, W)
        self.o.addForces(self._og(g), v[0], v[1], d[0], d[1], d[2], pos, force, color, radius, dt)
        self._back(vel, v, W)
        self._back(dye, d, W)

    def vorticity(self, g, vel, w):
        W = g.width
        v, o = self._dense(vel, W), self._dense(w, W)
        self.o.calculateVorticity(self._og(g), v[0], v[1], o[0])
        self._back(w, o, W)

    def applyVorticity(self, g, w, vel, strength, dt):
        W = g.width
        v, o = self._dense(vel, W), self._dense(w, W)
        self.o.applyVorticity(self._og(g), o[0], v[0], v[1], strength, dt)
        self._back(vel, v, W)

    def _sweeps(self, g, fn, a, b, n):
        """n single sweeps with the shrinking row ranges the blocked kernel guarantees."""
        W = g.width
        x, y = self._dense(a, W), self._dense(b, W)
        for k in range(n):
            rem = n - 1 - k
            og = OGrid(g.width, g.height, g.y0, g.rows, max(0, max(g.y0, g.yb - rem)),
                       min(g.height, min(g.y0 + g.rows, g.ye + rem)))
            for pl in range(x.shape[0]):
                fn(og, x[pl], y[pl])
            x, y = y, x
        # result is in x; report it as "in b" iff n is odd, like the ping-pong does
        if n % 2:
            self._back(b, x, W)
            self._back(a, y, W)
        else:
            self._back(a, x, W)
            self._back(b, y, W)
        return bool(n % 2)

    def diffuse_n(self, g, a, b, n, visc, dt):
        return self._sweeps(g, lambda og, s, d: self.o.diffuse(og, s, d, visc, dt), a, b, n)

    def pressure_n(self, g, div, pa, pb, n, p_is_zero):
        W = g.width
        dv = self._dense(div, W)[0]
        if p_is_zero:
            pa.zero_()
        return self._sweeps(g, lambda og, s, d: self.o.pressureSolve(og, dv, s, d), pa, pb, n)

    def project_n(self, g, vel, div, pa, pb, n):
        self.divergence(g, vel, div)
        in_b = self.pressure_n(g, div, pa, pb, n, True)
        self.subtractGradient(g, pb if in_b else pa, vel)
        return in_b

    def project_n_cfl(self, g, vel, div, pa, pb, n, out):
        from fluidsolver_amd import lib as L
        gd = L.Grid(g.width, g.height, g.pitch, g.y0, g.rows, max(g.y0, g.yb - n), min(g.y0 + g.rows, g.ye + n))
        self.divergence(gd, vel, div)
        # the gradient reads one row of the final iterate beyond [yb, ye): sweep on that range
        g1 = L.Grid(g.width, g.height, g.pitch, g.y0, g.rows, max(g.y0, g.yb - 1), min(g.y0 + g.rows, g.ye + 1))
        in_b = self.pressure_n(g1, div, pa, pb, n, True)
        self.subtractGradient(g, pb if in_b else pa, vel)
        self.max_abs(g, vel[1], out)
        return in_b

    def divergence(self, g, vel, div):
        W = g.width
        v, o = self._dense(vel, W), self._dense(div, W)
        self.o.divergence(self._og(g), v[0], v[1], o[0])
        self._back(div, o, W)

    def clear(self, g, t):
        t[:, g.yb - g.y0:g.ye - g.y0, :] = 0

    def subtractGradient(self, g, p, vel):
        W = g.width
        v, pp = self._dense(vel, W), self._dense(p, W)
        self.o.subtractGradient(self._og(g), pp[0], v[0], v[1])
        self._back(vel, v, W)

    def advect(self, g, vel, src, dst, dt, err):
        W = g.width
        v, s, d = self._dense(vel, W), self._dense(src, W), self._dense(dst, W)
        bad = 0
        for pl in range(s.shape[0]):
            bad += self.o.advect(self._og(g), v[0], v[1], s[pl], d[pl], dt)
        if bad:
            err[0] = 1
        self._back(dst, d, W)

    def fade(self, g, dye, rate):
        W = g.width
        d = self._dense(dye, W)
        for pl in range(d.shape[0]):
            self.o.fadeDensity(self._og(g), d[pl], rate)
        self._back(dye, d, W)

    def subtractGradient_cfl(self, g, p, vel, out):
        self.subtractGradient(g, p, vel)
        self.max_abs(g, vel[1], out)

    def max_abs(self, g, plane, out):
        out[0] = float(plane[g.yb - g.y0:g.ye - g.y0, :g.width].abs().max()) if g.ye > g.yb else 0.0
