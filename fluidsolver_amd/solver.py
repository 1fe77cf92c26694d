"""Host-side mirror of the reference's solver API over the C ABI (single GPU).

Class and method names follow fluidSolver/Models/FluidSolver.swift and ParticleSystem.swift so a
caller of the Swift classes finds the same surface:

    FluidSolver:    setResolution, reset, applyPreset, addForce(at:windowSize:), endInteraction,
                    update, texture, getVelocityAt + the @Published parameters (:89-233)
    ParticleSystem: addParticles(at:count:windowSize:), update(fluidSolver:windowSize:), reset

All compute happens in libfluidsolver_hip.so; this file only holds host state (parameters, the
previous touch position, spawn RNG) exactly where the Swift classes hold it.
"""
import ctypes as C
import time as _time

import numpy as np

from . import lib as L
from .presets import PRESETS_BY_NAME, palette_color

F = np.float32


class FluidSolver:
    """Mirror of class FluidSolver (FluidSolver.swift:17-673) on one MI355X."""

    DISPLAY_MODES = {"Color": L.DISPLAY_DENSITY, "Motion": L.DISPLAY_VELOCITY, "Speed": L.DISPLAY_SPEED,
                     "Vorticity": L.DISPLAY_VORTICITY, "Particles Only": L.DISPLAY_PARTICLES_ONLY,
                     None: L.DISPLAY_NONE}

    def __init__(self, gridWidth=512, gridHeight=None, device=0, resolutionScale=1, max_particles=0):
        self._lib = L.load()
        self._ctx = None
        self.device = device
        self.max_particles = int(max_particles)
        # @Published parameters with the defaults of FluidSolver.swift:139-233
        self.viscosity = 0.00015
        self.diffusion = 0.0
        self.deltaTime = 0.5
        self.vorticityStrength = 0.0
        self.fadeSpeed = 0.002
        self.solverIterations = 10
        self.forceMultiplier = 24.0
        self.colorMultiplier = 100.0
        self.brightness = 0.06
        self.forceRadius = 5.0
        self.displayMode = "Color"
        self.colorPalette = "Rainbow"
        self.resolutionScale = int(resolutionScale)
        self.gridWidth = int(gridWidth)
        self.gridHeight = int(gridHeight if gridHeight is not None else gridWidth)
        self._previousTouchPosition = None   # :209
        self._lastInputVelocity = np.zeros(2, F)  # :637
        self._setupTextures()

    # -- grid sizing, FluidSolver.swift:99-106 ------------------------------------------
    @property
    def scaledWidth(self):
        return max(32, self.gridWidth // self.resolutionScale)

    @property
    def scaledHeight(self):
        return max(32, self.gridHeight // self.resolutionScale)

    def _setupTextures(self):  # :322-343
        self.close()
        ctx = C.c_void_p()
        L.check(self._lib.fs_create(self.scaledWidth, self.scaledHeight, self.device, self.max_particles,
                                    C.byref(ctx)))
        self._ctx = ctx

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.fs_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setResolution(self, resolution):  # :252-255 -> reinitialize() :260-263
        if resolution != self.gridWidth or resolution != self.gridHeight:
            self.gridWidth = self.gridHeight = int(resolution)
            self._setupTextures()
            self.reset()

    def setResolutionScale(self, scale):  # didSet of resolutionScale, :89-96
        if scale != self.resolutionScale:
            self.resolutionScale = int(scale)
            self._setupTextures()
            self.reset()

    def reset(self):  # :371-374
        L.check(self._lib.fs_reset(self._ctx), self._ctx)
        self._previousTouchPosition = None

    def applyPreset(self, preset):  # :382-399 (resolution is deliberately not changed)
        if isinstance(preset, str):
            preset = PRESETS_BY_NAME[preset]
        self.viscosity = preset.viscosity
        self.diffusion = preset.diffusion
        self.deltaTime = preset.deltaTime
        self.vorticityStrength = preset.vorticityStrength
        self.fadeSpeed = preset.fadeSpeed
        self.solverIterations = preset.solverIterations
        self.forceMultiplier = preset.forceMultiplier
        self.colorMultiplier = preset.colorMultiplier
        self.brightness = preset.brightness

    def _params(self):
        return L.Params(self.viscosity, self.diffusion, self.deltaTime, self.vorticityStrength, self.fadeSpeed,
                        int(self.solverIterations), self.DISPLAY_MODES[self.displayMode], self.brightness)

    # -- interaction, :409-433 -------------------------------------------------------------
    def addForce(self, at, windowSize, time=None):
        """addForce(at:windowSize:).  `time` (0..1) replaces the wall-clock palette phase of :422-423
        when a deterministic colour is wanted."""
        pos = np.array([F(at[0] / windowSize[0]) * F(self.scaledWidth),
                        F(at[1] / windowSize[1]) * F(self.scaledHeight)], dtype=F)       # :410-413
        force = np.zeros(2, F)
        if self._previousTouchPosition is not None:
            force = (pos - self._previousTouchPosition) * F(self.forceMultiplier)         # :417
            self._lastInputVelocity = force
        if time is None:
            frameNumber = int((_time.time() - 978307200.0) * 60)                          # :422
            time = F(frameNumber % 360) / F(360.0)                                        # :423
        color = palette_color(self.colorPalette, time) * F(self.colorMultiplier)          # :424
        self.applyForce(pos, force, color)
        self._previousTouchPosition = pos

    def endInteraction(self):  # :431-433
        self._previousTouchPosition = None

    def applyForce(self, position, force, color):  # :444-470, grid units
        col = (C.c_float * 4)(*[float(c) for c in (list(color) + [0.0])[:4]])
        L.check(self._lib.fs_add_force(self._ctx, float(position[0]), float(position[1]), float(force[0]),
                                       float(force[1]), col, float(self.forceRadius), float(self.deltaTime)),
                self._ctx)

    # -- the step, :477-627 ---------------------------------------------------------------
    def update(self):
        p = self._params()
        L.check(self._lib.fs_set_params(self._ctx, C.byref(p)), self._ctx)
        L.check(self._lib.fs_step(self._ctx), self._ctx)

    def sync(self):
        L.check(self._lib.fs_sync(self._ctx), self._ctx)

    def set_option(self, option, value):
        L.check(self._lib.fs_set_option(self._ctx, option, int(value)), self._ctx)

    def get_option(self, option):
        """What THIS context runs with (fs_config_get reads the process-wide default a new context starts from)."""
        v = C.c_int()
        L.check(self._lib.fs_get_option(self._ctx, option, C.byref(v)), self._ctx)
        return v.value

    def stage(self, stage, n=0):
        p = self._params()
        L.check(self._lib.fs_set_params(self._ctx, C.byref(p)), self._ctx)
        L.check(self._lib.fs_stage(self._ctx, stage, int(n)), self._ctx)

    @property
    def texture(self):  # :630-632: the display texture, here as a host array [4,H,W]
        return self.read_field(L.FIELD_DISPLAY)

    def read_field(self, field):
        out = np.empty((L.FIELD_PLANES[field], self.scaledHeight, self.scaledWidth), dtype=F)
        L.check(self._lib.fs_read_field(self._ctx, field, out.ctypes.data_as(C.c_void_p), out.nbytes), self._ctx)
        return out

    def write_field(self, field, arr):
        a = np.ascontiguousarray(arr, dtype=F)
        assert a.shape == (L.FIELD_PLANES[field], self.scaledHeight, self.scaledWidth), a.shape
        L.check(self._lib.fs_write_field(self._ctx, field, a.ctypes.data_as(C.c_void_p), a.nbytes), self._ctx)

    def getVelocityAt(self, normalizedPos):
        """:648-663 done for real: nearest cell of the current velocity field (debug helper; reads the
        whole field back -- the tracer kernel does this lookup on the device)."""
        W, H = self.scaledWidth, self.scaledHeight
        x = int(max(F(0), min(F(W - 1), F(normalizedPos[0]) * F(W))))
        y = int(max(F(0), min(F(H - 1), F(normalizedPos[1]) * F(H))))
        vel = self.read_field(L.FIELD_VELOCITY)
        return vel[:, y, x] / np.array([W, H], F)

    @property
    def ctx(self):
        return self._ctx


class ParticleSystem:
    """Mirror of class ParticleSystem (ParticleSystem.swift:38-403); the update runs on the device.
    Needs a FluidSolver created with max_particles > 0."""

    maxParticles = 50000  # :53 (the device capacity is the solver's max_particles)

    def __init__(self, fluidSolver, seed=0x9E3779B9):
        self._solver = fluidSolver
        self._lib = L.load()
        self.momentum = 0.5           # :63
        self.fluidForce = 0.6         # :70
        self.fadeSpeed = 0.001        # :77 (unused by update(), as in the reference)
        self.spawnCount = 10          # :84
        self.spawnRadius = 15.0       # :91
        self.fluidDecayRate = 3.0     # :98
        self.linkToViscosity = True   # :104
        self.particleSize = 1.0       # :111
        self.isEnabled = True         # :118
        self._rng = np.random.default_rng(seed)   # Float.random is the system RNG in the reference
        self._host = [np.zeros(0, F) for _ in range(7)]   # px py vx vy life mass age

    @property
    def particleCount(self):
        return int(self._lib.fs_particles_count(self._solver.ctx))

    def set_arrays(self, px, py, vx, vy, life, mass, age):
        arrs = [np.ascontiguousarray(a, dtype=F) for a in (px, py, vx, vy, life, mass, age)]
        n = arrs[0].shape[0]
        assert all(a.shape == (n,) for a in arrs)
        self._host = arrs
        ptrs = [a.ctypes.data_as(C.c_void_p) for a in arrs]
        L.check(self._lib.fs_particles_set(self._solver.ctx, n, *ptrs), self._solver.ctx)

    def get_arrays(self):
        n = self.particleCount
        arrs = [np.empty(n, F) for _ in range(7)]
        ptrs = [a.ctypes.data_as(C.c_void_p) for a in arrs]
        L.check(self._lib.fs_particles_get(self._solver.ctx, n, *ptrs), self._solver.ctx)
        return arrs

    def addParticles(self, at, count=10, windowSize=None, seed=None):  # :186-231
        """Spawns on the device (fs_particles_spawn).  `seed` replaces the system RNG of the reference (Float.random,
        :203-212); by default every call takes the next number of this object's own seeded stream."""
        if not self.isEnabled:
            return
        if seed is None:
            seed = int(self._rng.integers(1, 2 ** 32))
        L.check(self._lib.fs_particles_spawn(self._solver.ctx, float(at[0]), float(at[1]), int(count),
                                             float(self.spawnRadius), int(seed) & 0xFFFFFFFF), self._solver.ctx)

    def _pparams(self, windowSize):
        return L.ParticleParams(self.momentum, self.fluidForce, self.fluidDecayRate, 1 if self.linkToViscosity else 0,
                                self._solver.viscosity, float(windowSize[0]), float(windowSize[1]))

    def update(self, fluidSolver=None, windowSize=None):  # :242-304
        s = fluidSolver or self._solver
        if not self.isEnabled:   # (no tracers: the library returns at once; asking for the count here would make every frame wait
            return               # for the compaction a spawn may have put off, see fs_particles_spawn)
        if windowSize is None:
            windowSize = (s.scaledWidth, s.scaledHeight)
        pp = self._pparams(windowSize)
        L.check(self._lib.fs_particles_update(s.ctx, C.byref(pp)), s.ctx)

    def stepWithSolver(self, windowSize=None, on=True):
        """Coordinator.draw calls FluidSolver.update() and then ParticleSystem.update() every frame (MetalView.swift:150-156).
        With this on, FluidSolver.update() does both: the tracer update becomes the last node of the frame's graph
        (FS_OPT_WITH_PARTICLES) instead of a launch of its own.  Call again after changing a tracer parameter."""
        s = self._solver
        if windowSize is None:
            windowSize = (s.scaledWidth, s.scaledHeight)
        pp = self._pparams(windowSize)
        L.check(self._lib.fs_set_particle_params(s.ctx, C.byref(pp)), s.ctx)
        s.set_option(L.OPT_WITH_PARTICLES, 1 if on else 0)

    def lineVertices(self, viewportSize, particleScale=None):
        """What render(in:viewportSize:) draws (:341-363 -> particleVertex, Shaders:446-476): per tracer the line
        (pos - vel * particleScale, pos) in NDC as [n, 4] = (x0, y0, x1, y1), and alpha = life as [n]."""
        n = self.particleCount
        xy, alpha = np.empty((n, 4), F), np.empty(n, F)
        scale = self.particleSize if particleScale is None else particleScale     # :356 passes particleSize
        L.check(self._lib.fs_particle_lines(self._solver.ctx, float(viewportSize[0]), float(viewportSize[1]), float(scale),
                                            xy.ctypes.data_as(C.c_void_p), alpha.ctypes.data_as(C.c_void_p)),
                self._solver.ctx)
        return xy, alpha

    def reset(self):  # :366-370
        self.set_arrays(*[np.zeros(0, F) for _ in range(7)])
