"""A SECOND CPU restatement of the reference's compute kernels, written independently of oracle/fluid_oracle.c:
whole-array numpy in strict fp32, straight from fluidSolver/Shaders/FluidShaders.metal (line numbers below) and the
step order of FluidSolver.update() (fluidSolver/Models/FluidSolver.swift:477-591).

TEST INFRASTRUCTURE ONLY.  The reference ships no golden vectors (parity is unpinned), so the
scalar C oracle cannot be checked against reference outputs.  What CAN be checked is that two restatements with
nothing in common but the shader text -- scalar C loops there, array slices here -- agree bit for bit
(tests/test_second_restatement.py): a transcription slip in either one (an operand order, a boundary test, a swap)
shows up as a difference.  Conventions shared with the oracle because they are readings of the language, not of this
code: fp32 everywhere, no FMA contraction, left-to-right evaluation, length(v) = sqrt(x*x + y*y), mix(a, b, t) =
a + (b - a)*t, exp evaluated in double and rounded once.  Fields are planar [H, W] float32 arrays (velocity 2
planes, dye 3).
"""
import numpy as np

F = np.float32


def _interior(H, W):
    """Shaders:91 (and :127, :169, :201, :232, :263, :297): gid.x == 0 || gid.x >= width-1 || gid.y == 0 || ..."""
    m = np.zeros((H, W), bool)
    if H > 2 and W > 2:
        m[1:-1, 1:-1] = True
    return m


def _nbr(a):
    """left, right, top, bottom of every cell (edge cells get a wrapped value that the callers mask out)"""
    return np.roll(a, 1, 1), np.roll(a, -1, 1), np.roll(a, 1, 0), np.roll(a, -1, 0)


def addForces(vel, dens, position, force, color, radius, dt):
    """Shaders:43-69, in place"""
    H, W = vel[0].shape
    radius, dt = F(radius), F(dt)
    gx = np.arange(W, dtype=F)[None, :] - F(position[0])                 # :53
    gy = np.arange(H, dtype=F)[:, None] - F(position[1])
    dist = np.sqrt(gx * gx + gy * gy)                                    # :54
    hit = dist < radius                                                  # :56
    arg = (-dist * dist) / (F(2.0) * radius * radius)                    # :57
    infl = np.exp(arg.astype(np.float64)).astype(F)
    for c in range(2):
        vel[c][hit] = (vel[c] + F(force[c]) * infl * dt)[hit]            # :61
    for c in range(3):
        dens[c][hit] = (dens[c] + F(color[c]) * infl * dt)[hit]          # :66


def diffuse(src, visc, dt):
    """Shaders:82-105, one plane, returns the output plane"""
    H, W = src.shape
    a = F(dt) * F(visc) * F(W) * F(H)                                    # :96
    l, r, t, b = _nbr(src)
    res = (src + a * (l + r + t + b)) / (F(1.0) + F(4.0) * a)            # :103
    return np.where(_interior(H, W), res, src).astype(F)                 # :91-94


def advect(u, v, planes, dt):
    """Shaders:118-151; `planes` = the planes of the input field, returns the output planes"""
    H, W = u.shape
    dt = F(dt)
    px = np.arange(W, dtype=F)[None, :] - u * dt                         # :133
    py = np.arange(H, dtype=F)[:, None] - v * dt
    px = np.minimum(np.maximum(px, F(0.5)), F(W) - F(1.5))               # :136
    py = np.minimum(np.maximum(py, F(0.5)), F(H) - F(1.5))               # :137
    fx, fy = np.floor(px), np.floor(py)                                  # :140
    sx, sy = px - fx, py - fy                                            # :142
    inside = _interior(H, W)
    # boundary cells never use their taps; give them a harmless in-range address
    x0 = np.where(inside, fx, 0).astype(np.int64)
    y0 = np.where(inside, fy, 0).astype(np.int64)
    x1 = np.where(inside, x0 + 1, 0)                                     # :141 (the clamp keeps i1 inside the grid)
    y1 = np.where(inside, y0 + 1, 0)
    out = []
    for q in planes:
        q00, q10, q01, q11 = q[y0, x0], q[y0, x1], q[y1, x0], q[y1, x1]  # :144-147
        top = q00 + (q10 - q00) * sx                                     # :149 mix(q00, q10, s.x)
        bot = q01 + (q11 - q01) * sx
        res = top + (bot - top) * sy
        out.append(np.where(inside, res, q).astype(F))                   # :127-130
    return out


def divergence(u, v):
    """Shaders:162-181"""
    H, W = u.shape
    ul, ur, _, _ = _nbr(u)
    _, _, vt, vb = _nbr(v)
    d = F(-0.5) * ((ur - ul) + (vb - vt))                                # :179
    return np.where(_interior(H, W), d, F(0)).astype(F)                  # :169-172


def pressureSolve(div, p):
    """Shaders:193-214"""
    H, W = p.shape
    l, r, t, b = _nbr(p)
    res = (div + l + r + t + b) / F(4.0)                                 # :212
    return np.where(_interior(H, W), res, F(0)).astype(F)                # :201-204


def subtractGradient(p, u, v):
    """Shaders:225-245, returns the new (u, v)"""
    H, W = p.shape
    l, r, t, b = _nbr(p)
    inside = _interior(H, W)
    return (np.where(inside, u - F(0.5) * (r - l), u).astype(F),         # :242
            np.where(inside, v - F(0.5) * (b - t), v).astype(F))         # :243


def calculateVorticity(u, v):
    """Shaders:256-275"""
    H, W = u.shape
    _, _, ut, ub = _nbr(u)
    vl, vr, _, _ = _nbr(v)
    curl = (ub - ut) - (vr - vl)                                         # :273
    return np.where(_interior(H, W), curl, F(0)).astype(F)               # :263-266


def applyVorticity(w, u, v, strength, dt):
    """Shaders:288-316, returns the new (u, v)"""
    H, W = w.shape
    l, r, t, b = _nbr(np.abs(w))                                         # :301-304
    gx, gy = r - l, b - t                                                # :307
    glen = np.sqrt(gx * gx + gy * gy) + F(0.00001)                       # :308
    gx, gy = gx / glen, gy / glen                                        # :309
    sc = F(strength) * w                                                 # :311 strength * center * float2(g.y, -g.x)
    fx, fy = sc * gy, sc * (-gx)
    inside = _interior(H, W)
    return (np.where(inside, u + fx * F(dt), u).astype(F),               # :314
            np.where(inside, v + fy * F(dt), v).astype(F))


def fadeDensity(d, fade):
    """Shaders:326-334"""
    return np.maximum(d * (F(1.0) - F(fade)), F(0.0)).astype(F)          # :331-332


def renderDensity(dens, brightness):
    """Shaders:357-366 -> 4 planes"""
    return [(d * F(brightness)).astype(F) for d in dens] + [np.ones_like(dens[0])]


def renderVelocity(u, v, scale):
    """Shaders:378-394"""
    s = F(scale)
    speed = np.sqrt(u * u + v * v)                                       # :384
    return [u * s + F(0.5), v * s + F(0.5), speed * s, np.ones_like(u)]  # :389-393


def renderSpeed(u, v, scale):
    """Shaders:405-414"""
    sp = np.sqrt(u * u + v * v) * F(scale)                               # :411
    return [sp, sp, sp, np.ones_like(u)]


class Solver:
    """FluidSolver.update() / applyForce() on zero-initialised fields (FluidSolver.swift:348-366, :444-470, :477-591)."""

    def __init__(self, W, H):
        self.W, self.H = W, H
        z = lambda n: [np.zeros((H, W), F) for _ in range(n)]            # noqa: E731
        self.vel, self.dens = z(2), z(3)
        self.p, self.div, self.w = z(1)[0], z(1)[0], z(1)[0]

    def add_force(self, pos, force, color, radius, dt):
        addForces(self.vel, self.dens, pos, force, color, radius, dt)

    def update(self, viscosity, diffusion, dt, vorticityStrength, fadeSpeed, solverIterations):
        if vorticityStrength > 0:                                        # :486-502
            self.w = calculateVorticity(*self.vel)
            self.vel = list(applyVorticity(self.w, self.vel[0], self.vel[1], vorticityStrength, dt))
        if viscosity > 0:                                                # :505-518
            for _ in range(solverIterations):
                self.vel = [diffuse(c, viscosity, dt) for c in self.vel]
        self.div = divergence(*self.vel)                                 # :522-525
        self.p = np.zeros_like(self.p)                                   # :528-530
        for _ in range(solverIterations):                                # :533-541
            self.p = pressureSolve(self.div, self.p)
        self.vel = list(subtractGradient(self.p, *self.vel))             # :544-547
        self.vel = advect(self.vel[0], self.vel[1], self.vel, dt)        # :550-558
        if diffusion > 0:                                                # :562-574
            for _ in range(solverIterations):
                self.dens = [diffuse(c, diffusion, dt) for c in self.dens]
        self.dens = advect(self.vel[0], self.vel[1], self.dens, dt)      # :577-584
        self.dens = [fadeDensity(c, fadeSpeed) for c in self.dens]       # :587-591


def particles_update(u, v, arrs, momentum, fluidForce, fluidDecayRate, linkToViscosity, viscosity, window):
    """ParticleSystem.update(), fluidSolver/Models/ParticleSystem.swift:242-304, on SoA arrays (px, py, vx, vy, life,
    mass, age), with the velocity lookup of FluidSolver.getVelocityAt (FluidSolver.swift:648-657) done for real on the
    field -- the build's definitions where the reference is a stub: fluidVel = cell velocity /
    (W, H); a dead tracer (life == 0) is skipped and stays in the arrays.  Returns the new arrays."""
    px, py, vx, vy, life, mass, age = arrs
    H, W = u.shape
    ww, wh = F(window[0]), F(window[1])
    nx, ny = px * (F(1.0) / ww), py * (F(1.0) / wh)                                        # :245-255
    cx = np.maximum(F(0), np.minimum(F(W - 1), nx * F(W))).astype(np.int64)                # Solver:650
    cy = np.maximum(F(0), np.minimum(F(H - 1), ny * F(H))).astype(np.int64)                # Solver:651
    fvx, fvy = u[cy, cx] / F(W), v[cy, cx] / F(H)
    ageDecay = np.maximum(F(0), F(1.0) - age * F(fluidDecayRate))                          # :259
    eff = F(0.1) + (F(viscosity) / F(0.01)) * F(0.9) if linkToViscosity else F(fluidForce)  # :262-268
    k = mass * eff * ageDecay                                                              # :271
    fcx, fcy = fvx * k * ww, fvy * k * wh
    em = F(momentum) + (F(1.0) - F(momentum)) * np.minimum(F(1.0), age * F(2.0))           # :272
    nvx, nvy = fcx + vx * em, fcy + vy * em                                                # :273-275
    npx, npy = px + nvx, py + nvy                                                          # :276
    nage = age + F(1.0) / F(60.0)                                                          # :279
    m = F(50)                                                                              # :282
    off = (npx < -m) | (npx > ww + m) | (npy < -m) | (npy > wh + m)                        # :283-286
    nl = np.where(off, F(0), life) * F(0.999)                                              # :287, :291
    nl = np.where(nl < F(0.01), F(0), nl)                                                  # :294-295
    alive = life != 0
    return [np.where(alive, new, old).astype(F) for new, old in
            zip((npx, npy, nvx, nvy, nl, mass, nage), (px, py, vx, vy, life, mass, age))]


def _xorshift32(x):
    x ^= (x << np.uint32(13)) & np.uint32(0xFFFFFFFF)
    x ^= x >> np.uint32(17)
    x ^= (x << np.uint32(5)) & np.uint32(0xFFFFFFFF)
    return x


def particles_spawn(cap, arrs, x, y, count, radius, seed):
    """ParticleSystem.addParticles, fluidSolver/Models/ParticleSystem.swift:186-231, whole-array: the reference's array is
    the live tracers in age order (dead ones are removed when they die, :294-296); `count` appends with removeFirst() at
    the cap (:196-199) leave the last `cap` elements of (live + new).  New tracer j: offsets U(-r, r)^2 (:202-205), zero
    velocity (:208), life = alpha ~ U(0.3, 1) (:211, :221), mass ~ U(0.1, 1) (:212), age 0 (:223); the four draws come from
    an xorshift32 stream seeded seed ^ (j + 1) * 0x9E3779B9 (the harness's stand-in for Float.random)."""
    live = arrs[4] != 0
    old = [np.asarray(a, F)[live] for a in arrs]
    j = np.arange(1, count + 1, dtype=np.uint64)
    st = (np.uint32(seed) ^ ((j * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)).astype(np.uint32)
    st = np.where(st == 0, np.uint32(1), st)
    draws = []
    with np.errstate(over="ignore"):
        for lo, hi in ((-radius, radius), (-radius, radius), (0.3, 1.0), (0.1, 1.0)):
            st = _xorshift32(st)
            r01 = (st >> np.uint32(8)).astype(F) * F(1.0 / 16777216.0)
            draws.append(F(lo) + (F(hi) - F(lo)) * r01)
    z = np.zeros(count, F)
    new = [F(x) + draws[0], F(y) + draws[1], z, z, draws[2], draws[3], z]
    if cap <= 0:
        return [a[:0] for a in old]
    return [np.concatenate([a, b]).astype(F)[-cap:] for a, b in zip(old, new)]


# ---- input front-end: ColorPalette.getColor(for:) and FluidSolver.addForce(at:windowSize:) ----------------------------
SWIFT_FLOAT_PI = np.frombuffer(np.uint32(0x40490FDA).tobytes(), dtype=np.float32)[0]   # Float.pi: pi rounded toward zero


def _sinf(x):
    return np.sin(x.astype(np.float64)).astype(F)


def _hsb(h):
    """hsbToRgb(h, 1, 1), fluidSolver/Models/ColorPalette.swift:107-128, for an array of hues."""
    c = F(1.0) * F(1.0)
    x = c * (F(1) - np.abs(np.fmod(h * F(6), F(2)) - F(1)))
    m = F(1.0) - c
    z = np.zeros_like(h)
    cc = np.full_like(h, c)
    sixth = [F(k) / F(6.0) for k in range(1, 6)]
    sector = sum((h >= t).astype(np.int64) for t in sixth)
    table = [(cc, x, z), (x, cc, z), (z, cc, x), (z, x, cc), (x, z, cc), (cc, z, x)]
    rgb = [np.choose(sector, [t[k] for t in table]) + m for k in range(3)]
    return np.stack(rgb + [np.ones_like(h)], axis=-1).astype(F)


def palette_colors(name, frames):
    """getColor(for: Float(frameNumber % 360) / 360.0), ColorPalette.swift:40-95 with the phase of FluidSolver.swift:422-423,
    for an array of frame numbers; whole-array, written from the Swift text (Float arithmetic, Float.pi)."""
    t = (np.asarray(frames) % 360).astype(F) / F(360.0)
    pi = SWIFT_FLOAT_PI
    one = np.ones_like(t)
    ph = t * F(2) * pi
    if name == "Rainbow":
        return _hsb(t)
    if name == "Neon":
        return _hsb(np.fmod(t * F(1.5), F(1.0)))
    if name == "Fire":
        return np.stack([one, np.minimum(F(1.0), t * F(2)), np.maximum(F(0), (t - F(0.5)) * F(2)), one], -1).astype(F)
    if name == "Greyscale":
        b = F(0.5) + F(0.5) * _sinf(ph)
        return np.stack([b, b, b, one], -1).astype(F)
    coef = {"Ocean": ((0.2, 0.3, None), (0.5, 0.4, pi / F(3)), (0.8, 0.2, None)),
            "Plasma": ((0.5, 0.5, None), (0.3, 0.5, F(2) * pi / F(3)), (0.8, 0.2, F(4) * pi / F(3))),
            "Sunset": ((0.9, 0.1, None), (0.3, 0.4, pi / F(2)), (0.4, 0.6, pi)),
            "Vapor": ((0.7, 0.3, None), (0.3, 0.3, pi / F(2)), (0.9, 0.1, pi))}[name]
    ch = [F(a) + F(b) * _sinf(ph if sh is None else ph + sh) for a, b, sh in coef]
    return np.stack(ch + [one], -1).astype(F)


def add_force_inputs(at, window, grid, previous, force_multiplier, color_multiplier, palette, frame):
    """What FluidSolver.addForce(at:windowSize:) hands to applyForce (FluidSolver.swift:409-428): grid position, force from
    the previous touch (zero on the first), palette colour times colorMultiplier.  Returns (pos, force, rgba)."""
    pos = np.array([F(at[0] / window[0]) * F(grid[0]), F(at[1] / window[1]) * F(grid[1])], F)     # :410-413
    force = np.zeros(2, F) if previous is None else (pos - previous) * F(force_multiplier)      # :415-419
    color = palette_colors(palette, np.array([frame]))[0] * F(color_multiplier)                   # :422-424
    return pos, force, color
