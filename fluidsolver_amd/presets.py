"""Parameter presets and colour palettes of the reference (data + host-side input front-end).

Presets: the ten numeric parameter sets of fluidSolver/Models/FluidPresets.swift:27-177.
Palettes: ColorPalette.getColor(for:), fluidSolver/Models/ColorPalette.swift:40-128, evaluated in
fp32 like the Swift `Float` code.
"""
import math
from collections import namedtuple

import numpy as np

F = np.float32
# Swift's Float.pi is pi rounded TOWARD ZERO (0x40490FDA = 3.14159250...), not the nearest float (0x40490FDB)
FLOAT_PI = np.array([0x40490FDA], dtype=np.uint32).view(np.float32)[0]

FluidPreset = namedtuple("FluidPreset", "name viscosity diffusion deltaTime vorticityStrength fadeSpeed "
                                        "solverIterations forceMultiplier colorMultiplier brightness "
                                        "doVorticityConfinement gridResolution")

PRESETS = [
    FluidPreset("Default", 0.00001, 0.0, 0.59, 0.0, 0.005, 15, 24.0, 100.0, 0.06, False, 177),
    FluidPreset("Viscous", 0.001, 0.0, 0.5, 0.0, 0.003, 20, 30.0, 80.0, 0.08, False, 150),
    FluidPreset("Turbulent", 0.00001, 0.0, 0.8, 25.0, 0.002, 10, 50.0, 120.0, 0.1, True, 200),
    FluidPreset("Smoke", 0.00005, 0.00001, 0.4, 10.0, 0.008, 15, 35.0, 60.0, 0.15, True, 128),
    FluidPreset("Water", 0.000001, 0.0, 1.0, 5.0, 0.001, 25, 40.0, 90.0, 0.07, True, 150),
    FluidPreset("Ink", 0.0001, 0.00005, 0.6, 15.0, 0.004, 18, 45.0, 150.0, 0.12, True, 180),
    FluidPreset("Fast Flow", 0.000001, 0.0, 2.0, 8.0, 0.01, 8, 60.0, 80.0, 0.05, True, 100),
    FluidPreset("High Detail", 0.00001, 0.0, 0.5, 12.0, 0.003, 30, 30.0, 100.0, 0.08, True, 256),
    FluidPreset("Paint", 0.0005, 0.0001, 0.3, 0.0, 0.0005, 20, 20.0, 200.0, 0.2, False, 150),
    FluidPreset("Minimal", 0.0, 0.0, 0.5, 0.0, 0.02, 5, 50.0, 50.0, 0.1, False, 80),
]
PRESETS_BY_NAME = {p.name: p for p in PRESETS}

PALETTES = ("Rainbow", "Ocean", "Fire", "Plasma", "Neon", "Sunset", "Vapor", "Greyscale")


def hsb_to_rgb(h, s, b):
    """ColorPalette.swift:107-128 (fp32)."""
    h, s, b = F(h), F(s), F(b)
    c = b * s
    x = c * (F(1) - abs(F(math.fmod(float(h * F(6)), 2.0)) - F(1)))
    m = b - c
    if h < F(1.0) / F(6.0):
        rgb = (c, x, F(0))
    elif h < F(2.0) / F(6.0):
        rgb = (x, c, F(0))
    elif h < F(3.0) / F(6.0):
        rgb = (F(0), c, x)
    elif h < F(4.0) / F(6.0):
        rgb = (F(0), x, c)
    elif h < F(5.0) / F(6.0):
        rgb = (x, F(0), c)
    else:
        rgb = (c, F(0), x)
    return np.array([rgb[0] + m, rgb[1] + m, rgb[2] + m, F(1.0)], dtype=F)


def _sin(x):
    return F(math.sin(float(x)))


def palette_color(palette, time):
    """ColorPalette.getColor(for:), ColorPalette.swift:40-95.  `time` in [0,1)."""
    t = F(time)
    pi = FLOAT_PI
    if palette == "Rainbow":
        return hsb_to_rgb(t, 1.0, 1.0)
    if palette == "Ocean":
        ph = t * F(2) * pi
        return np.array([F(0.2) + F(0.3) * _sin(ph), F(0.5) + F(0.4) * _sin(ph + pi / F(3)),
                         F(0.8) + F(0.2) * _sin(ph), 1.0], dtype=F)
    if palette == "Fire":
        return np.array([1.0, min(F(1.0), t * F(2)), max(F(0), (t - F(0.5)) * F(2)), 1.0], dtype=F)
    if palette == "Plasma":
        ph = t * F(2) * pi
        return np.array([F(0.5) + F(0.5) * _sin(ph), F(0.3) + F(0.5) * _sin(ph + F(2) * pi / F(3)),
                         F(0.8) + F(0.2) * _sin(ph + F(4) * pi / F(3)), 1.0], dtype=F)
    if palette == "Neon":
        return hsb_to_rgb(F(math.fmod(float(t * F(1.5)), 1.0)), 1.0, 1.0)
    if palette == "Sunset":
        ph = t * F(2) * pi
        return np.array([F(0.9) + F(0.1) * _sin(ph), F(0.3) + F(0.4) * _sin(ph + pi / F(2)),
                         F(0.4) + F(0.6) * _sin(ph + pi), 1.0], dtype=F)
    if palette == "Vapor":
        ph = t * F(2) * pi
        return np.array([F(0.7) + F(0.3) * _sin(ph), F(0.3) + F(0.3) * _sin(ph + pi / F(2)),
                         F(0.9) + F(0.1) * _sin(ph + pi), 1.0], dtype=F)
    if palette == "Greyscale":
        br = F(0.5) + F(0.5) * _sin(t * F(2) * pi)
        return np.array([br, br, br, 1.0], dtype=F)
    raise ValueError("unknown palette %r" % (palette,))
